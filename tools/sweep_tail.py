#!/usr/bin/env python3
"""Does a finer source split (more, smaller workgroups) fix the tail effect of small per-GPU target counts?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx
for n, world in ((262144, 8), (262144, 4), (262144, 2), (65536, 1), (262144, 1)):
    st = rx.plummer_sphere(n)
    for bpt in (2, 4):
        for s in (32, 64, 128, 256, 512):
            e = rx.NBodyEngine(); e.set_shard(0, world)
            e.set_launch(jsplit=s, bodies_per_thread=bpt, variant=5)
            e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
            for _ in range(3): e.step_local(0.01)
            e.synchronize()
            e.profile(True); e.profile_reset()
            for _ in range(10): e.step_local(0.01)
            ms, cnt = e.profile_read(rx.NBX_K_FORCE); ims, _ = e.profile_read(rx.NBX_K_INTEGRATE)
            lo, hi = e.slab()
            print(json.dumps({"n": n, "world": world, "bpt": bpt, "S": e.last_launch()["jsplit"], "grid": e.last_launch()["grid"],
                              "k1_ms": ms / cnt, "k2_ms": ims / cnt, "rate": (hi - lo) * (n - 1) / ((ms + ims) / cnt * 1e-3)}), flush=True)
            e.close()
