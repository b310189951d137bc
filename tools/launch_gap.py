#!/usr/bin/env python3
"""Wall time per step vs kernel time per step (HIP events) for small per-GPU shapes: how big are the launch gaps?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx
for n, world in ((262144, 8), (65536, 1), (16384, 1), (10000, 1)):
    st = rx.plummer_sphere(n)
    e = rx.NBodyEngine(); e.set_shard(0, world)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    for _ in range(5): e.step_local(0.01)
    e.synchronize()
    steps = 200
    t0 = time.perf_counter()
    for _ in range(steps): e.step_local(0.01)
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    e.profile(True); e.profile_reset()
    for _ in range(50): e.step_local(0.01)
    ms, cnt = e.profile_read(rx.NBX_K_FORCE); ims, _ = e.profile_read(rx.NBX_K_INTEGRATE)
    print(json.dumps({"n": n, "world": world, "launch": e.last_launch(), "wall_ms_per_step": (t2 - t0) / steps * 1e3,
                      "host_enqueue_ms_per_step": (t1 - t0) / steps * 1e3, "kernels_ms_per_step": (ms + ims) / cnt}), flush=True)
