#!/usr/bin/env python3
"""Launch-shape sweep of K1 on one GPU for several (N, world) shapes; world > 1 emulates the
per-GPU compute of a sharded run (rank 0's slab, no all-gather). Prints JSON lines."""
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402

shapes = [(262144, 1), (262144, 2), (262144, 4), (262144, 8), (65536, 1), (16384, 1), (10000, 1), (1048576, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("/")) for a in sys.argv[1:]]
for n, world in shapes:
    st = rx.plummer_sphere(n)
    best = None
    for variant, bpt, s in itertools.product(tuple(int(v) for v in os.environ.get("VARIANTS", "1").split(",")), (2, 4), (0, 8, 16, 32, 64)):
        e = rx.NBodyEngine()
        e.set_shard(0, world)
        e.set_launch(jsplit=s, bodies_per_thread=bpt, variant=variant)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
        for _ in range(2):
            e.step_local(0.01)
        e.synchronize()
        steps = 5 if n >= 262144 else 20
        t0 = time.perf_counter()
        for _ in range(steps):
            e.step_local(0.01)
        e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        lo, hi = e.slab()
        rate = (hi - lo) * (n - 1) / dt
        rec = {"n": n, "world": world, "variant": variant, "bpt": bpt, "jsplit_req": s, "launch": e.last_launch(),
               "ms": dt * 1e3, "per_gpu_interactions_per_s": rate}
        print(json.dumps(rec), flush=True)
        if best is None or rate > best["per_gpu_interactions_per_s"]:
            best = rec
        e.close()
    print("BEST " + json.dumps(best), flush=True)
