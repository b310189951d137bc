#!/usr/bin/env python3
"""A few Barnes-Hut steps of the 1 M-body Plummer case (BASELINE config #4) for rocprofv3 runs.
usage: bh_steps.py [host|device] [steps] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402

tree = sys.argv[1] if len(sys.argv) > 1 else "device"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
st = rx.plummer_sphere(n, dim=2)
e = rx.NBodyEngine(mode="fast")
e.set_bh_tree(tree)
e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
import time
e.step_barnes_hut(0.5, 0.01, 1); e.synchronize(); e.bh_host_timing()     # warm-up (allocations)
ts = []
for _ in range(steps):
    t0 = time.perf_counter(); e.step_barnes_hut(0.5, 0.01, 1); e.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print("ok median_ms %.3f min_ms %.3f" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3), e.bh_host_timing())
