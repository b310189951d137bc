#!/usr/bin/env python3
"""Bit-exact Barnes-Hut step timings (host tree + k_bh_eval_strict): python tools/bh_strict_probe.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402

out = {}
for n, theta in ((10000, 0.85), (100000, 0.85), (1048576, 0.5)):
    e = rx.NBodyEngine(mode="strict")
    if n <= 100000:
        e.seed(5); e.stable_orbits(n, 0.5, 30.0)     # the reference's default scene (hs:42), seeded
    else:
        st = rx.plummer_sphere(n, dim=2)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e.step_barnes_hut(theta, 0.01, 1); e.synchronize()
    e.profile(True); e.profile_reset(); e.bh_host_timing()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); e.step_barnes_hut(theta, 0.01, 1); e.synchronize(); ts.append(time.perf_counter() - t0)
    ms, cnt = e.profile_read(rx.NBX_K_BH_EVAL)
    out[str(n)] = {"theta": theta, "ms_per_step_median": float(np.median(ts)) * 1e3, "eval_kernel_ms": ms / cnt, "host": e.bh_host_timing()}
print(json.dumps(out))
