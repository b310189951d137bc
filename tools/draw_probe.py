#!/usr/bin/env python3
"""host nb_draw timing (state resident on the host: pure splat) -- python tools/draw_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx
out = {}
for n in (65536, 262144, 1048576):
    e = rx.NBodyEngine()
    e.seed(3); e.stable_orbits(n, 0.5, 30.0)
    e.draw(512, 512)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); e.draw(512, 512); ts.append(time.perf_counter() - t0)
    out[n] = float(np.median(ts)) * 1e3
print(json.dumps({"threads": os.environ.get("NBX_HOST_THREADS", "default"), "draw_ms": out}))
