#!/usr/bin/env python3
"""Barnes-Hut steps of the reference's own scene (nb_stable_orbits(n, 0.5, 30), theta 0.85, dt 0.01; RustNBodyExperiment.hs:42-47)
for rocprofv3 runs.   usage: bh_small_steps.py [bodies] [steps] [exact|reference]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fold = sys.argv[3] if len(sys.argv) > 3 else "exact"
e = rx.NBodyEngine(mode="fast")
e.set_bh_fold(fold)
e.seed(1); e.stable_orbits(n, 0.5, 30.0)
for _ in range(5):
    e.step_barnes_hut(0.85, 0.01, 1)
e.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    e.step_barnes_hut(0.85, 0.01, 1)
e.synchronize()
print("ok n %d fold %s ms_per_step_back_to_back %.4f" % (n, fold, (time.perf_counter() - t0) / steps * 1e3))
