#!/usr/bin/env python3
"""How often does the reference-fold device tree hand a step to the host build over a LONG run of the reference's own scenes?
(nb_random_disk / nb_stable_orbits, 10 000 bodies, theta 0.85, dt 0.01 -- the Haskell app's defaults.)  One JSON line per scene."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS  # noqa: E402

CASES = (("stable_orbits", 10000), ("random_disk", 10000), ("random_disk", 2000), ("stable_orbits", 65536), ("random_disk", 65536))
if len(sys.argv) > 1:   # e.g. random_disk:65536
    CASES = tuple((a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:])
for scene, n in CASES:
    e = rx.NBodyEngine()
    e.seed(11)
    if scene == "stable_orbits":
        e.stable_orbits(n, 0.5, 30.0)
    else:
        e.random_disk(n)
    marks, fb = [], []
    t0 = time.perf_counter()
    steps = 1000 if n <= 10000 else 300
    for k in range(steps):
        e.step_barnes_hut(0.85, 0.01, 1)
        if (k + 1) % (steps // 10) == 0:
            e.synchronize()
            marks.append(round((time.perf_counter() - t0) * 1e3 / (k + 1), 4))
            fb.append(e.get_stat(NBX_STAT_BH_FALLBACKS))
    print(json.dumps({"scene": scene, "n": n, "steps": steps, "fallbacks_cumulative_by_tenth": fb, "ms_per_step_cumulative": marks}), flush=True)
