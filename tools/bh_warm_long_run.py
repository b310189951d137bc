#!/usr/bin/env python3
"""Long runs on the device tree with the warm sort (bh_sort.hip, round 5): how often is a build refused (bucket overflow of the
sort, EPS crowds, ...) and handed to the host tree over 1 000 steps of systems that collapse, orbit or fly apart?
One JSON line per scene: fallbacks (cumulative, by tenth of the run), the last refusal's reasons, ms per step (cumulative)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_REFUSAL  # noqa: E402

CASES = (("plummer", 1048576), ("random_disk", 262144), ("stable_orbits", 262144), ("stable_orbits", 1048576), ("two_galaxies", 524288),
         ("random_disk", 65536))
if len(sys.argv) > 1:   # e.g. plummer:1048576
    CASES = tuple((a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:])
STEPS = int(os.environ.get("NBX_LONG_STEPS", "1000"))
for scene, n in CASES:
    e = rx.NBodyEngine()
    if os.environ.get("NBX_LONG_FOLD"):     # reference | exact (default: the engine's own choice, by cost)
        e.set_bh_fold(os.environ["NBX_LONG_FOLD"])
    e.seed(11)
    if scene == "stable_orbits":
        e.stable_orbits(n, 0.5, 30.0)
    elif scene == "random_disk":
        e.random_disk(n)
    elif scene == "two_galaxies":
        e.two_galaxies(n)
    else:
        e.plummer_sphere(n, dim=2)
    e.forces(0.5)
    marks, fb = [], []
    e.synchronize()
    t0 = time.perf_counter()
    for k in range(STEPS):
        e.step_barnes_hut(0.5, 0.01, 1)
        if (k + 1) % (STEPS // 10) == 0:
            e.synchronize()
            marks.append(round((time.perf_counter() - t0) * 1e3 / (k + 1), 4))
            fb.append(e.get_stat(NBX_STAT_BH_FALLBACKS))
    st = e.get_particles()
    import numpy as np
    print(json.dumps({"scene": scene, "n": n, "steps": STEPS, "theta": 0.5, "fallbacks_cumulative_by_tenth": fb,
                      "fallback_rate": fb[-1] / STEPS, "class_switches": e.get_stat(NBX_STAT_BH_CLASS_SWITCHES),
                      "fold": os.environ.get("NBX_LONG_FOLD", "default"), "last_refusal_why": "0x%x" % e.get_stat(NBX_STAT_BH_REFUSAL),
                      "ms_per_step_cumulative": marks, "finite": bool(np.isfinite(st["px"]).all() and np.isfinite(st["vx"]).all()),
                      "inc_sort": os.environ.get("NBX_INC_SORT", "1")}), flush=True)
