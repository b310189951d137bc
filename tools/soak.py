#!/usr/bin/env python3
"""Soak: tens of thousands of steps of the reference's scenes through one engine -- no crash, finite state, no growth of the
process (event pools, workspaces, pinned buffers).  python tools/soak.py [steps]"""
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for scene, n in (("random_disk", 10000), ("stable_orbits", 10000), ("random_disk", 30000)):
    e = rx.NBodyEngine(); e.seed(3)
    (e.random_disk if scene == "random_disk" else lambda k: e.stable_orbits(k, 0.5, 30.0))(n)
    e.step_barnes_hut(0.85, 0.01, 1); e.synchronize()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter()
    k_steps = steps if n <= 10000 else steps // 10
    for k in range(k_steps):
        e.step_barnes_hut(0.85, 0.01, 1)
        if k % 500 == 499:
            e.draw(512, 512)
        if k % 5000 == 4999:
            e.step_brute_force(0.01)
    e.synchronize()
    st = e.get_particles()
    ok = bool(np.isfinite(st["px"]).all() and np.isfinite(st["vx"]).all())
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print(scene, n, "steps", k_steps, "finite", ok, "ms/step %.3f" % ((time.perf_counter() - t0) / k_steps * 1e3),
          "handed over", e.get_stat(NBX_STAT_BH_FALLBACKS), "maxrss growth KiB", rss1 - rss0, flush=True)
    e.close()
