#!/usr/bin/env python3
"""Probe of the warm sort (bh_sort.hip, round 5): N steps of Barnes-Hut on the device tree, fallbacks and refusal reasons."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_exp_amd as rx
from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_REFUSAL, NBX_STAT_BH_LAST_TREE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
make = sys.argv[3] if len(sys.argv) > 3 else "plummer"
e = rx.NBodyEngine()
if make == "plummer":
    e.plummer_sphere(n, dim=2)
elif make == "orbits":
    e.seed(1); e.stable_orbits(n, 0.5, 30.0)
else:
    e.seed(1); e.random_disk(n)
e.forces(0.5)
for k in range(steps):
    t0 = time.perf_counter()
    e.step_barnes_hut(0.5, 0.01, 1)
    e.synchronize()
    print(k, "ms %.3f" % ((time.perf_counter() - t0) * 1e3), "fallbacks", e.get_stat(NBX_STAT_BH_FALLBACKS), "why 0x%x" % e.get_stat(NBX_STAT_BH_REFUSAL),
          "tree", e.get_stat(NBX_STAT_BH_LAST_TREE))
