#!/usr/bin/env python3
"""Barnes-Hut steps of the benchmark's 2-D Plummer model beyond config #4's size: ms per step and how many steps the device
build handed to the host tree (chains of bodies within EPS: docs/rounds/r05.md section 10).  One JSON line per size.
Usage: bh_dense_probe.py [bodies ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_REFUSAL  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1048576, 1500000, 2097152, 2600000, 3000000, 4194304]
for n in sizes:
    st = rx.plummer_sphere(n, dim=2)
    e = rx.NBodyEngine(mode="fast")
    e.set_bh_fold("exact")
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for _ in range(5):
        e.step_barnes_hut(0.5, 0.01, 1)
    e.synchronize()
    f_warm = e.get_stat(NBX_STAT_BH_FALLBACKS)
    t0 = time.perf_counter()
    for _ in range(30):
        e.step_barnes_hut(0.5, 0.01, 1)
    e.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(json.dumps({"scene": "plummer_sphere dim=2 a=5", "bodies": n, "theta": 0.5, "steps": 30, "ms_per_step": round(ms, 3),
                      "handed_to_host_tree_in_warmup": f_warm, "handed_to_host_tree": e.get_stat(NBX_STAT_BH_FALLBACKS) - f_warm,
                      "last_refusal": hex(e.get_stat(NBX_STAT_BH_REFUSAL))}), flush=True)
    e.close()
