#!/usr/bin/env python3
"""Barnes-Hut steps of the benchmark's 2-D Plummer model beyond config #4's size: ms per step and how many steps the device
build handed to the host tree (chains of bodies within EPS: docs/rounds/r05.md section 10).  One JSON line per size.
Round 6: chains of close bodies are replayed on the device (bh_build.hip 3b) -- the tallies of the last build (bodies merged, merges
decided on a search cut short) are printed, and with --accuracy the force error of the engine in its FINAL state (35 steps into the
collapse) against the oracle's fp64 arbiter on the reference's tree (bench.bh_accuracy, the checker leg of bench.py), like c4_err_*.
Usage: bh_dense_probe.py [--accuracy] [bodies ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_STAT_BH_CHAIN_APPROX, NBX_STAT_BH_CHAIN_MERGED, NBX_STAT_BH_FALLBACKS, NBX_STAT_BH_REFUSAL  # noqa: E402

ACC = "--accuracy" in sys.argv
sizes = [int(a) for a in sys.argv[1:] if a != "--accuracy"] or [1048576, 1500000, 2097152, 2600000, 3000000, 4194304]
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
    rec = {"scene": "plummer_sphere dim=2 a=5", "bodies": n, "theta": 0.5, "steps": 30, "ms_per_step": round(ms, 3),
           "handed_to_host_tree_in_warmup": f_warm, "handed_to_host_tree": e.get_stat(NBX_STAT_BH_FALLBACKS) - f_warm,
           "last_refusal": hex(e.get_stat(NBX_STAT_BH_REFUSAL)), "bodies_merged_by_the_chain_replay_last_build": e.get_stat(NBX_STAT_BH_CHAIN_MERGED),
           "merges_on_a_search_cut_short_last_build": e.get_stat(NBX_STAT_BH_CHAIN_APPROX)}
    if ACC:
        import numpy as np
        import bench
        q = e.get_particles()
        fin = {k: np.ascontiguousarray(q[k]) for k in ("px", "py", "vx", "vy", "m")}
        fin["pz"] = np.zeros(n, np.float32); fin["vz"] = np.zeros(n, np.float32)
        acc = bench.bh_accuracy(fin, 0.5, e, os.cpu_count() or 16)
        rec["force_error_after_35_steps"] = {k: acc.get(k) for k in ("vs", "p50", "p99", "p999", "max", "bodies_beyond_2e-5", "bodies_beyond_2e-4",
                                                                    "within_allowance", "oracle_f32_vs_arbiter", "oracle_seconds", "error")}
        rec["force_error_tree"] = "device" if e.get_stat(NBX_STAT_BH_FALLBACKS) - f_warm == rec["handed_to_host_tree"] else "host"
    print(json.dumps(rec), flush=True)
    e.close()
