#!/usr/bin/env python3
"""The A/B table of the fast Barnes-Hut walks from what tools/bh_walk_ab.py and tools/bh_walk_pmc.sh left in
gpurun_out/: per size and walk -- traversal ms (HIP events: conversion + walk), walk kernel ms (rocprofv3), wave turns (scalar
load pairs / node visits), instructions per turn, VALU / scalar busy, wave slots occupied, SQ_WAIT_INST_ANY share, scalar-cache
hit rate, HBM bytes (FETCH_SIZE x 2 KiB + WRITE_SIZE KiB).   usage: bh_walk_table.py <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
O = os.path.join(ROOT, "gpurun_out")
ab = {}
p = os.path.join(O, f"{tag}_bh_walk_ab.jsonl")
if os.path.exists(p):
    for ln in open(p):
        d = json.loads(ln)
        ab[(d["bodies"], d["walk"], d["fold"])] = d
names = {0: "nodes", 1: "groups", 2: "groups_compiled"}
print("%-8s %-16s %9s %9s %9s %7s %7s %7s %6s %6s %6s %6s %6s %9s" % ("bodies", "walk", "trav_ms", "walk_ms", "turns", "VALU/t", "SALU/t", "br/t", "valu", "scal", "slots", "wait", "K$hit", "HBM_MB"))
for n in (10000, 1048576):
    f = os.path.join(O, f"{tag}_bh_walk_pmc_n{n}.json")
    if not os.path.exists(f):
        continue
    pm = json.load(open(f))
    for w in (0, 2, 1):
        walk = [v for k, v in pm.items() if k.startswith(f"walk{w}:") and ("k_bh_walk_groups" in k or "k_bh_eval" in k)]
        conv = [v for k, v in pm.items() if k.startswith(f"walk{w}:") and "k_bh_groups" in k]
        if not walk:
            continue
        v = walk[0]
        turns = v.get("SQ_INSTS_SMEM", 0) / (2.0 if w else 1.0)
        t = ab.get((n, names[w], "exact"), {})
        hbm = (v.get("hbm_read_bytes", 0) + v.get("hbm_write_bytes", 0) + sum(c.get("hbm_read_bytes", 0) + c.get("hbm_write_bytes", 0) for c in conv)) / 1e6
        print("%-8d %-16s %9.4f %9.4f %9.0f %7.1f %7.1f %7.1f %6.2f %6.2f %6.2f %6.2f %6.2f %9.1f" % (
            n, names[w], t.get("traversal_ms", float("nan")), v.get("avg_duration_ns_under_pmc", 0) / 1e6, turns,
            v.get("SQ_INSTS_VALU", 0) / max(turns, 1), v.get("SQ_INSTS_SALU", 0) / max(turns, 1), v.get("SQ_INSTS_BRANCH", 0) / max(turns, 1),
            v.get("valu_busy_frac_per_simd", float("nan")), v.get("scalar_busy_frac_per_simd", float("nan")),
            v.get("wave_slots_occupied_frac", float("nan")), v.get("wait_inst_any_share", float("nan")),
            v.get("scalar_cache_hit_rate", float("nan")), hbm))
