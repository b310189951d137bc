#!/usr/bin/env python3
"""Timeline of the walk kernel (nbx_bh_walk_trace): when and where each of the walks of one traversal ran.  Answers: how long do
walks take and how much do they differ; does a walk's length follow the number of groups it loads; how many walks are resident
over time (the tail); how evenly are the XCDs / CUs loaded.  usage: bh_walk_trace.py [bodies] [theta]  -> one JSON object"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
    theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    e = rx.NBodyEngine(mode="fast")
    e.set_bh_fold("exact")
    st = rx.plummer_sphere(n, dim=2)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for _ in range(3):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    e.forces(theta)
    tr = e.bh_walk_trace(theta)
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    ran = t1 > 0
    t0, t1 = t0[ran], t1[ran]
    turns = (tr[ran, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    hw = (tr[ran, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (tr[ran, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    begin = int(t0.min())          # s_memrealtime: one 100 MHz clock for the whole device
    t0 = t0 - begin
    t1 = t1 - begin
    begin = 0
    dur = (t1 - t0).astype(np.float64)
    span = float(t1.max() - begin)
    # residency over time
    edges = np.linspace(0, span, 41)
    mid = 0.5 * (edges[1:] + edges[:-1])
    resident = [(int(((t0 - begin) <= m).sum() - ((t1 - begin) <= m).sum())) for m in mid]
    place = xcc * 4096 + se * 512 + sh * 256 + cu * 4 + simd
    per_simd_busy = {}
    for pl in np.unique(place):
        sel = place == pl
        per_simd_busy[int(pl)] = float(dur[sel].sum())
    busy = np.array(list(per_simd_busy.values()))
    per_xcc = {}
    for x in np.unique(xcc):
        sel = xcc == x
        sx = float(t1[sel].max())
        ed = np.linspace(0, sx, 21)
        md = 0.5 * (ed[1:] + ed[:-1])
        per_xcc[int(x)] = {"walks": int(sel.sum()), "span_ticks": sx, "turns": int(turns[sel].sum()),
                           "sum_walk_ticks_over_span_x_1024_slots": float(dur[sel].sum() / (sx * 1024.0)),
                           "second_round_first_start": float(np.sort(t0[sel])[min(1024, int(sel.sum()) - 1)]),
                           "resident_over_time_20_bins": [int(((t0[sel]) <= m).sum() - ((t1[sel]) <= m).sum()) for m in md]}
    c = np.corrcoef(turns, dur)[0, 1] if turns.std() > 0 else float("nan")
    out = {"bodies": n, "theta": theta, "walks": int(ran.sum()), "tick_ns": 10, "kernel_span_ticks": span,
           "walk_ticks": {"mean": float(dur.mean()), "p05": float(np.percentile(dur, 5)), "p50": float(np.median(dur)),
                          "p95": float(np.percentile(dur, 95)), "max": float(dur.max())},
           "walk_turns": {"mean": float(turns.mean()), "p05": float(np.percentile(turns, 5)), "p50": float(np.median(turns)),
                          "p95": float(np.percentile(turns, 95)), "max": int(turns.max())},
           "corr_turns_ticks": float(c), "ticks_per_turn_mean": float((dur / np.maximum(turns, 1)).mean()),
           "start_ticks_after_kernel_start": {"p50": float(np.median(t0 - begin)), "p95": float(np.percentile(t0 - begin, 95)),
                                              "first_round_started_by": float(np.sort(t0 - begin)[min(8191, len(t0) - 1)])},
           "resident_walks_over_time_40_bins": resident, "simd_slots_seen": int(len(busy)),
           "sum_of_walk_ticks_per_simd": {"mean": float(busy.mean()), "min": float(busy.min()), "max": float(busy.max())},
           "per_xcc": per_xcc}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
