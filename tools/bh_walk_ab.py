#!/usr/bin/env python3
"""A/B of the fast Barnes-Hut walks: the node walk of rounds 1-3 (NBX_OPT_BH_WALK = 0) against the
child-group walk of round 4 (1), same bodies, same tree, same step: traversal time (HIP events around conversion + walk),
tree build, ms per step of a back-to-back loop, and the work both do (node visits, pair laws, opening tests, group loads).
One JSON line per (bodies, theta, walk).  Usage: bh_walk_ab.py [n:theta ...]   (default: the sizes DESIGN.md quotes)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import (NBX_OPT_BH_WALK, NBX_OPT_BH_WAVE, NBX_STAT_BH_CLASS_SWITCHES, NBX_STAT_BH_FALLBACKS,  # noqa: E402
                                 NBX_STAT_BH_LAST_TREE)


def run(n, theta, walk, wave=1, steps=30, fold="exact", tree=None):
    e = rx.NBodyEngine(mode="fast")
    e.set_bh_fold(fold)
    if tree:
        e.set_bh_tree(tree)
    e.set_option(NBX_OPT_BH_WALK, walk)
    e.set_option(NBX_OPT_BH_WAVE, wave)
    if os.environ.get("NBX_AB_FUSE_KICK"):
        from rust_exp_amd.engine import NBX_OPT_BH_FUSE_KICK
        e.set_option(NBX_OPT_BH_FUSE_KICK, int(os.environ["NBX_AB_FUSE_KICK"]))
    if n == 10000:
        e.seed(1); e.stable_orbits(n, 0.5, 30.0)
    else:
        st = rx.plummer_sphere(n, dim=2)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for _ in range(5):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    e.profile(True); e.profile_reset(); e.bh_host_timing()
    for _ in range(steps):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    ev, cnt = e.profile_read(rx.NBX_K_BH_EVAL)
    tb, tcnt = e.profile_read(rx.NBX_K_TREE_BUILD)
    ki, kcnt = e.profile_read(rx.NBX_K_INTEGRATE)      # (0 launches when the walk applies the kick-drift itself)
    e.profile(False)
    wk = e.bh_work_detail(theta)
    fx, fy, _ = e.forces(theta)
    out = {"bodies": n, "theta": theta, "walk": {0: "nodes", 1: "groups", 2: "groups_compiled"}[walk], "wave": wave, "fold": fold,
           "ms_per_step_back_to_back": round(ms_step, 4), "traversal_ms": round(ev / max(cnt, 1), 4),
           "tree_build_ms": round(tb / max(tcnt, 1), 4), "kick_drift_ms": round(ki / max(kcnt, 1), 4), "nodes": e.bh_host_timing()["nodes"],
           "visits_per_body": wk["node_visits"] / n, "pairs_per_body": wk["pair_evals"] / n,
           "opening_tests_per_body": wk["opening_tests"] / n, "group_loads_per_body": wk["group_loads"] / n,
           "force_checksum": float(np.abs(fx).sum() + np.abs(fy).sum()),
           "last_tree": "device" if e.get_stat(NBX_STAT_BH_LAST_TREE) == 1 else "host", "host_hand_overs": e.get_stat(NBX_STAT_BH_FALLBACKS),
           "class_switches": e.get_stat(NBX_STAT_BH_CLASS_SWITCHES)}
    e.close()
    return out


def main():
    cases = [a.split(":") for a in sys.argv[1:]] or [("10000", "0.85"), ("65536", "0.5"), ("262144", "0.5"), ("1048576", "0.5")]
    for n, theta in cases:
        for walk in (0, 2, 1):
            print(json.dumps(run(int(n), float(theta), walk)), flush=True)
        if int(n) <= 65536:
            print(json.dumps(run(int(n), float(theta), 1, fold="reference")), flush=True)


if __name__ == "__main__":
    main()
