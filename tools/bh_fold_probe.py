#!/usr/bin/env python3
"""Device-built quadtree with the reference's running fold (NBX_OPT_BH_FOLD = 1) against the host tree: are the flattened trees
bit-identical, and what does a Barnes-Hut step cost with fold = reference / exact and with the host build?
Usage: python tools/bh_fold_probe.py [theta]   -> one JSON line per size."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def state(kind, n):
    e = rx.NBodyEngine()
    e.seed(5)
    if kind == "orbits":
        e.stable_orbits(n, 0.5, 30.0)
    elif kind == "disk":
        e.random_disk(n)
    else:
        e.plummer_sphere(n, dim=2)
    return e.get_particles()


def timed(st, theta, tree, fold, steps):
    e = rx.NBodyEngine()
    e.set_bh_tree(tree)
    e.set_bh_fold(fold)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for _ in range(5):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.step_barnes_hut(theta, 0.01, 1)
    e.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, e.get_stat(rx.engine.NBX_STAT_BH_LAST_TREE), e.get_stat(rx.engine.NBX_STAT_BH_FALLBACKS)


def main():
    theta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.85
    for kind, n in (("orbits", 1000), ("orbits", 10000), ("disk", 10000), ("plummer", 10000), ("orbits", 30000), ("disk", 65536), ("plummer", 65536),
                    ("plummer", 131072), ("plummer", 262144)):
        st = state(kind, n)
        e = rx.NBodyEngine()
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
        e.set_bh_fold("reference")
        rec = {"kind": kind, "n": n, "theta": theta}
        try:
            host, dev = e.bh_flat_dump(False), e.bh_flat_dump("device")
            same = len(host) == len(dev)
            rec["nodes"] = [len(host), len(dev)]
            if same:
                for k in ("px", "py", "m", "s", "q"):
                    bad = int((host[k].view(np.uint32) != dev[k].view(np.uint32)).sum())
                    rec["diff_" + k] = bad
                rec["diff_skip"] = int((host["skip"] != dev["skip"]).sum())
        except rx.NBodyError as ex:
            rec["dump_error"] = str(ex)
        for tree, fold in (("device", "reference"), ("device", "exact"), ("host", "auto")):
            ms, last, fb = timed(st, theta, tree, fold, 50 if n <= 65536 else 20)
            rec[f"ms_{tree}_{fold}"] = round(ms, 4)
            rec[f"tree_{tree}_{fold}"] = [last, fb]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
