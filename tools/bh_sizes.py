#!/usr/bin/env python3
"""Barnes-Hut step, traversal and build times of the shipped path (child-group walk, device tree) over sizes; both tree classes
where they apply, and -- round 6 -- the DEFAULT class (NBX_OPT_BH_FOLD = -1: by cost) beside them: no size's default step may be
slower than the next larger size's.  One JSON line per (bodies, fold).  Usage: bh_sizes.py [n:theta ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bh_walk_ab import run  # noqa: E402

cases = [a.split(":") for a in sys.argv[1:]] or [("600", "0.85"), ("1024", "0.85"), ("2000", "0.85"), ("4096", "0.85"), ("10000", "0.85"), ("16384", "0.85"), ("20000", "0.5"), ("32768", "0.5"),
                                                  ("65536", "0.5"), ("131072", "0.5"), ("262144", "0.5"), ("524288", "0.5"),
                                                  ("1048576", "0.5"), ("2097152", "0.5")]
for n, theta in cases:
    print(json.dumps(run(int(n), float(theta), 1, fold="auto")), flush=True)
    print(json.dumps(run(int(n), float(theta), 1)), flush=True)
    if int(n) <= 65536:
        r = run(int(n), float(theta), 1, fold="reference", tree="device")   # (below 1 024 bodies that class defaults to the host build)
        print(json.dumps(r), flush=True)
