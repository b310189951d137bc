#!/usr/bin/env python3
"""One case of tests/fuzz_fast.py by seed: the device-built flattened tree (reference fold) against the host tree, node by node.
    python tools/fuzz_case_tree_diff.py <seed>
Prints the first differing nodes, or why the device build handed the case over (NBX_LOG=1 adds the build's reason bits)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_exp_amd as rx  # noqa: E402
from fuzz_fast import make_case  # noqa: E402

seed = int(sys.argv[1])
x, y, vx, vy, m, theta, mk, clumps, scale = make_case(seed)
n = len(x)
print("seed", seed, "n", n, "scale", scale, "masses", mk, "clusters added", clumps, flush=True)
e = rx.NBodyEngine()
e.set_particles(x, y, vx, vy, m)
e.set_bh_fold("reference")
host = e.bh_flat_dump(False)
try:
    dev = e.bh_flat_dump("device")
except rx.NBodyError as ex:
    print("device build handed the case over:", ex)
    sys.exit(0)
print("nodes: host", len(host), "device", len(dev))
k = min(len(host), len(dev))
for f in ("skip", "interior", "px", "py", "m", "s"):
    a, b = host[f][:k], dev[f][:k]
    bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32)) if a.dtype.kind == "f" else np.flatnonzero(a != b)
    print(f, len(bad), "differ", bad[:8])
    for j in bad[:3]:
        print("    node", j, "host", host[j], "device", dev[j])
u, c = np.unique(np.stack([x, y], 1), axis=0, return_counts=True)
print("distinct positions", len(u), "| positions held by several bodies", int((c > 1).sum()), "| most bodies on one position", int(c.max()))
