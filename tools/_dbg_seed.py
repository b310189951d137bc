import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rust_exp_amd as rx
from fuzz_fast import make_case
seed = int(sys.argv[1])
x, y, vx, vy, m, theta, mk, clumps, scale = make_case(seed)
n = len(x)
print("seed", seed, "n", n, "scale", scale, "clumps", clumps, flush=True)
e = rx.NBodyEngine(); e.set_particles(x, y, vx, vy, m); e.set_bh_fold("reference")
host = e.bh_flat_dump(False)
try:
    dev = e.bh_flat_dump("device")
except Exception as ex:
    print("device dump refused:", ex); sys.exit(0)
print("nodes", len(host), len(dev))
if len(host) != len(dev):
    # first structural difference
    k = min(len(host), len(dev))
    d = np.flatnonzero((host["skip"][:k] != dev["skip"][:k]) | (host["interior"][:k] != dev["interior"][:k]))
    print("first diffs at", d[:10])
    i = int(d[0]) if len(d) else k - 1
    for j in range(max(0, i - 3), min(k, i + 6)):
        print(j, "H", host[j], "D", dev[j])
else:
    for kf in ("skip", "interior", "px", "py", "m", "s"):
        bad = np.flatnonzero(host[kf].view(np.uint32) != dev[kf].view(np.uint32)) if host[kf].dtype.kind == "f" else np.flatnonzero(host[kf] != dev[kf])
        print(kf, len(bad), bad[:8])
        for j in bad[:4]:
            print("   ", j, "H", host[j], "D", dev[j])
# duplicates
pts = np.stack([x, y], 1)
u, c = np.unique(pts, axis=0, return_counts=True)
print("distinct positions", len(u), "max multiplicity", c.max(), "positions with copies", int((c > 1).sum()))
