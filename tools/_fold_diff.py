import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rust_exp_amd as rx
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n = int(rng.choice([2, 3, 17, 255, 256, 257, 1000, 4097, 9000, 20000, 70000, 150000]))
scale = float(rng.choice([1e-2, 1.0, 30.0, 3e3]))
x = (rng.normal(0, 1, n) * scale).astype(np.float32)
y = (rng.normal(0, 1, n) * scale).astype(np.float32)
if rng.random() < 0.5 and n > 50:
    k = n // 5
    x[:k] = x[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
    y[:k] = y[k:2 * k] + (rng.normal(0, 1e-3, k) * scale).astype(np.float32)
mk = rng.choice(["unit", "wide", "common"])
if mk == "common":
    m = np.full(n, rng.choice([0.37, 1.0, 2.5e-3]), np.float32)
    k_exc = int(rng.choice([0, 1, 5, 30]))
    if k_exc and n > 2 * k_exc:
        m[rng.choice(n, k_exc, replace=False)] = (10.0 ** rng.uniform(-2, 3, k_exc)).astype(np.float32)
else:
    m = {"unit": rng.uniform(0.5, 2.0, n), "wide": 10.0 ** rng.uniform(-3, 3, n)}[mk].astype(np.float32)
e = rx.NBodyEngine()
e.set_particles(x, y, np.zeros(n), np.zeros(n), m)
host = e.bh_flat_dump(False)
dev = e.bh_flat_dump("device")
print("n", n, "scale", scale, "nodes", len(host), len(dev))
if len(host) == len(dev):
    for k in ("px", "py", "m", "s", "skip", "interior"):
        bad = np.flatnonzero(host[k].view(np.uint32) != dev[k].view(np.uint32)) if host[k].dtype.kind == "f" else np.flatnonzero(host[k] != dev[k])
        print(k, bad.size, bad[:10])
        for b in bad[:4]:
            print("   node", b, "host", host[b], "dev", dev[b])
            # members of this node: leaves in [b, skip)
            sub = host[b:host["skip"][b]]
            lf = sub[sub["interior"] == 0]
            print("   leaves", len(lf), lf[:6])
else:
    # first differing position
    L = min(len(host), len(dev))
    d = np.flatnonzero((host["skip"][:L] != dev["skip"][:L]) | (host["interior"][:L] != dev["interior"][:L]))
    print("first structural difference at", d[:5])
    for b in d[:2]:
        print(" host", host[max(0,b-2):b+4]); print(" dev ", dev[max(0,b-2):b+4])
