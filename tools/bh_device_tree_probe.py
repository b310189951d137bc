#!/usr/bin/env python3
"""Device-built quadtree vs the oracle (diagnostic behind the tolerances in tests/test_gpu_bh_device_tree.py):
forces through the fast traversal on the host tree and on the device tree against orc_bh_forces, and the structural
difference between the two flattened trees, on presets, sub-EPS pairs in random arrival order, duplicates and clumps."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from oracle import binding as ob  # noqa: E402  (diagnostic tool: the oracle is the checker here)


def systems():
    rng = np.random.default_rng(3)
    yield "orbits_50k", ob.stable_orbits(50000, 0.5, 30.0, 44)
    yield "disk_20k", ob.random_disk(20000, 41)
    st = rx.plummer_sphere(262144, dim=2)
    yield "plummer_262k", ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    st = rx.plummer_sphere(1048576, dim=2)
    yield "plummer_1m", ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    for n0, k in ((5000, 800), (100000, 5000)):
        x = rng.uniform(-20, 20, n0).astype(np.float32); y = rng.uniform(-20, 20, n0).astype(np.float32)
        dx = rng.uniform(-9e-5, 9e-5, k).astype(np.float32); dy = rng.uniform(-9e-5, 9e-5, k).astype(np.float32)
        x = np.concatenate([x, x[:k] + dx]); y = np.concatenate([y, y[:k] + dy])
        perm = rng.permutation(len(x))
        x, y = x[perm], y[perm]
        yield f"pairs_{n0}+{k}", ob.particles(x, y, np.zeros(len(x)), np.zeros(len(x)), rng.uniform(0.5, 2.0, len(x)))
    x = rng.uniform(-20, 20, 3000).astype(np.float32); y = rng.uniform(-20, 20, 3000).astype(np.float32)
    x = np.concatenate([x, x[:500] + np.float32(3e-5), x[:100], x[:100]]); y = np.concatenate([y, y[:500], y[:100], y[:100]])
    yield "pairs_and_triplicates", ob.particles(x, y, np.zeros(len(x)), np.zeros(len(x)), rng.uniform(0.5, 2.0, len(x)))
    c = rng.normal(0, 8, (40, 2)).astype(np.float32)
    pts = (c[rng.integers(0, 40, 30000)] + rng.normal(0, 2e-4, (30000, 2))).astype(np.float32)      # 40 clumps ~ 2 EPS wide
    yield "clumps_30k", ob.particles(pts[:, 0], pts[:, 1], np.zeros(30000), np.zeros(30000), np.ones(30000))


def main():
    out = {}
    for name, p in systems():
        rec = {"n": len(p)}
        for theta in (0.5, 0.85):
            rc, ofx, ofy = ob.bh_forces(p, theta, nthreads=16)
            if rc != 0:
                rec[f"theta{theta}"] = {"oracle_rc": rc}
                continue
            scale = max(np.abs(ofx).max(), np.abs(ofy).max())
            r = {}
            for where in ("host", "device"):
                e = rx.NBodyEngine()
                e.set_bh_tree(where)
                e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
                fx, fy, _ = e.forces(theta)
                err = np.maximum(np.abs(fx - ofx), np.abs(fy - ofy)) / scale
                r[where] = {"max": float(err.max()), "p999": float(np.percentile(err, 99.9)), "median": float(np.median(err)),
                            "last_tree": e.get_stat(rx.engine.NBX_STAT_BH_LAST_TREE), "fallbacks": e.get_stat(rx.engine.NBX_STAT_BH_FALLBACKS)}
            rec[f"theta{theta}"] = r
        e = rx.NBodyEngine()
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        try:
            host = e.bh_flat_dump(False)
            dev = e.bh_flat_dump("device")
            s = {"host_nodes": len(host), "device_nodes": len(dev)}
            if len(host) == len(dev):
                leaf = host["interior"] == 0
                s.update({"skip_equal": bool(np.array_equal(host["skip"], dev["skip"])),
                          "interior_equal": bool(np.array_equal(host["interior"], dev["interior"])),
                          "s_equal": bool(np.array_equal(host["s"].view(np.uint32), dev["s"].view(np.uint32))),
                          "leaf_records_differing": int(sum((host[k][leaf].view(np.uint32) != dev[k][leaf].view(np.uint32)) for k in ("px", "py", "m")).astype(bool).sum()),
                          "host_leaves": int(leaf.sum())})
            else:
                s["host_leaves"] = int((host["interior"] == 0).sum()); s["device_leaves"] = int((dev["interior"] == 0).sum())
            rec["structure"] = s
        except Exception as ex:   # noqa: BLE001
            rec["structure"] = {"error": str(ex)[:200]}
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bh_device_tree_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
