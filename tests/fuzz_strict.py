#!/usr/bin/env python3
"""Fuzz campaign of the bit-exact mode against the oracle, run by hand on a GPU box (pytest does not collect it; the
suite's randomized tests use fixed seeds):   python tests/fuzz_strict.py [first_seed] [count]
Random sizes 2 .. 70 000, coordinate scales 1e-3 .. 2e5, mass ranges inside and outside the short-division guard, clumps,
near-duplicates and exact duplicates; one brute-force step and one Barnes-Hut step, positions and velocities compared bit
for bit (a tree the oracle refuses must be refused by the library too).  Half of the cases ask for the device-built tree."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from oracle import binding as ob  # noqa: E402


def main():

    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([2, 3, 17, 255, 256, 257, 1000, 4097, 9000, 20000, 70000]))
        scale = float(rng.choice([1e-3, 1.0, 30.0, 3e3, 2e5]))
        x = (rng.normal(0, 1, n) * scale).astype(np.float32)
        y = (rng.normal(0, 1, n) * scale).astype(np.float32)
        if rng.random() < 0.5 and n > 50:       # clumps, near-duplicates, exact duplicates
            k = n // 5
            x[:k] = x[k:2 * k] + (rng.normal(0, 1e-5, k) * scale).astype(np.float32)
            y[:k] = y[k:2 * k] + (rng.normal(0, 1e-5, k) * scale).astype(np.float32)
            x[2 * k:2 * k + 20] = x[0]; y[2 * k:2 * k + 20] = y[0]
        mk = rng.choice(["unit", "wide", "tiny", "huge"])
        m = {"unit": rng.uniform(0.5, 2.0, n), "wide": 10.0 ** rng.uniform(-9, 9, n), "tiny": 10.0 ** rng.uniform(-14, -8, n),
             "huge": 10.0 ** rng.uniform(8, 13, n)}[mk].astype(np.float32)
        p = ob.particles(x, y, rng.normal(0, 1, n), rng.normal(0, 1, n), m)
        theta = float(rng.choice([0.3, 0.5, 0.85, 0.95]))
        e = rx.NBodyEngine(mode="strict")
        kernel = int(rng.choice([0, 1, 8, 16]))    # NBX_OPT_STRICT_KERNEL: every all-pairs kernel must give the same bits
        e.set_strict_kernel(kernel)
        tree = str(rng.choice(["host", "device"]))   # late round 3: the bit-exact mode on the device-built tree (reference fold,
        e.set_bh_tree(tree)                          # on request only): the same bits, or a refusal that ends in the host build
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        q = p.copy()
        ok = True
        try:
            e.step_brute_force(0.01); ob.step_brute_force(q, 0.01)
            rc = ob.step_barnes_hut(q, theta, 0.01, 4)
            try:
                e.step_barnes_hut(theta, 0.01, 1)
                gpu_rc = 0
            except rx.NBodyError as ex:
                gpu_rc = ex.code
            if rc != 0 or gpu_rc != 0:
                ok = (rc != 0) == (gpu_rc != 0)       # both must refuse (depth > 50 etc.)
            else:
                st = e.get_particles()
                with np.errstate(invalid="ignore"):
                    for kx in ("px", "py", "vx", "vy"):
                        if not np.array_equal(st[kx].view(np.uint32), q[kx].view(np.uint32)):
                            ok = False
        except Exception as ex:   # noqa: BLE001
            ok = False
            print("seed", seed, "exception", repr(ex))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "n", n, "scale", scale, "masses", mk, "theta", theta, "kernel", kernel, "tree", tree)
    print("fuzz: %d cases, %d mismatches, %.1f s" % (count, bad, time.time() - t0))


if __name__ == "__main__":
    main()
