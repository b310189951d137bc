#!/usr/bin/env python3
"""Fuzz of the single-process multi-engine group (NBX_GROUP_EXCHANGE=copy: several engines share the one GPU), run by
hand on a GPU box (pytest does not collect it):   python tests/fuzz_group.py [first_seed] [count]
Random group sizes 2..6, body counts from fewer-than-engines to 80 000 (even and ragged slabs, empty slabs), bit-exact
and fast modes, host and device trees; the group must reproduce the plain engine bit for bit (fast brute force:
within the tolerance of a different launch shape)."""
import os
import sys
import time

import numpy as np

os.environ["NBX_GROUP_EXCHANGE"] = "copy"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from rust_exp_amd.engine import NBX_OPT_BH_TREE  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        G = int(rng.integers(2, 7))
        n = int(rng.choice([1, 2, 3, 5, 64, 257, 1000, 4099, 20000, 66000, 80000]))
        mode = str(rng.choice(["strict", "fast"]))
        tree = int(rng.integers(0, 2))
        x = rng.normal(0, 12, n).astype(np.float32); y = rng.normal(0, 12, n).astype(np.float32)
        vx = rng.normal(0, 1, n).astype(np.float32); vy = rng.normal(0, 1, n).astype(np.float32)
        m = rng.uniform(0.5, 2.0, n).astype(np.float32)
        why = []
        try:
            g = rx.NBodyGroup([0] * G, mode=mode); g.set_option(NBX_OPT_BH_TREE, tree); g.set_particles(x, y, vx, vy, m)
            e = rx.NBodyEngine(mode=mode); e.set_option(NBX_OPT_BH_TREE, tree); e.set_particles(x, y, vx, vy, m)
            ops = [str(o) for o in rng.choice(["bh", "brute", "bh0"], 5)]
            for k, op in enumerate(ops):
                th = float(rng.choice([0.4, 0.85]))
                exact = mode == "strict" or all(o == "bh" for o in ops[:k + 1])
                if op == "bh":
                    g.step_barnes_hut(th, 0.01, 1); e.step_barnes_hut(th, 0.01, 1)
                elif op == "brute":
                    g.step_brute_force(0.01); e.step_brute_force(0.01)
                else:
                    g.step_barnes_hut(0.0, 0.01, 1); e.step_barnes_hut(0.0, 0.01, 1)
                a, b = g.get_particles(), e.get_particles()
                for key in ("px", "py", "vx", "vy"):
                    if exact:
                        if not np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)):
                            why.append(f"{op}#{k} {key} bits")
                    elif not np.allclose(a[key], b[key], rtol=0, atol=(1e-4 if key[0] == "p" else 5e-2)):
                        why.append(f"{op}#{k} {key} tol {np.abs(a[key] - b[key]).max():.2e}")
                if why:
                    break
            if not why and not np.array_equal(g.draw(64, 64), e.draw(64, 64)) and mode == "strict":
                why.append("draw")
            g.close()
        except Exception as ex:   # noqa: BLE001
            why.append("exception " + repr(ex))
        if why:
            bad += 1
            print("FAIL seed", seed, "G", G, "n", n, mode, "tree", tree, why[:3])
    print("fuzz group: %d cases, %d failures, %.1f s" % (count, bad, time.time() - t0))


if __name__ == "__main__":
    main()
