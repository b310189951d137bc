#!/usr/bin/env python3
"""State-machine fuzz of the level-2 API in the bit-exact mode, run by hand on a GPU box (pytest does not collect it):
    python tests/fuzz_api.py [first_seed] [count]
Random sequences of set_particles (sizes change), brute-force / Barnes-Hut steps, forces, draw, get_particles, option
flips (tree on host/device -- ignored by the bit-exact mode --, shared walk on/off, mode fast<->strict and back),
checkpoint save/load; the oracle runs the same sequence and every observable must match bit for bit."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402
from oracle import binding as ob  # noqa: E402
from rust_exp_amd.engine import NBX_OPT_BH_WAVE  # noqa: E402


def make(rng):
    n = int(rng.choice([1, 2, 7, 300, 1024, 5000, 33000, 70000]))
    kind = rng.choice(["disk", "orbits", "normal"])
    if kind == "disk":
        return ob.random_disk(n, int(rng.integers(1, 1 << 30)))
    if kind == "orbits" and n > 1:
        return ob.stable_orbits(n, 0.5, 30.0, int(rng.integers(1, 1 << 30)))
    x = rng.normal(0, 10, n); y = rng.normal(0, 10, n)
    return ob.particles(x, y, rng.normal(0, 1, n), rng.normal(0, 1, n), rng.uniform(0.5, 2, n))


def same(a, b):
    with np.errstate(invalid="ignore"):
        return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        e = rx.NBodyEngine(mode="strict")
        q = make(rng)
        e.set_particles(q["px"], q["py"], q["vx"], q["vy"], q["m"])
        log = []
        ok = True
        for step in range(40):
            op = rng.choice(["brute", "bh", "bh", "get", "draw", "forces", "set", "tree", "wave", "mode", "ckpt"])
            log.append(op)
            try:
                if op == "brute":
                    if len(q) <= 33000:
                        e.step_brute_force(0.01); ob.step_brute_force(q, 0.01)
                elif op == "bh":
                    th = float(rng.choice([0.0, 0.4, 0.85])) if len(q) <= 33000 else float(rng.choice([0.4, 0.85]))
                    rc = ob.step_barnes_hut(q, th, 0.01, 4)
                    try:
                        e.step_barnes_hut(th, 0.01, 1)
                        if rc != 0:
                            ok = False
                    except rx.NBodyError:
                        if rc == 0:
                            ok = False
                        else:       # both refused: the oracle's state is undefined after a panic; restart from a new state
                            q = make(rng); e.set_particles(q["px"], q["py"], q["vx"], q["vy"], q["m"])
                elif op == "get":
                    st = e.get_particles()
                    ok = ok and all(same(st[k], q[k]) for k in ("px", "py", "vx", "vy", "m"))
                elif op == "draw":
                    w, h = int(rng.choice([64, 200, 512])), int(rng.choice([48, 200, 512]))
                    ok = ok and np.array_equal(e.draw(w, h), ob.draw(q, w, h))
                elif op == "forces" and len(q) <= 33000:
                    fx, fy, _ = e.forces(0.0)
                    wx, wy = ob.brute_forces(q, 0, len(q))
                    ok = ok and same(fx, wx) and same(fy, wy)
                elif op == "set":
                    q = make(rng)
                    e.set_particles(q["px"], q["py"], q["vx"], q["vy"], q["m"])
                elif op == "tree":
                    e.set_bh_tree(str(rng.choice(["host", "device"])))
                elif op == "wave":
                    e.set_option(NBX_OPT_BH_WAVE, int(rng.integers(0, 2)))
                elif op == "mode":       # a fast step would leave the bit-exact trajectory: only flip there and back
                    e.set_mode("fast"); e.set_mode("strict")
                elif op == "ckpt":
                    with tempfile.TemporaryDirectory() as d:
                        path = os.path.join(d, "s.ckpt")
                        e.save(path)
                        e2 = rx.NBodyEngine(mode="strict")
                        e2.load(path)
                        st = e2.get_particles()
                        ok = ok and all(same(st[k], q[k]) for k in ("px", "py", "vx", "vy", "m"))
            except Exception as ex:   # noqa: BLE001
                ok = False
                print("seed", seed, "exception in", op, repr(ex))
            if not ok:
                break
        if ok:
            st = e.get_particles()
            ok = all(same(st[k], q[k]) for k in ("px", "py", "vx", "vy"))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "n", len(q), "ops", log)
    print("fuzz api: %d sequences, %d mismatches, %.1f s" % (count, bad, time.time() - t0))


if __name__ == "__main__":
    main()
