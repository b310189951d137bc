"""GPU: the bit-exact mode (k_force_strict + k_integrate_f2) against the EXACT binary32 model of tests/f32_exact.py
(fractions.Fraction, one explicit round-to-nearest-even per operation of rs-src/nbody.rs:132-160, :174-183) on the same
120-pair table the oracle is pinned to in tests/test_oracle_exact_arithmetic.py -- no CPU float on the checking side."""
import random

import numpy as np
import pytest

import f32_exact as fx
from test_oracle_exact_arithmetic import CASES, _same, bh_cases, bh_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt_bits", [0x3C23D70A, 0x3BA3D70A])
def test_strict_gpu_step_equals_exact_binary32_arithmetic(rx, dt_bits):
    dt = fx.from_bits(dt_bits)
    rng = random.Random(7)
    e = rx.NBodyEngine(mode="strict")
    checked = 0
    for k, c in enumerate(CASES):
        third = CASES[(k * 7 + 3) % len(CASES)]
        for nb in (2, 3):
            rows = [c[:3], c[3:]] + ([third[:3]] if nb == 3 else [])
            vel = [((rng.randint(-3, 3) + 127) << 23 | rng.getrandbits(23) | (rng.getrandbits(1) << 31),
                    (rng.randint(-3, 3) + 127) << 23 | rng.getrandbits(23) | (rng.getrandbits(1) << 31)) for _ in rows]
            bodies = [[fx.from_bits(r[0]), fx.from_bits(r[1]), fx.from_bits(v[0]), fx.from_bits(v[1]), fx.from_bits(r[2])]
                      for r, v in zip(rows, vel)]
            want = fx.brute_step(bodies, dt)
            arr = lambda col: np.array(col, dtype=np.uint32).view(np.float32)   # noqa: E731  bit patterns, not float math
            e.set_particles(arr([r[0] for r in rows]), arr([r[1] for r in rows]), arr([v[0] for v in vel]),
                            arr([v[1] for v in vel]), arr([r[2] for r in rows]))
            e.step_brute_force(float(np.array([dt_bits], np.uint32).view(np.float32)[0]))
            st = e.get_particles()
            for i in range(nb):
                got = [int(st[f][i:i + 1].view(np.uint32)[0]) for f in ("px", "py", "vx", "vy")]
                exp = [fx.to_bits(want[i][j]) for j in range(4)]
                assert all(_same(g, x) for g, x in zip(got, exp)), (k, nb, i, [hex(x) for x in got], [hex(x) for x in exp])
                checked += 1
    assert checked == 600


def test_strict_gpu_barnes_hut_equals_exact_binary32_arithmetic(rx):
    """The bit-exact Barnes-Hut path (host quadtree + k_bh_eval_strict) against the exact Fraction model of nbody.rs:203-377 on
    the 40 small systems of the oracle's test: every body's force, bit for bit."""
    e = rx.NBodyEngine(mode="strict")
    checked = 0
    for case, (cols, theta_bits) in enumerate(bh_cases()):
        n = len(cols)
        try:
            want, _ = bh_model(cols, theta_bits)
        except AssertionError:
            continue
        arr = lambda col: np.array(col, dtype=np.uint32).view(np.float32)   # noqa: E731
        e.set_particles(arr([c[0] for c in cols]), arr([c[1] for c in cols]), np.zeros(n, np.float32), np.zeros(n, np.float32),
                        arr([c[2] for c in cols]))
        gx, gy, _ = e.forces(float(np.array([theta_bits], np.uint32).view(np.float32)[0]))
        for i in range(n):
            assert _same(int(gx[i:i + 1].view(np.uint32)[0]), fx.to_bits(want[i][0])), (case, i, "fx")
            assert _same(int(gy[i:i + 1].view(np.uint32)[0]), fx.to_bits(want[i][1])), (case, i, "fy")
        checked += 1
    assert checked >= 35
