"""Arithmetic-level known-answer tests: the oracle's force() and kick-drift against an EXACT
evaluation with fractions.Fraction + explicit round-to-nearest-even to binary32 after every operation (tests/f32_exact.py).
Inputs are binary32 BIT PATTERNS; nothing on the checking side goes through C or numpy floating point, so this pins the
oracle (oracle/nbody_oracle.c, built by gcc) to IEEE-754 semantics of the reference's source expressions
(rs-src/nbody.rs:174-183, :132-160) independently of compiler flags, contraction or x87/SSE code generation.
The GPU's bit-exact mode is held to the same table in tests/test_gpu_exact_arithmetic.py."""
import random
from fractions import Fraction

import numpy as np
import pytest

import f32_exact as fx


def _b(x):
    return fx.bits_of_float(x)


# hand-picked (px1, py1, m1, px2, py2, m2) -- as Python floats that are exactly binary32 after the struct round trip
HAND = [
    (0.0, 0.0, 1.0, 1.0, 0.0, 1.0),                     # unit separation
    (0.0, 0.0, 1000.0, 0.5, 0.0, 1.0),                  # sun - inner planet (nb_stable_orbits scales)
    (0.0, 0.0, 1000.0, 30.0, 0.0, 1.0),
    (1.0, 2.0, 3.0, 1.0, 2.0, 4.0),                     # coincident: d = 0, f * 0 = 0
    (1.0, 1.0, 1.0, 1.0000001, 1.0, 1.0),               # one ulp apart in x
    (1.0, 1.0, 1.0, 1.0, 0.99999994, 1.0),              # one ulp apart in y, other side
    (-23.0, 17.5, 0.1, 22.999998, -17.499998, 1.4999999),   # nb_random_disk extremes
    (49.999996, -49.999996, 1.0, -49.999996, 49.999996, 1.0),   # corners of the viewport
    (0.01, 0.0, 1.0, 0.0, 0.0, 1.0),                    # d^2 = EPS: the softening length
    (0.0, 0.0, 1.0, 0.0070710676, 0.0070710676, 1.0),   # d^2 ~ EPS on the diagonal
    (3.0, 4.0, 2.0, 0.0, 0.0, 5.0),                     # 3-4-5
    (1e-3, 1e-3, 1e-3, -1e-3, -1e-3, 1e-3),
    (12345.678, -9876.543, 1.0, 12345.679, -9876.542, 1.0),    # far from the origin: few bits of separation
    (1.0e-20, 0.0, 1.0, 0.0, 0.0, 1.0),                 # dx*dx underflows to a subnormal / zero
    (0.0, 0.0, 1.0e-30, 1.0, 1.0, 1.0e-10),             # m1*m2 underflows into the subnormals
    (0.0, 0.0, 1.0e15, 1.0e10, 1.0e10, 1.0e15),         # large but finite everywhere
    (0.1, 0.2, 0.3, 0.4, 0.5, 0.6),                     # "decimal" inputs: every operation rounds
    (1.5, -2.25, 1.0, -3.75, 4.125, 8.0),               # dyadic inputs: many operations exact
    (16777216.0, 0.0, 1.0, 16777218.0, 0.0, 1.0),       # 2^24 neighbourhood
    (0.33333334, 0.6666667, 1.0, 0.6666667, 0.33333334, 1.0),
]


def _random_cases(count, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        vals = []
        for k in range(6):
            mass = k in (2, 5)
            e = rng.randint(-8, 6) if not mass else rng.randint(-10, 10)      # 2^-8 .. 2^6 coordinates, 2^-10 .. 2^10 masses
            u = (rng.getrandbits(1) << 31 if not mass else 0) | ((e + 127) << 23) | rng.getrandbits(23)
            vals.append(u)
        out.append(tuple(vals))
    # close pairs: second body a few ulps from the first
    for _ in range(count // 4):
        e = rng.randint(-4, 5)
        x = ((e + 127) << 23) | rng.getrandbits(23)
        y = (1 << 31) | ((rng.randint(-4, 5) + 127) << 23) | rng.getrandbits(23)
        out.append((x, y, _b(1.0), x + rng.randint(-40, 40), y + rng.randint(-40, 40), _b(rng.choice([1.0, 0.125, 1000.0]))))
    return out


CASES = [tuple(_b(v) for v in c) for c in HAND] + _random_cases(80, 20260928)


def _same(a_bits, b_bits):
    """equal bit patterns; +0 and -0 are the same number (the model carries no sign of zero)"""
    return a_bits == b_bits or ((a_bits | b_bits) & 0x7FFFFFFF) == 0


def test_rounding_model_itself():
    # exact values round to themselves; ties go to even; subnormals and the smallest normal behave
    for u in (0x3F800000, 0x3F800001, 0x00000001, 0x007FFFFF, 0x00800000, 0x7F7FFFFF, 0xBF000000, 0x00000000):
        assert fx.to_bits(fx.rn(fx.from_bits(u))) == u
    one, ulp = fx.from_bits(0x3F800000), Fraction(1, 1 << 23)
    assert fx.to_bits(fx.rn(one + ulp / 2)) == 0x3F800000           # tie -> even (mantissa 0)
    assert fx.to_bits(fx.rn(one + 3 * ulp / 2)) == 0x3F800002       # tie -> even (mantissa 2)
    assert fx.to_bits(fx.rn(one + ulp / 2 + Fraction(1, 1 << 60))) == 0x3F800001
    tiny = Fraction(1, 1 << 149)
    assert fx.to_bits(fx.rn(tiny / 2)) == 0 and fx.to_bits(fx.rn(3 * tiny / 2)) == 2 and fx.to_bits(fx.rn(tiny * Fraction(3, 4))) == 1
    assert fx.to_bits(fx.EPS) == 0x38D1B717                          # 0.0001f


def test_force_matches_exact_binary32_arithmetic(ob):
    """orc_force == Fraction model, bit for bit, on 120 pairs (both orders: the law is antisymmetric bit for bit)."""
    assert len(CASES) >= 100
    for c in CASES:
        for (a, b) in ((c[:3], c[3:]), (c[3:], c[:3])):
            ex, ey = fx.force(*[fx.from_bits(u) for u in a + b])
            gx, gy = ob.force(*[fx.float_of_bits(u) for u in a + b])
            assert _same(fx.bits_of_float(float(gx)), fx.to_bits(ex)), (c, hex(fx.to_bits(ex)), float(gx))
            assert _same(fx.bits_of_float(float(gy)), fx.to_bits(ey)), (c, hex(fx.to_bits(ey)), float(gy))


@pytest.mark.parametrize("dt_bits", [0x3C23D70A, 0x3BA3D70A, 0x3CA3D70A])   # 0.01 (hs:45), 0.005, 0.02 (the caller's x2 / /2 keys)
def test_kick_drift_matches_exact_binary32_arithmetic(ob, dt_bits):
    """One nb_step_brute_force on 2-body and 3-body systems built from the table: force, ascending-j sum from +0.0,
    (dt*F)/m, drift with the NEW velocity -- every operation rounded once, in source order (nbody.rs:132-160)."""
    dt = fx.from_bits(dt_bits)
    rng = random.Random(7)
    for k, c in enumerate(CASES):
        third = CASES[(k * 7 + 3) % len(CASES)]
        for nb in (2, 3):
            rows = [c[:3], c[3:]] + ([third[:3]] if nb == 3 else [])
            vel = [((rng.randint(-3, 3) + 127) << 23 | rng.getrandbits(23) | (rng.getrandbits(1) << 31),
                    (rng.randint(-3, 3) + 127) << 23 | rng.getrandbits(23) | (rng.getrandbits(1) << 31)) for _ in rows]
            bodies = [[fx.from_bits(r[0]), fx.from_bits(r[1]), fx.from_bits(v[0]), fx.from_bits(v[1]), fx.from_bits(r[2])]
                      for r, v in zip(rows, vel)]
            try:
                want = fx.brute_step(bodies, dt)
            except (AssertionError, ZeroDivisionError):
                continue    # overflow / zero mass: outside the model (and outside anything the caller sends)
            p = ob.particles([fx.float_of_bits(r[0]) for r in rows], [fx.float_of_bits(r[1]) for r in rows],
                             [fx.float_of_bits(v[0]) for v in vel], [fx.float_of_bits(v[1]) for v in vel],
                             [fx.float_of_bits(r[2]) for r in rows])
            ob.step_brute_force(p, fx.float_of_bits(dt_bits))
            for i in range(nb):
                got = [int(np.asarray(p[f][i]).view(np.uint32)) for f in ("px", "py", "vx", "vy")]
                exp = [fx.to_bits(want[i][j]) for j in range(4)]
                assert all(_same(g, e) for g, e in zip(got, exp)), (k, nb, i, [hex(x) for x in got], [hex(x) for x in exp])


def test_sqrt_model():
    for u in (0x3F800000, 0x40000000, 0x40490FDB, 0x00000001, 0x007FFFFF, 0x7F7FFFFF, 0x3A83126F, 0x411FFFFF):
        got = fx.to_bits(fx.sqrt(fx.from_bits(u)))
        want = fx.bits_of_float(float(np.sqrt(np.float32(fx.float_of_bits(u)))))      # numpy's f32 sqrt is IEEE: a cross-check of the model only
        assert got == want, (hex(u), hex(got), hex(want))


def bh_cases():
    """40 small systems as (columns of bit patterns [x, y, m], theta bits); some with a sub-EPS pair and an exact duplicate"""
    rng = random.Random(11)
    out = []
    for case in range(40):
        n = rng.choice([2, 3, 5, 9, 17, 30])
        cols = []
        for k in range(n):
            x = ((rng.randint(-3, 4) + 127) << 23) | rng.getrandbits(23) | (rng.getrandbits(1) << 31)
            y = ((rng.randint(-3, 4) + 127) << 23) | rng.getrandbits(23) | (rng.getrandbits(1) << 31)
            m = ((rng.randint(-4, 4) + 127) << 23) | rng.getrandbits(23)
            cols.append([x, y, m])
        if case % 4 == 1 and n > 3:                 # a pair closer than EPS (merged) and an exact duplicate
            cols[1] = [cols[0][0] + 3, cols[0][1] - 2, cols[1][2]]
            cols[2] = list(cols[0])
        out.append((cols, rng.choice([0x3F000000, 0x3F59999A, 0x3E99999A, 0x3F733333])))    # theta 0.5, 0.85, 0.3, 0.95
    return out


def bh_model(cols, theta_bits):
    bodies = [[fx.from_bits(c[0]), fx.from_bits(c[1]), Fraction(0), Fraction(0), fx.from_bits(c[2])] for c in cols]
    return fx.bh_forces(bodies, fx.from_bits(theta_bits))


def test_barnes_hut_tree_and_traversal_match_exact_binary32_arithmetic(ob):
    """The oracle's quadtree (sequential insert, f32 running fold of add_mass, (x1+x2)*0.5 midpoints, EPS merge) and its
    traversal (sqrt, s/d < theta, hierarchical sums) against the exact Fraction model, bit for bit: forces of every body and the
    root node's folded mass / centre, on 40 small random systems incl. sub-EPS pairs and duplicates (nbody.rs:203-377)."""
    checked = 0
    for case, (cols, theta_bits) in enumerate(bh_cases()):
        n = len(cols)
        try:
            want, root = bh_model(cols, theta_bits)
        except AssertionError:
            continue                                 # depth > 50 / rounding-boundary root: outside the model
        p = ob.particles([fx.float_of_bits(c[0]) for c in cols], [fx.float_of_bits(c[1]) for c in cols], np.zeros(n), np.zeros(n),
                         [fx.float_of_bits(c[2]) for c in cols])
        rc, gx, gy = ob.bh_forces(p, fx.float_of_bits(theta_bits))
        assert rc == 0
        for i in range(n):
            assert _same(int(gx[i:i + 1].view(np.uint32)[0]), fx.to_bits(want[i][0])), (case, i, "fx")
            assert _same(int(gy[i:i + 1].view(np.uint32)[0]), fx.to_bits(want[i][1])), (case, i, "fy")
        rc, st = ob.bh_tree_stats(p)
        assert rc == 0 and _same(fx.bits_of_float(st["m"]), fx.to_bits(root.m)) and _same(fx.bits_of_float(st["px"]), fx.to_bits(root.px))
        checked += 1
    assert checked >= 35


def test_draw_viewport_transform_and_colours_match_exact_arithmetic(ob):
    """nb_draw's f32 viewport transform (aspect, origin, scale: nbody.rs:494-506), the truncating pixel cast (:536-537) and the
    colour helpers (:585-617) against the exact model: a single body with velocity (+1, 0) lights its own pixel with the body
    colour and the pixel to its left with the tail colour (octant 0, :541-554) -- compared for 300 positions across the
    viewport and five framebuffer shapes, including positions an ulp either side of a pixel boundary."""
    col_body = fx.rgb_to_abgr32(255, 215, 130, fx.from_bits(fx.bits_of_float(0.3)))
    col_tail = fx.rgb_to_abgr32(255, 215, 130, fx.from_bits(fx.bits_of_float(0.25)))
    assert (col_body, col_tail) == (0x0027404C, 0x0020353F)
    assert ob.lib().orc_rgb_to_abgr32(255, 215, 130, 0.3) == col_body and ob.lib().orc_add_abgr32(0x00F0F0F0, col_body) == fx.add_abgr32(0x00F0F0F0, col_body)
    rng = random.Random(3)
    for (w, h) in ((512, 512), (640, 480), (101, 37), (64, 200), (3, 3)):
        for k in range(60):
            if k % 3 == 0:       # an ulp around a pixel boundary: boundary x_b = j * 100 / w - 50
                j = rng.randint(1, w - 1)
                xb = fx.bits_of_float(j * 100.0 / w - 50.0)
                xbits = xb + rng.randint(-2, 2)
            else:
                xbits = fx.bits_of_float(rng.uniform(-49.5, 49.5))
            ybits = fx.bits_of_float(rng.uniform(-49.5, 49.5) * h / w)
            xi, yi = fx.draw_pixel(fx.from_bits(xbits), fx.from_bits(ybits), w, h)
            p = ob.particles([fx.float_of_bits(xbits)], [fx.float_of_bits(ybits)], [1.0], [0.0], [1.0])
            fb = ob.draw(p, w, h)
            want = np.zeros((h, w), np.uint32)
            if 0 <= xi < w and 0 <= yi < h:
                want[yi, xi] = fx.add_abgr32(int(want[yi, xi]), col_body)
            if 0 <= xi - 1 < w and 0 <= yi < h:
                want[yi, xi - 1] = fx.add_abgr32(int(want[yi, xi - 1]), col_tail)
            cx, cy = w // 2, h // 2
            for (a, b) in ((cx, cy), (cx + 1, cy), (cx, cy + 1), (cx - 1, cy), (cx, cy - 1)):
                want[b, a] = 0x00FF00FF
            assert np.array_equal(fb, want), (w, h, hex(xbits), hex(ybits), xi, yi)
