"""IEEE-754 binary32 arithmetic evaluated EXACTLY with fractions.Fraction and an explicit round-to-nearest-even
per operation -- no C float, no numpy float on the way.  Test infrastructure: a third, arithmetic-level pin for the
oracle (and through it for the GPU's bit-exact mode), independent of any compiler's float code generation.

Restates, operation by operation, rs-src/nbody.rs:174-183 (force) and :153-160 (kick-drift); rustc evaluates every
f32 operation separately with round-to-nearest-even and never contracts a*b+c (no fast-math in the crate).
"""
import struct
from fractions import Fraction

EMIN, MANT = -126, 23
F32_MAX = Fraction((1 << 24) - 1) * Fraction(2) ** (127 - 23)


def from_bits(u):
    """binary32 bit pattern -> exact Fraction (finite values only)"""
    s, e, m = u >> 31, (u >> 23) & 0xFF, u & 0x7FFFFF
    assert e != 0xFF, "inf/nan have no exact value"
    v = Fraction(m, 1 << 23) * Fraction(2) ** EMIN if e == 0 else (1 + Fraction(m, 1 << 23)) * Fraction(2) ** (e - 127)
    return -v if s else v


def rn(q):
    """round an exact rational to the nearest binary32 value, ties to even (subnormals included)"""
    if q == 0:
        return Fraction(0)
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    assert Fraction(2) ** e <= a < Fraction(2) ** (e + 1)
    quantum = Fraction(2) ** (max(e, EMIN) - MANT)
    k = a / quantum
    lo = k.numerator // k.denominator
    rem = k - lo
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (lo & 1)):
        lo += 1
    r = lo * quantum
    assert r <= F32_MAX, "overflow to infinity: outside this model"
    return -r if q < 0 else r


def to_bits(v, negative_zero=False):
    """exact Fraction that IS a binary32 value -> its bit pattern"""
    if v == 0:
        return 0x80000000 if negative_zero else 0
    s = 1 if v < 0 else 0
    a = abs(v)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    if e < EMIN:
        m = a / Fraction(2) ** (EMIN - MANT)
        assert m.denominator == 1 and m < (1 << 23)
        return (s << 31) | int(m)
    m = (a / Fraction(2) ** e - 1) * (1 << 23)
    assert m.denominator == 1, "not a binary32 value"
    return (s << 31) | ((e + 127) << 23) | int(m)


def bits_of_float(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def float_of_bits(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


EPS = from_bits(bits_of_float(0.0001))     # nbody.rs:17  `static EPS : f32 = 0.0001`


def add(a, b):
    return rn(a + b)


def sub(a, b):
    return rn(a - b)


def mul(a, b):
    return rn(a * b)


def div(a, b):
    return rn(a / b)


def force(px1, py1, m1, px2, py2, m2):
    """nbody.rs:174-183, one rounding per operation, in source order"""
    dx = sub(px2, px1)
    dy = sub(py2, py1)
    dist_sq = add(mul(dx, dx), mul(dy, dy))
    f = div(mul(m1, m2), add(dist_sq, EPS))
    return mul(f, dx), mul(f, dy)


def brute_step(bodies, dt):
    """nbody.rs:132-160 on a list of [px, py, vx, vy, m] (exact Fractions that are binary32 values)"""
    n = len(bodies)
    forces = []
    for i in range(n):
        fx = fy = Fraction(0)            # Force { fx: 0.0, fy: 0.0 }
        for j in range(n):
            if i == j:
                continue
            ax, ay = force(bodies[i][0], bodies[i][1], bodies[i][4], bodies[j][0], bodies[j][1], bodies[j][4])
            fx, fy = add(fx, ax), add(fy, ay)
        forces.append((fx, fy))
    out = []
    for (px, py, vx, vy, m), (fx, fy) in zip(bodies, forces):
        vx = add(vx, div(mul(dt, fx), m))     # `dt * forces[i].fx / particles[i].m` parses as (dt*fx)/m
        vy = add(vy, div(mul(dt, fy), m))
        px = add(px, mul(dt, vx))             # with the NEW velocity
        py = add(py, mul(dt, vy))
        out.append([px, py, vx, vy, m])
    return out
