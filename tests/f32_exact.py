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


# ---- Barnes-Hut pieces (rs-src/nbody.rs:203-377) in the same exact model -----------------------------------------------------
def sqrt(a):
    """correctly rounded binary32 square root of an exact non-negative rational (Rust's f32::sqrt is IEEE sqrt)"""
    import math

    if a == 0:
        return Fraction(0)
    assert a > 0
    # scale so that the integer square root carries > 60 significant bits, then round the (inexact) root: a tie is impossible
    # unless the root is exact, because the square of a 25-bit midpoint would need 50 bits while `a` is a 24-bit value
    e = a.numerator.bit_length() - a.denominator.bit_length()
    shift = 140 - e
    shift += shift & 1                                     # even, so that sqrt(2^shift) is a power of two
    scaled = a * Fraction(2) ** shift
    root = math.isqrt(scaled.numerator // scaled.denominator)
    exact = root * root == scaled                           # (scaled is an integer when exact; harmless otherwise)
    approx = Fraction(root, 1) / Fraction(2) ** (shift // 2)
    if exact:
        return rn(approx)
    # the true root lies in (approx, approx + 2^-(shift/2)): 70 bits below the leading one, far inside any rounding interval
    lo, hi = rn(approx), rn(approx + Fraction(1, 2 ** (shift // 2)))
    assert lo == hi, "root too close to a rounding boundary for this model"
    return lo


HALF = Fraction(1, 2)


class XNode:
    """struct Node (nbody.rs:206-214) with exact-model fields"""

    def __init__(self, x1, y1, x2, y2):
        self.x1, self.y1, self.x2, self.y2 = x1, y1, x2, y2
        self.px = self.py = self.m = Fraction(0)
        self.children = None

    def add_mass(self, px, py, m):                          # :303-320
        assert m > 0
        if self.m == 0:
            self.px, self.py, self.m = px, py, m
            return
        inv = div(Fraction(1), add(self.m, m))              # 1.0 / (self.m + m)
        self.px = mul(add(mul(self.px, self.m), mul(px, m)), inv)
        self.py = mul(add(mul(self.py, self.m), mul(py, m)), inv)
        self.m = add(self.m, m)

    def centre(self):
        return mul(add(self.x1, self.x2), HALF), mul(add(self.y1, self.y2), HALF)   # (x1 + x2) * 0.5

    def quadrant(self, x, y):                               # :322-331 ; [UL, UR, LL, LR]
        cx, cy = self.centre()
        if y < cy:
            return 2 if x < cx else 3
        return 0 if x < cx else 1

    def create_children(self):                              # :286-301
        cx, cy = self.centre()
        self.children = [XNode(self.x1, cy, cx, self.y2), XNode(cx, cy, self.x2, self.y2),
                         XNode(self.x1, self.y1, cx, cy), XNode(cx, self.y1, self.x2, cy)]

    def insert(self, px, py, m, depth):                     # :226-284
        assert depth <= 50
        if self.children is not None:
            self.add_mass(px, py, m)
            self.children[self.quadrant(px, py)].insert(px, py, m, depth + 1)
            return
        too_close = abs(rn(self.px - px)) < EPS and abs(rn(self.py - py)) < EPS
        if self.m == 0 or too_close:
            self.add_mass(px, py, m)
            return
        po, qo, mo = self.px, self.py, self.m
        self.px = self.py = self.m = Fraction(0)
        self.create_children()
        self.insert(po, qo, mo, depth + 1)
        self.insert(px, py, m, depth + 1)

    def compute_force(self, px, py, m, theta):              # :333-377
        if self.children is not None:
            s = sub(self.x2, self.x1)
            dx, dy = sub(self.px, px), sub(self.py, py)
            d = sqrt(add(mul(dx, dx), mul(dy, dy)))
            if d != 0 and div(s, d) < theta:                # s / 0 = +inf: never accepted
                return force(px, py, m, self.px, self.py, self.m)
            fx = fy = Fraction(0)
            for c in self.children:
                ax, ay = c.compute_force(px, py, m, theta)
                fx, fy = add(fx, ax), add(fy, ay)
            return fx, fy
        if (self.px == px and self.py == py) or self.m == 0:
            return Fraction(0), Fraction(0)
        return force(px, py, m, self.px, self.py, self.m)


def bh_forces(bodies, theta):
    """forces of nb_step_barnes_hut's traversal on a list of [px, py, vx, vy, m] (exact binary32 values); root box = min/max"""
    xs, ys = [b[0] for b in bodies], [b[1] for b in bodies]
    root = XNode(min(xs), min(ys), max(xs), max(ys))        # :388-410
    for b in bodies:
        root.insert(b[0], b[1], b[4], 0)
    return [root.compute_force(b[0], b[1], b[4], theta) for b in bodies], root


# ---- nb_draw's viewport transform and colour helpers (rs-src/nbody.rs:494-506, :525-537, :585-617) -------------------------------
def trunc_i32(q):
    """`as i32` of a finite in-range value: toward zero"""
    n = abs(q.numerator) // q.denominator
    return -n if q < 0 else n


def draw_pixel(px, py, w, h):
    """(x as i32, y as i32) of a body at (px, py) -- every f32 operation of nbody.rs:494-506 and :525-537 rounded once"""
    W, H = Fraction(w), Fraction(h)                              # `w as f32`: exact for the sizes used
    vp_wdh, org = Fraction(100), Fraction(0)
    aspect = div(H, W)
    half = div(vp_wdh, Fraction(2))
    x1 = sub(org, half)
    y1 = mul(sub(org, half), aspect)
    x2 = add(org, half)
    y2 = mul(add(org, half), aspect)
    scalex = mul(div(Fraction(1), sub(x2, x1)), W)
    scaley = mul(div(Fraction(1), sub(y2, y1)), H)
    return trunc_i32(mul(sub(px, x1), scalex)), trunc_i32(mul(sub(py, y1), scaley))


def rgb_to_abgr32(r, g, b, factor):
    """nbody.rs:585-593 ; factor an exact binary32 value"""
    ch = [min(255, trunc_i32(mul(Fraction(c), factor))) for c in (r, g, b)]
    return ch[0] | (ch[2] << 16) | (ch[1] << 8)


def add_abgr32(c1, c2):
    """nbody.rs:595-617"""
    out = 0
    for sh in (24, 16, 8, 0):
        out |= min(255, ((c1 >> sh) & 0xFF) + ((c2 >> sh) & 0xFF)) << sh
    return out
