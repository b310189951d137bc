"""TEST INFRASTRUCTURE ONLY -- an independent restatement of the reference's presets (rs-src/nbody.rs:39-104) in plain
Python with numpy.float32 scalars.  The reference draws from rand 0.3.14's OS-seeded thread_rng, so preset VALUES are
unpinnable; what is restated is everything around the bit source: rand 0.3's f32 construction (top 24 bits * 2^-24),
Range::ind_sample = lo + (hi - lo) * u, the draw order and the arithmetic of the two presets.  The bit source is the
same splitmix64 the C oracle and the library use, so all three can be compared bit for bit.  cos / sin / sqrt are libm's
cosf / sinf / sqrtf (what Rust's f32 methods call), reached through ctypes."""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("cosf", "sinf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
PI = F(3.14159274101257324)
M64 = (1 << 64) - 1


class Rng:
    def __init__(self, seed):
        self.s = seed & M64

    def next_u64(self):                      # splitmix64
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def next_f32(self):                      # rand 0.3 Rng::next_f32: 24 random bits scaled into [0, 1)
        return F(self.next_u64() >> 40) * F(1.0 / 16777216.0)

    def range(self, lo, hi):                 # Range::new(lo, hi).ind_sample
        return F(lo) + (F(hi) - F(lo)) * self.next_f32()


def cosf(x): return F(_libm.cosf(float(x)))
def sinf(x): return F(_libm.sinf(float(x)))
def sqrtf(x): return F(_libm.sqrtf(float(x)))


def random_disk(n, seed):                    # nbody.rs:39-71
    rng = Rng(seed)
    out = []
    for _ in range(max(n, 0)):
        x = rng.range(0.0, 1.0)
        y = rng.range(0.0, 1.0)
        r = sqrtf(x)                         # uniform_sample_disk
        theta = F(2.0) * PI * y
        x = r * cosf(theta)
        y = r * sinf(theta)
        x = x * F(23.0)
        y = y * F(23.0)
        vx = rng.range(-3.5, 3.5)
        vy = rng.range(-3.5, 3.5)
        m = rng.range(0.1, 1.5)
        out.append((x, y, vx, vy, m))
    return np.array(out, dtype=np.float32).reshape(-1, 5)


def stable_orbits(n, rmin, rmax, seed):      # nbody.rs:73-104
    rng = Rng(seed)
    rmin = F(rmin); rmax = F(rmax)
    speed = sqrtf(F(1.0) * F(1000.0))
    out = [(F(0.0), F(0.0), F(0.0), F(0.0), F(1000.0))]
    for _ in range(n - 1):
        r = (rmax - rmin) * rng.range(0.0, 1.0) + rmin
        theta = F(2.0) * PI * rng.range(0.0, 1.0)
        out.append((r * cosf(theta), r * sinf(theta), -speed * sinf(theta), speed * cosf(theta), F(1.0)))
    return np.array(out, dtype=np.float32).reshape(-1, 5)
