"""Synthetic initial conditions for benches and tests (NOT in the reference, which only has
nb_random_disk / nb_stable_orbits, nbody.rs:39-104).  Seeded with splitmix64 -> top 24 bits ->
[0,1) f32, the same construction the library's presets use, so every rank / language produces
identical bytes.  Definitions follow SURVEY.md section 8(d).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_uniform(seed, count):
    """`count` f32 samples in [0,1): sample k = top 24 bits of splitmix64 output k."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def plummer_sphere(n, seed=0x5EED0001, a=5.0, rmax=45.0, total_mass=1000.0, dim=3):
    """Plummer sphere: r = a / sqrt(u^(-2/3) - 1) clipped to rmax (inside the +-55 kill box),
    isotropic direction, zero velocities (throughput is state independent), m = total_mass / n.
    dim=2 keeps the same x,y and sets z = 0 (reference-surface parity runs).
    Returns dict of float32 arrays px,py,pz,vx,vy,vz,m."""
    u = splitmix64_uniform(seed, 3 * n).astype(np.float64).reshape(3, n)
    u0 = np.clip(u[0], 1e-7, 1.0 - 1e-7)
    r = a / np.sqrt(u0 ** (-2.0 / 3.0) - 1.0)
    r = np.minimum(r, rmax)
    cos_t = 2.0 * u[1] - 1.0
    sin_t = np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t))
    phi = 2.0 * np.pi * u[2]
    out = {
        "px": (r * sin_t * np.cos(phi)).astype(np.float32),
        "py": (r * sin_t * np.sin(phi)).astype(np.float32),
        "pz": (r * cos_t).astype(np.float32) if dim == 3 else np.zeros(n, np.float32),
        "vx": np.zeros(n, np.float32),
        "vy": np.zeros(n, np.float32),
        "vz": np.zeros(n, np.float32),
        "m": np.full(n, total_mass / max(n, 1), np.float32),
    }
    return out


def two_galaxies(n, seed=0x5EED0002, rmin=0.5, rmax=12.0):
    """Two nb_stable_orbits-style disks (n/2 bodies each: a 1000-mass core + unit planets on
    circular orbits, nbody.rs:85-102), centres (+-15, 0), bulk velocities (-+3, +-1). 2-D.
    float64 arithmetic on the f32 samples, ONE rounding to f32 per stored value: the same definition as the
    library's nbx_two_galaxies (c_api.cpp / host_ops.cpp), bit for bit (tests/test_workload_generators.py).
    (Round 2 evaluated this in numpy float32, whose SIMD sin/cos are not reproducible outside numpy.)"""
    half = n // 2
    u = splitmix64_uniform(seed, 2 * n).astype(np.float64).reshape(2, n)
    g = (np.arange(n) >= half)
    cx = np.where(g, 15.0, -15.0)
    cvx = np.where(g, -3.0, 3.0)
    cvy = np.where(g, 1.0, -1.0)
    speed = np.sqrt(1000.0)
    r = (rmax - rmin) * u[0] + rmin
    th = 2.0 * np.pi * u[1]
    c, s = np.cos(th), np.sin(th)
    px = (cx + r * c).astype(np.float32)
    py = (r * s).astype(np.float32)
    vx = (cvx - speed * s).astype(np.float32)
    vy = (cvy + speed * c).astype(np.float32)
    m = np.ones(n, np.float32)
    for lo in ((0, half) if n > 1 else (0,) if n == 1 else ()):
        if lo < n:
            px[lo], py[lo], vx[lo], vy[lo], m[lo] = cx[lo], 0.0, cvx[lo], cvy[lo], 1000.0
    z = np.zeros(n, np.float32)
    return {"px": px, "py": py, "pz": z, "vx": vx, "vy": vy, "vz": z.copy(), "m": m}
