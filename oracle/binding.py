"""ctypes binding of oracle/libnbody_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product path in rust-exp_amd/ never does; it fails loudly without the HIP library).
Function-by-function reference citations live in nbody_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnbody_oracle.so")

# AoS record of the reference: struct Particle {px,py,vx,vy,m} (nbody.rs:19-26)
PARTICLE = np.dtype([("px", "<f4"), ("py", "<f4"), ("vx", "<f4"), ("vy", "<f4"), ("m", "<f4")])

ORC_OK = 0
ORC_PANIC_DEPTH = -1
ORC_PANIC_SAME_POS = -2
ORC_PANIC_SUBDIVIDE = -3
ORC_PANIC_MASS = -4
ORC_PANIC_NTHREADS = -5


def build(force=False):
    import fcntl

    src = os.path.join(_HERE, "nbody_oracle.c")
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:   # several test processes may build at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "-s"])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p = C.POINTER(C.c_float)
        f64p = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.orc_force.argtypes = [C.c_float] * 6 + [f32p, f32p]
        L.orc_force.restype = None
        L.orc_brute_forces.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_brute_forces.restype = None
        L.orc_brute_forces_mt.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_brute_forces_mt.restype = C.c_int
        L.orc_brute_forces_f64.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_brute_forces_f64.restype = None
        L.orc_step_brute_force.argtypes = [vp, C.c_int, C.c_float]
        L.orc_step_brute_force.restype = C.c_int
        L.orc_step_brute_force_mt.argtypes = [vp, C.c_int, C.c_float, C.c_int]
        L.orc_step_brute_force_mt.restype = C.c_int
        L.orc_step_barnes_hut.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_step_barnes_hut.restype = C.c_int
        L.orc_bh_forces.argtypes = [vp, C.c_int, C.c_float, C.c_int, vp, vp]
        L.orc_bh_forces.restype = C.c_int
        L.orc_bh_forces_exact.argtypes = [vp, C.c_int, C.c_float, C.c_int, vp, vp]
        L.orc_bh_forces_exact.restype = C.c_int
        L.orc_bh_tree_stats.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 3 + [f32p] * 3
        L.orc_bh_tree_stats.restype = C.c_int
        L.orc_bh_tree_dump.argtypes = [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
        L.orc_bh_tree_dump.restype = C.c_int
        L.orc_next_f32.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_next_f32.restype = C.c_float
        L.orc_random_disk.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_random_disk.restype = C.c_int
        L.orc_stable_orbits.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64)]
        L.orc_stable_orbits.restype = C.c_int
        L.orc_rgb_to_abgr32.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_float]
        L.orc_rgb_to_abgr32.restype = C.c_uint32
        L.orc_add_abgr32.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_add_abgr32.restype = C.c_uint32
        L.orc_draw.argtypes = [vp, C.c_int, C.c_int32, C.c_int32, vp]
        L.orc_draw.restype = None
        L.orc_sizeof_particle.restype = C.c_int
        assert L.orc_sizeof_particle() == PARTICLE.itemsize
        _lib = L
        _ = f64p
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def particles(px, py, vx, vy, m):
    n = len(px)
    p = np.zeros(n, dtype=PARTICLE)
    p["px"], p["py"], p["vx"], p["vy"], p["m"] = px, py, vx, vy, m
    return p


def force(px1, py1, m1, px2, py2, m2):
    fx, fy = C.c_float(), C.c_float()
    lib().orc_force(px1, py1, m1, px2, py2, m2, C.byref(fx), C.byref(fy))
    return np.float32(fx.value), np.float32(fy.value)


def brute_forces(p, i0=0, i1=None, nthreads=1):
    n = len(p)
    i1 = n if i1 is None else i1
    fx = np.zeros(i1 - i0, np.float32)
    fy = np.zeros(i1 - i0, np.float32)
    if nthreads > 1 and i0 == 0:
        rc = lib().orc_brute_forces_mt(_ptr(p), n, i1, nthreads, _ptr(fx), _ptr(fy))
        assert rc == 0
    else:
        lib().orc_brute_forces(_ptr(p), n, i0, i1, _ptr(fx), _ptr(fy))
    return fx, fy


def brute_forces_f64(p, i0=0, i1=None):
    n = len(p)
    i1 = n if i1 is None else i1
    fx = np.zeros(i1 - i0, np.float64)
    fy = np.zeros(i1 - i0, np.float64)
    lib().orc_brute_forces_f64(_ptr(p), n, i0, i1, _ptr(fx), _ptr(fy))
    return fx, fy


def step_brute_force(p, dt, nthreads=1):
    """In place on the PARTICLE array p."""
    if nthreads > 1:
        return lib().orc_step_brute_force_mt(_ptr(p), len(p), dt, nthreads)
    return lib().orc_step_brute_force(_ptr(p), len(p), dt)


def step_barnes_hut(p, theta, dt, nthreads=1):
    return lib().orc_step_barnes_hut(_ptr(p), len(p), theta, dt, nthreads)


def bh_forces(p, theta, nthreads=1):
    fx = np.zeros(len(p), np.float32)
    fy = np.zeros(len(p), np.float32)
    rc = lib().orc_bh_forces(_ptr(p), len(p), theta, nthreads, _ptr(fx), _ptr(fy))
    return rc, fx, fy


def bh_forces_exact(p, theta, nthreads=1):
    """fp64 arbiter (not in the reference): the reference's tree and opening law with exact node masses / centres and fp64
    arithmetic throughout (nbody_oracle.c, orc_bh_forces_exact)."""
    fx = np.zeros(len(p), np.float64)
    fy = np.zeros(len(p), np.float64)
    rc = lib().orc_bh_forces_exact(_ptr(p), len(p), theta, nthreads, _ptr(fx), _ptr(fy))
    return rc, fx, fy


def bh_tree_stats(p):
    nodes, leaves, depth = C.c_int(), C.c_int(), C.c_int()
    m, x, y = C.c_float(), C.c_float(), C.c_float()
    rc = lib().orc_bh_tree_stats(_ptr(p), len(p), C.byref(nodes), C.byref(leaves), C.byref(depth),
                                 C.byref(m), C.byref(x), C.byref(y))
    return rc, dict(nodes=nodes.value, leaves=leaves.value, depth=depth.value, m=m.value, px=x.value, py=y.value)


def bh_tree_dump(p):
    """Pre-order dump of the reference-faithful tree: rows of x1,y1,x2,y2,px,py,m,has_children."""
    cap = max(16, 16 * len(p))
    while True:
        out = np.zeros((cap, 8), np.float32)
        cnt = C.c_int()
        rc = lib().orc_bh_tree_dump(_ptr(p), len(p), _ptr(out), cap, C.byref(cnt))
        if rc != 0:
            return rc, None
        if cnt.value <= cap:
            return rc, out[: cnt.value].copy()
        cap = cnt.value


def random_disk(n, seed):
    s = C.c_uint64(seed)
    p = np.zeros(max(n, 0), dtype=PARTICLE)
    cnt = lib().orc_random_disk(_ptr(p), n, C.byref(s))
    return p[:cnt]


def stable_orbits(n, rmin, rmax, seed):
    s = C.c_uint64(seed)
    p = np.zeros(max(n, 1), dtype=PARTICLE)
    cnt = lib().orc_stable_orbits(_ptr(p), n, rmin, rmax, C.byref(s))
    return p[:cnt]


def draw(p, w, h):
    fb = np.zeros(w * h, np.uint32)
    lib().orc_draw(_ptr(p), len(p), w, h, _ptr(fb))
    return fb.reshape(h, w)
