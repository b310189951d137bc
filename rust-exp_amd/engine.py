"""ctypes binding of lib/libnbody_mi355x.so (C ABI: include/nbody_mi355x.h).

Mirrors the reference interface: the six `nb_*` functions keep the names and argument meaning of
rs-src/nbody.rs:34-35,:39-40,:73-74,:106-107,:186-187,:482-483 (as imported by
hs-src/RustNBodyExperiment.hs:101-106); `NBodyEngine` wraps the additive handle API.
No CPU fallback: a missing library raises at import, a missing GPU raises at the first step.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_SO = os.path.join(_PKG, "lib", "libnbody_mi355x.so")

NBX_OK = 0
NBX_ERR_INVALID = -1
NBX_ERR_NO_DEVICE = -2
NBX_ERR_HIP = -3
NBX_ERR_TREE_DEPTH = -4
NBX_ERR_TREE = -5
NBX_ERR_ALLOC = -6
NBX_ERR_STATE = -7

NBX_OPT_FORCE_MODE = 0
NBX_OPT_JSPLIT = 1
NBX_OPT_BODIES_PER_THREAD = 2
NBX_OPT_DIM = 3
NBX_OPT_PROFILE = 4
NBX_OPT_KERNEL_VARIANT = 5
NBX_OPT_SOURCE_PRECISION = 6
NBX_OPT_DRAW_DEVICE = 7
NBX_OPT_BH_TREE = 8
NBX_OPT_BH_WAVE = 9
NBX_OPT_STRICT_KERNEL = 13
NBX_OPT_BH_FOLD = 14
NBX_OPT_BH_ASYNC = 15
NBX_OPT_BH_WALK = 18
NBX_OPT_BH_FUSE_KICK = 20
# (10-12 and 17 became enum nbx_stat in round 5; 16 and 19 -- measured losers -- were removed)

NBX_STAT_BH_FALLBACKS = 0
NBX_STAT_BH_LAST_TREE = 1
NBX_STAT_DRAW_AMBIGUOUS = 2
NBX_STAT_BH_REFUSAL = 3
NBX_STAT_BH_CLASS_SWITCHES = 4
NBX_STAT_BH_COLD_RESORTS = 5
NBX_STAT_BH_CHAIN_MERGED = 6
NBX_STAT_BH_CHAIN_APPROX = 7

NBX_GROUP_INFO_EXCHANGE = 0
NBX_GROUP_INFO_RCCL_RANKS = 1
NBX_GROUP_INFO_ENQUEUE_THREADS = 2
NBX_GROUP_INFO_FP32_STALE = 3

NBX_K_FORCE = 0
NBX_K_INTEGRATE = 1
NBX_K_BH_EVAL = 2
NBX_K_EXCHANGE = 3
NBX_K_TREE_BUILD = 4


class NBodyError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"nbody_mi355x error {code}: {text}")
        self.code = code


class _DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 64),
        ("compute_units", C.c_int32),
        ("clock_khz", C.c_int32),
        ("wavefront_size", C.c_int32),
        ("lds_bytes_per_cu", C.c_int32),
        ("peak_fp32_flops", C.c_double),
        ("hbm_bytes", C.c_uint64),
    ]


def lib_path():
    return _SO


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU).
    Serialised with a file lock: the ranks of a multi-GPU launch may all get here at once."""
    import fcntl

    os.makedirs(os.path.join(_PKG, "lib"), exist_ok=True)
    with open(os.path.join(_PKG, "lib", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if not f.startswith(".")] + [
                os.path.join(_PKG, "..", "include", "nbody_mi355x.h")
            ]
            stale = not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
            if force or stale:
                have_hipcc = any(os.access(os.path.join(p, "hipcc"), os.X_OK)
                                 for p in os.environ.get("PATH", "").split(os.pathsep))
                if not have_hipcc:
                    if os.path.exists(_SO):
                        return _SO  # box without a compiler on PATH: use the prebuilt library that travelled with the tree
                    raise RuntimeError("hipcc not found and libnbody_mi355x.so is not built")
                subprocess.check_call(["make", "-C", _CSRC, "-s", "-j8"])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


_lib = None

_f32p = C.POINTER(C.c_float)


def lib():
    """Load the library (building it if needed). Raises if it cannot be loaded: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    build()
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 or os.environ.get("NBX_PRELOAD_TORCH"):
        # One process per GPU under torch.distributed.run: PyTorch's wheel bundles its own HIP runtime, which must be the
        # first one loaded (loaded second, torch reports "No HIP GPUs"). Import it here so that the order in which the
        # caller imports things does not matter; single-process hosts never import torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(_SO)
    E = C.c_void_p
    i32 = C.c_int32
    # level 1
    L.nb_num_particles.argtypes = []
    L.nb_num_particles.restype = i32
    L.nb_random_disk.argtypes = [i32]
    L.nb_random_disk.restype = None
    L.nb_stable_orbits.argtypes = [i32, C.c_float, C.c_float]
    L.nb_stable_orbits.restype = None
    L.nb_step_brute_force.argtypes = [C.c_float]
    L.nb_step_brute_force.restype = None
    L.nb_step_barnes_hut.argtypes = [C.c_float, C.c_float, i32]
    L.nb_step_barnes_hut.restype = None
    L.nb_draw.argtypes = [i32, i32, C.c_void_p]
    L.nb_draw.restype = None
    # level 2
    L.nbx_last_error.restype = C.c_char_p
    L.nbx_version.restype = C.c_char_p
    L.nbx_device_count.restype = i32
    L.nbx_device_info_get.argtypes = [i32, C.POINTER(_DeviceInfo)]
    L.nbx_device_info_get.restype = i32
    L.nbx_create.argtypes = [C.POINTER(E), i32]
    L.nbx_create.restype = i32
    L.nbx_destroy.argtypes = [E]
    L.nbx_destroy.restype = None
    L.nbx_set_option.argtypes = [E, i32, C.c_int64]
    L.nbx_set_option.restype = i32
    L.nbx_get_option.argtypes = [E, i32]
    L.nbx_get_option.restype = C.c_int64
    L.nbx_seed.argtypes = [E, C.c_uint64]
    L.nbx_seed.restype = i32
    L.nbx_random_disk.argtypes = [E, i32]
    L.nbx_random_disk.restype = i32
    L.nbx_stable_orbits.argtypes = [E, i32, C.c_float, C.c_float]
    L.nbx_stable_orbits.restype = i32
    L.nbx_plummer_sphere.argtypes = [E, i32, C.c_uint64, i32]
    L.nbx_plummer_sphere.restype = i32
    L.nbx_two_galaxies.argtypes = [E, i32, C.c_uint64]
    L.nbx_two_galaxies.restype = i32
    L.nbx_query_option.argtypes = [E, i32, C.POINTER(C.c_int64)]
    L.nbx_query_option.restype = i32
    L.nbx_get_stat.argtypes = [E, i32]
    L.nbx_get_stat.restype = C.c_int64
    L.nbx_num_particles.argtypes = [E]
    L.nbx_num_particles.restype = i32
    L.nbx_set_particles.argtypes = [E, i32] + [C.c_void_p] * 5
    L.nbx_set_particles.restype = i32
    L.nbx_set_particles3.argtypes = [E, i32] + [C.c_void_p] * 7
    L.nbx_set_particles3.restype = i32
    L.nbx_get_particles.argtypes = [E, i32] + [C.c_void_p] * 5
    L.nbx_get_particles.restype = i32
    L.nbx_get_particles3.argtypes = [E, i32] + [C.c_void_p] * 7
    L.nbx_get_particles3.restype = i32
    L.nbx_save.argtypes = [E, C.c_char_p]
    L.nbx_save.restype = i32
    L.nbx_load.argtypes = [E, C.c_char_p]
    L.nbx_load.restype = i32
    L.nbx_step_brute_force.argtypes = [E, C.c_float]
    L.nbx_step_brute_force.restype = i32
    L.nbx_step_barnes_hut.argtypes = [E, C.c_float, C.c_float, i32]
    L.nbx_step_barnes_hut.restype = i32
    L.nbx_step_local.argtypes = [E, C.c_float]
    L.nbx_step_local.restype = i32
    L.nbx_synchronize.argtypes = [E]
    L.nbx_synchronize.restype = i32
    L.nbx_forces.argtypes = [E, C.c_float, i32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.nbx_forces.restype = i32
    L.nbx_draw.argtypes = [E, i32, i32, C.c_void_p]
    L.nbx_draw.restype = i32
    L.nbx_bh_tree_dump.argtypes = [E, C.c_void_p, i32]
    L.nbx_bh_tree_dump.restype = i32
    L.nbx_bh_flat_dump.argtypes = [E, C.c_void_p, i32, i32]
    L.nbx_bh_flat_dump.restype = i32
    L.nbx_set_shard.argtypes = [E, i32, i32]
    L.nbx_set_shard.restype = i32
    L.nbx_get_slab.argtypes = [E, C.POINTER(i32), C.POINTER(i32)]
    L.nbx_get_slab.restype = i32
    L.nbx_bind_positions.argtypes = [E, C.c_void_p, C.c_size_t]
    L.nbx_bind_positions.restype = i32
    L.nbx_half_sources_bytes.argtypes = [E]
    L.nbx_half_sources_bytes.restype = C.c_size_t
    L.nbx_bind_half_sources.argtypes = [E, C.c_void_p, C.c_size_t]
    L.nbx_bind_half_sources.restype = i32
    L.nbx_positions_device.argtypes = [E]
    L.nbx_positions_device.restype = C.c_void_p
    L.nbx_positions_bytes.argtypes = [E]
    L.nbx_positions_bytes.restype = C.c_size_t
    L.nbx_set_stream.argtypes = [E, C.c_void_p]
    L.nbx_set_stream.restype = i32
    G = C.c_void_p
    L.nbx_group_create.argtypes = [C.POINTER(G), C.POINTER(i32), i32]
    L.nbx_group_create.restype = i32
    L.nbx_group_destroy.argtypes = [G]
    L.nbx_group_destroy.restype = None
    L.nbx_group_size.argtypes = [G]
    L.nbx_group_size.restype = i32
    L.nbx_group_engine.argtypes = [G, i32]
    L.nbx_group_engine.restype = C.c_void_p
    L.nbx_group_set_option.argtypes = [G, i32, C.c_int64]
    L.nbx_group_set_option.restype = i32
    L.nbx_group_num_particles.argtypes = [G]
    L.nbx_group_num_particles.restype = i32
    L.nbx_group_set_particles3.argtypes = [G, i32] + [C.c_void_p] * 7
    L.nbx_group_set_particles3.restype = i32
    L.nbx_group_get_particles3.argtypes = [G, i32] + [C.c_void_p] * 7
    L.nbx_group_get_particles3.restype = i32
    L.nbx_group_step_brute_force.argtypes = [G, C.c_float]
    L.nbx_group_step_brute_force.restype = i32
    L.nbx_group_step_barnes_hut.argtypes = [G, C.c_float, C.c_float, i32]
    L.nbx_group_step_barnes_hut.restype = i32
    L.nbx_group_synchronize.argtypes = [G]
    L.nbx_group_synchronize.restype = i32
    L.nbx_group_draw.argtypes = [G, i32, i32, C.c_void_p]
    L.nbx_group_draw.restype = i32
    L.nbx_group_exchanges.argtypes = [G]
    L.nbx_group_exchanges.restype = i32
    L.nbx_group_info.argtypes = [G, i32]
    L.nbx_group_info.restype = C.c_int64
    L.nbx_group_exchange_note.argtypes = [G]
    L.nbx_group_exchange_note.restype = C.c_char_p
    L.nbx_group_set_enqueue_threads.argtypes = [G, i32]
    L.nbx_group_set_enqueue_threads.restype = i32
    L.nbx_profile_reset.argtypes = [E]
    L.nbx_profile_reset.restype = i32
    L.nbx_profile_read.argtypes = [E, i32, C.POINTER(C.c_double), C.POINTER(i32)]
    L.nbx_profile_read.restype = i32
    L.nbx_bh_work.argtypes = [E, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.nbx_bh_work.restype = i32
    L.nbx_bh_work_detail.argtypes = [E, C.c_float, C.POINTER(C.c_uint64)]
    L.nbx_bh_work_detail.restype = i32
    L.nbx_bh_walk_trace.argtypes = [E, C.c_float, i32, C.c_void_p]
    L.nbx_bh_walk_trace.restype = i32
    L.nbx_bh_take_threshold.argtypes = [C.c_float, C.c_float]
    L.nbx_bh_take_threshold.restype = C.c_float
    L.nbx_bh_take_thresholds_device.argtypes = [E, i32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.nbx_bh_take_thresholds_device.restype = i32
    L.nbx_bh_host_timing.argtypes = [E, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(i32)]
    L.nbx_bh_host_timing.restype = i32
    L.nbx_last_launch.argtypes = [E] + [C.POINTER(i32)] * 6
    L.nbx_last_launch.restype = i32
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise NBodyError(rc, lib().nbx_last_error().decode(errors="replace"))
    return rc


def device_count():
    return int(lib().nbx_device_count())


def device_info(device=0):
    info = _DeviceInfo()
    _check(lib().nbx_device_info_get(device, C.byref(info)))
    return {
        "name": info.name.decode(),
        "arch": info.arch.decode(),
        "compute_units": info.compute_units,
        "clock_khz": info.clock_khz,
        "wavefront_size": info.wavefront_size,
        "lds_bytes_per_cu": info.lds_bytes_per_cu,
        "peak_fp32_flops": info.peak_fp32_flops,
        "hbm_bytes": info.hbm_bytes,
    }


# ---- level 1: the reference's six symbols (process-global state) -----------------------------

def nb_num_particles():
    return int(lib().nb_num_particles())


def nb_random_disk(num_particles):
    lib().nb_random_disk(num_particles)


def nb_stable_orbits(num_particles, rmin, rmax):
    lib().nb_stable_orbits(num_particles, rmin, rmax)


def nb_step_brute_force(dt):
    lib().nb_step_brute_force(dt)


def nb_step_barnes_hut(theta, dt, nthreads):
    lib().nb_step_barnes_hut(theta, dt, nthreads)


def nb_draw(w, h):
    fb = np.zeros(w * h, np.uint32)
    lib().nb_draw(w, h, fb.ctypes.data_as(C.c_void_p))
    return fb.reshape(h, w)


# ---- level 2 ---------------------------------------------------------------------------------

def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class NBodyEngine:
    """Handle-based engine. Method names follow the reference entry points without the nb_ prefix."""

    def __init__(self, device=0, mode="fast"):
        self._L = lib()
        h = C.c_void_p()
        _check(self._L.nbx_create(C.byref(h), device))
        self._h = h
        self._owned = True
        self.set_mode(mode)

    @classmethod
    def _borrow(cls, handle, keepalive):
        """View of an engine owned by somebody else (a group member, nbx_group_engine): never destroyed from here."""
        self = cls.__new__(cls)
        self._L = lib()
        self._h = C.c_void_p(handle)
        self._owned = False
        self._keepalive = keepalive
        return self

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._L.nbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # options
    def set_option(self, opt, value):
        _check(self._L.nbx_set_option(self._h, opt, int(value)))

    def set_mode(self, mode):
        self.set_option(NBX_OPT_FORCE_MODE, {"fast": 0, "strict": 1}[mode])

    def get_option(self, opt):
        return int(self._L.nbx_get_option(self._h, opt))

    def get_stat(self, stat):
        """What the engine has done so far (enum nbx_stat): fallbacks, where the last tree was built, ..."""
        v = int(self._L.nbx_get_stat(self._h, stat))
        if v == -(1 << 63):
            raise NBodyError(NBX_ERR_INVALID, "unknown stat %d" % stat)
        return v

    def query_option(self, opt):
        """Value of an option with the status checked apart from it (-1 is a legitimate value of some options)."""
        v = C.c_int64()
        _check(self._L.nbx_query_option(self._h, opt, C.byref(v)))
        return int(v.value)

    def set_launch(self, jsplit=0, bodies_per_thread=0, dim=0, variant=-1):
        self.set_option(NBX_OPT_JSPLIT, jsplit)
        self.set_option(NBX_OPT_BODIES_PER_THREAD, bodies_per_thread)
        self.set_option(NBX_OPT_DIM, dim)
        self.set_option(NBX_OPT_KERNEL_VARIANT, variant)

    def set_strict_kernel(self, kernel=0):
        """bit-exact all-pairs kernel: 0 = by size, 16 / 8 = waves per 64-target workgroup (producers + summing wave),
        1 = one thread per body; results are bit-identical"""
        self.set_option(NBX_OPT_STRICT_KERNEL, kernel)

    # presets (nbody.rs:39-104) with a seedable generator
    def seed(self, seed):
        _check(self._L.nbx_seed(self._h, seed))

    def random_disk(self, n):
        _check(self._L.nbx_random_disk(self._h, n))

    def stable_orbits(self, n, rmin, rmax):
        _check(self._L.nbx_stable_orbits(self._h, n, rmin, rmax))

    # benchmark workloads, generated by the library so that every host language gets the same bytes
    def plummer_sphere(self, n, seed=0x5EED0001, dim=3):
        _check(self._L.nbx_plummer_sphere(self._h, n, seed, dim))

    def two_galaxies(self, n, seed=0x5EED0002):
        _check(self._L.nbx_two_galaxies(self._h, n, seed))

    def num_particles(self):
        return _check(self._L.nbx_num_particles(self._h))

    # state
    def set_particles(self, px, py, vx, vy, m, pz=None, vz=None):
        px, py, vx, vy, m = map(_f32, (px, py, vx, vy, m))
        n = len(px)
        assert all(len(a) == n for a in (py, vx, vy, m))
        if pz is None and vz is None:
            _check(self._L.nbx_set_particles(self._h, n, _p(px), _p(py), _p(vx), _p(vy), _p(m)))
        else:
            pz = _f32(np.zeros(n) if pz is None else pz)
            vz = _f32(np.zeros(n) if vz is None else vz)
            _check(self._L.nbx_set_particles3(self._h, n, _p(px), _p(py), _p(pz), _p(vx), _p(vy), _p(vz), _p(m)))

    def get_particles(self):
        n = self.num_particles()
        out = {k: np.zeros(n, np.float32) for k in ("px", "py", "pz", "vx", "vy", "vz", "m")}
        _check(self._L.nbx_get_particles3(self._h, n, *[_p(out[k]) for k in ("px", "py", "pz", "vx", "vy", "vz", "m")]))
        return out

    def save(self, path):
        _check(self._L.nbx_save(self._h, os.fsencode(path)))

    def load(self, path):
        return _check(self._L.nbx_load(self._h, os.fsencode(path)))

    # steps
    def step_brute_force(self, dt):
        _check(self._L.nbx_step_brute_force(self._h, dt))

    def step_barnes_hut(self, theta, dt, nthreads=1):
        _check(self._L.nbx_step_barnes_hut(self._h, theta, dt, nthreads))

    def step_local(self, dt):
        _check(self._L.nbx_step_local(self._h, dt))

    def synchronize(self):
        _check(self._L.nbx_synchronize(self._h))

    def forces(self, theta=0.0):
        lo, hi = self.slab()
        fx = np.zeros(hi - lo, np.float32)
        fy = np.zeros(hi - lo, np.float32)
        fz = np.zeros(hi - lo, np.float32)
        _check(self._L.nbx_forces(self._h, theta, hi - lo, _p(fx), _p(fy), _p(fz)))
        return fx, fy, fz

    def set_bh_tree(self, where):
        """'host' (reference-faithful insertion build), 'device' (bh_build.hip) or 'auto' (default: device in the fast
        mode from 512 bodies on)."""
        self.set_option(NBX_OPT_BH_TREE, {"host": 0, "device": 1, "auto": -1}[where])

    def set_bh_fold(self, how):
        """Interior nodes of the device-built tree: 'reference' (the f32 running fold in arrival order, nbody.rs:303-320: the host
        tree bit for bit), 'exact' (roundings of exact sums) or 'auto' (reference up to 65 536 bodies)."""
        self.set_option(NBX_OPT_BH_FOLD, {"exact": 0, "reference": 1, "auto": -1}[how])

    def set_draw_device(self, on=True):
        """True / False force the device / host draw; None = by size (the default)."""
        self.set_option(NBX_OPT_DRAW_DEVICE, -1 if on is None else (1 if on else 0))

    def draw(self, w, h):
        fb = np.zeros(w * h, np.uint32)
        _check(self._L.nbx_draw(self._h, w, h, _p(fb)))
        return fb.reshape(h, w)

    def bh_tree_dump(self):
        cnt = _check(self._L.nbx_bh_tree_dump(self._h, None, 0))
        rows = np.zeros((max(cnt, 1), 8), np.float32)
        cnt = _check(self._L.nbx_bh_tree_dump(self._h, _p(rows), cnt))
        return rows[:cnt]

    def bh_flat_dump(self, threaded=False):
        """Flattened tree as a structured array (px, py, m, s, skip, interior)."""
        dt = np.dtype([("px", "<f4"), ("py", "<f4"), ("m", "<f4"), ("s", "<f4"), ("skip", "<i4"), ("interior", "<i4"),
                       ("q", "<f4"), ("pad1", "<i4")])
        mode = 2 if threaded == "device" else int(bool(threaded))
        cnt = _check(self._L.nbx_bh_flat_dump(self._h, None, 0, mode))
        rows = np.zeros(max(cnt, 1), dt)
        cnt = _check(self._L.nbx_bh_flat_dump(self._h, _p(rows), cnt, mode))
        return rows[:cnt]

    # sharding
    def set_shard(self, rank, world):
        _check(self._L.nbx_set_shard(self._h, rank, world))

    def slab(self):
        lo, hi = C.c_int32(), C.c_int32()
        _check(self._L.nbx_get_slab(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def positions_bytes(self):
        return int(self._L.nbx_positions_bytes(self._h))

    def bind_positions(self, device_ptr, nbytes):
        _check(self._L.nbx_bind_positions(self._h, C.c_void_p(device_ptr), nbytes))

    def set_source_precision(self, bits):
        """16: all-pairs sources come from a half4 copy (BASELINE config #5); 32: default."""
        self.set_option(NBX_OPT_SOURCE_PRECISION, bits)

    def half_sources_bytes(self):
        return int(self._L.nbx_half_sources_bytes(self._h))

    def bind_half_sources(self, device_ptr, nbytes):
        _check(self._L.nbx_bind_half_sources(self._h, C.c_void_p(device_ptr), nbytes))

    def set_stream(self, hip_stream):
        _check(self._L.nbx_set_stream(self._h, C.c_void_p(hip_stream)))

    # profiling
    def profile(self, on=True):
        self.set_option(NBX_OPT_PROFILE, 1 if on else 0)

    def profile_reset(self):
        _check(self._L.nbx_profile_reset(self._h))

    def profile_read(self, kernel_id):
        ms, cnt = C.c_double(), C.c_int32()
        _check(self._L.nbx_profile_read(self._h, kernel_id, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def bh_work(self, theta):
        v, q = C.c_uint64(), C.c_uint64()
        _check(self._L.nbx_bh_work(self._h, theta, C.byref(v), C.byref(q)))
        return {"node_visits": v.value, "pair_evals": q.value}

    def bh_work_detail(self, theta):
        out = (C.c_uint64 * 4)()
        _check(self._L.nbx_bh_work_detail(self._h, theta, out))
        return {"node_visits": out[0], "pair_evals": out[1], "opening_tests": out[2], "group_loads": out[3]}

    def bh_walk_trace(self, theta):
        """[walks, 4] uint64: s_memrealtime (10 ns ticks) start, end, groups loaded (bit 31: redone with the LDS spill) | chunk << 32, HW_ID | XCC_ID << 32 of every walk of one traversal."""
        n = _check(self._L.nbx_bh_walk_trace(self._h, theta, 1, np.zeros(4, np.uint64).ctypes.data))
        out = np.zeros((n, 4), np.uint64)
        _check(self._L.nbx_bh_walk_trace(self._h, theta, n, out.ctypes.data))
        return out

    def bh_take_thresholds(self, s, theta):
        """bh_threshold.h's T for every (s[i], theta[i]), evaluated on this engine's GPU (test hook)."""
        s = np.ascontiguousarray(s, dtype=np.float32)
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        out = np.empty_like(s)
        _check(self._L.nbx_bh_take_thresholds_device(self._h, s.size, s.ctypes.data, theta.ctypes.data, out.ctypes.data))
        return out

    def bh_host_timing(self):
        ms = (C.c_double * 4)()
        steps, nodes = C.c_int32(), C.c_int32()
        _check(self._L.nbx_bh_host_timing(self._h, ms, C.byref(steps), C.byref(nodes)))
        k = max(steps.value, 1)
        return {"download_ms": ms[0] / k, "build_ms": ms[1] / k, "flatten_ms": ms[2] / k, "upload_ms": ms[3] / k,
                "steps": steps.value, "nodes": nodes.value}

    def last_launch(self):
        v = [C.c_int32() for _ in range(6)]
        _check(self._L.nbx_last_launch(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("grid", "block", "jsplit", "bodies_per_thread", "dim", "variant"), (x.value for x in v)))


class NBodyGroup:
    """Single-process multi-GPU group (nbx_group_*): G engines, slab-sharded, one RCCL all-gather per step
    issued by the library. What NB_GPUS=<n> gives the six nb_* symbols."""

    def __init__(self, devices, mode="fast"):
        self._L = lib()
        devs = list(devices)
        arr = (C.c_int32 * len(devs))(*devs)
        h = C.c_void_p()
        _check(self._L.nbx_group_create(C.byref(h), arr, len(devs)))
        self._h = h
        self.set_option(NBX_OPT_FORCE_MODE, {"fast": 0, "strict": 1}[mode])

    def close(self):
        if getattr(self, "_h", None):
            self._L.nbx_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return _check(self._L.nbx_group_size(self._h))

    def engine(self, i):
        """Member engine i (per-engine options, profiling, slab, forces)."""
        h = self._L.nbx_group_engine(self._h, i)
        if not h:
            raise IndexError(i)
        return NBodyEngine._borrow(h, self)

    def set_source_precision(self, bits):
        self.set_option(NBX_OPT_SOURCE_PRECISION, bits)

    def set_option(self, opt, value):
        _check(self._L.nbx_group_set_option(self._h, opt, int(value)))

    def num_particles(self):
        return _check(self._L.nbx_group_num_particles(self._h))

    def set_particles(self, px, py, vx, vy, m, pz=None, vz=None):
        px, py, vx, vy, m = map(_f32, (px, py, vx, vy, m))
        n = len(px)
        pz = _f32(np.zeros(n) if pz is None else pz)
        vz = _f32(np.zeros(n) if vz is None else vz)
        _check(self._L.nbx_group_set_particles3(self._h, n, _p(px), _p(py), _p(pz), _p(vx), _p(vy), _p(vz), _p(m)))

    def get_particles(self):
        n = self.num_particles()
        out = {k: np.zeros(n, np.float32) for k in ("px", "py", "pz", "vx", "vy", "vz", "m")}
        _check(self._L.nbx_group_get_particles3(self._h, n, *[_p(out[k]) for k in ("px", "py", "pz", "vx", "vy", "vz", "m")]))
        return out

    def step_brute_force(self, dt):
        _check(self._L.nbx_group_step_brute_force(self._h, dt))

    def step_barnes_hut(self, theta, dt, nthreads=1):
        _check(self._L.nbx_group_step_barnes_hut(self._h, theta, dt, nthreads))

    def synchronize(self):
        _check(self._L.nbx_group_synchronize(self._h))

    def draw(self, w, h):
        fb = np.zeros(w * h, np.uint32)
        _check(self._L.nbx_group_draw(self._h, w, h, _p(fb)))
        return fb.reshape(h, w)

    def exchanges(self):
        return _check(self._L.nbx_group_exchanges(self._h))

    def info(self):
        """How the per-step exchange runs: kind ('rccl' | 'peer_copy' | 'peer_copy_after_rccl_failure'), the ranks
        ncclCommInitAll was given, enqueue threads, and why the group fell back (if it did)."""
        kind = int(self._L.nbx_group_info(self._h, NBX_GROUP_INFO_EXCHANGE))
        return {"exchange": {0: "rccl", 1: "peer_copy", 2: "peer_copy_after_rccl_failure"}.get(kind, str(kind)),
                "rccl_ranks": int(self._L.nbx_group_info(self._h, NBX_GROUP_INFO_RCCL_RANKS)),
                "enqueue_threads": int(self._L.nbx_group_info(self._h, NBX_GROUP_INFO_ENQUEUE_THREADS)),
                "fp32_stale": bool(self._L.nbx_group_info(self._h, NBX_GROUP_INFO_FP32_STALE)),
                "note": (self._L.nbx_group_exchange_note(self._h) or b"").decode(errors="replace")}

    def set_enqueue_threads(self, on=True):
        _check(self._L.nbx_group_set_enqueue_threads(self._h, 1 if on else 0))
