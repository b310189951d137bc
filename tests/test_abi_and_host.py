"""CPU: the C-ABI library loads, exports every symbol include/nbody_mi355x.h declares, refuses to
step without a GPU (no CPU fallback), and its host-side pieces (presets, draw, quadtree build)
agree bit for bit with the oracle.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal, golden, particles_from

HEADER = os.path.join(ROOT, "include", "nbody_mi355x.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nbx?_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_six_reference_symbols():
    # rs-src/nbody.rs:34-35,:39-40,:73-74,:106-107,:186-187,:482-483 / RustNBodyExperiment.hs:101-106
    six = {"nb_num_particles", "nb_random_disk", "nb_stable_orbits", "nb_step_brute_force", "nb_step_barnes_hut", "nb_draw"}
    assert six <= set(declared_symbols())


def test_library_exports_every_declared_symbol(rx):
    L = C.CDLL(rx.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"


def test_library_is_the_hip_build(rx):
    # the product must be the gfx950 code object, not something else
    blob = open(rx.lib_path(), "rb").read()
    assert b"gfx950" in blob
    assert b"k_force_tile" in blob


def test_no_cpu_fallback_without_device(rx):
    if rx.device_count() > 0:
        pytest.skip("device present")
    e = rx.NBodyEngine()
    e.seed(1)
    e.stable_orbits(64, 0.5, 30.0)
    with pytest.raises(rx.NBodyError) as ei:
        e.step_brute_force(0.01)
    assert ei.value.code == rx.NBX_ERR_NO_DEVICE
    with pytest.raises(rx.NBodyError):
        e.step_barnes_hut(0.5, 0.01, 1)
    with pytest.raises(rx.NBodyError):
        e.forces()
    # state untouched by the refused step
    st = e.get_particles()
    assert st["m"][0] == 1000.0


def test_product_presets_equal_oracle_presets(rx, ob):
    e = rx.NBodyEngine()
    e.seed(7)
    e.random_disk(257)
    st = e.get_particles()
    o = ob.random_disk(257, 7)
    for k in ("px", "py", "vx", "vy", "m"):
        assert_bit_equal(st[k], o[k], "disk " + k)
    assert not st["pz"].any() and not st["vz"].any()
    e.seed(9)
    e.stable_orbits(100, 0.5, 30.0)
    st = e.get_particles()
    o = ob.stable_orbits(100, 0.5, 30.0, 9)
    for k in ("px", "py", "vx", "vy", "m"):
        assert_bit_equal(st[k], o[k], "orbits " + k)
    # reference edge cases: i32 arithmetic, `0..n-1` empty for n <= 1 (nbody.rs:95)
    e.stable_orbits(0, 1.0, 2.0)
    assert e.num_particles() == 1
    e.random_disk(0)
    assert e.num_particles() == 0
    e.random_disk(-3)
    assert e.num_particles() == 0
    # consecutive presets continue one generator stream, like one thread_rng
    e.seed(5); e.random_disk(10); e.random_disk(10)
    a = e.get_particles()["px"].copy()
    e.seed(5); e.random_disk(10)
    assert not np.array_equal(a, e.get_particles()["px"])


def test_product_presets_match_golden(rx):
    g = golden("presets")
    e = rx.NBodyEngine()
    e.seed(7); e.random_disk(257)
    st = e.get_particles()
    got = np.stack([st[k] for k in ("px", "py", "vx", "vy", "m")], axis=1)
    assert np.array_equal(got.view(np.uint32), g["disk_seed7_n257"].view(np.uint32))


def test_set_get_roundtrip_2d_and_3d(rx):
    rng = np.random.default_rng(3)
    n = 300
    a = {k: rng.normal(size=n).astype(np.float32) for k in ("px", "py", "pz", "vx", "vy", "vz")}
    a["m"] = rng.uniform(0.1, 2, n).astype(np.float32)
    e = rx.NBodyEngine()
    e.set_particles(a["px"], a["py"], a["vx"], a["vy"], a["m"], a["pz"], a["vz"])
    st = e.get_particles()
    for k in a:
        assert_bit_equal(st[k], a[k], k)
    e.set_particles(a["px"], a["py"], a["vx"], a["vy"], a["m"])
    st = e.get_particles()
    assert not st["pz"].any() and not st["vz"].any()
    e.set_particles([], [], [], [], [])
    assert e.num_particles() == 0


@pytest.mark.parametrize("shape", [(64, 48), (512, 512), (100, 37)])
def test_product_draw_equals_oracle_draw(rx, ob, shape):
    w, h = shape
    p = ob.stable_orbits(1024, 0.5, 30.0, 1)
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    assert np.array_equal(e.draw(w, h), ob.draw(p, w, h))
    d = ob.random_disk(3000, 8)
    d["px"][:50] *= 3.0   # push some bodies out of the viewport
    e.set_particles(d["px"], d["py"], d["vx"], d["vy"], d["m"])
    assert np.array_equal(e.draw(w, h), ob.draw(d, w, h))


@pytest.mark.parametrize("shape", [(512, 512), (101, 37)])
def test_product_draw_of_a_big_system_uses_threads_and_equals_oracle_draw(rx, ob, shape):
    """n >= 65536: all host threads splat into hit counters, resolved to min(255, hits * colour) per channel --
    the saturating adds of nbody.rs:595-617 in any order. Heavy overlap (saturation) and out-of-view bodies included."""
    w, h = shape
    p = ob.stable_orbits(150000, 0.5, 30.0, 2)
    p["px"][:2000] = 0.125; p["py"][:2000] = -0.25        # 2000 bodies on one pixel
    p["px"][2000:2500] *= 4.0                              # out of the viewport
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    assert np.array_equal(e.draw(w, h), ob.draw(p, w, h))


def test_product_draw_matches_golden(rx):
    g = golden("draw_n1024_orbits")
    e = rx.NBodyEngine()
    e.set_particles(g["in_px"], g["in_py"], g["in_vx"], g["in_vy"], g["in_m"])
    assert np.array_equal(e.draw(64, 48), g["fb_64x48"])
    assert np.array_equal(e.draw(512, 512), g["fb_512x512"])


def test_level1_host_entry_points(rx):
    # the six-symbol surface on the process-global engine (host-side ones only on CPU)
    os.environ.setdefault("NB_SEED", "42")
    rx.nb_stable_orbits(500, 0.5, 30.0)
    assert rx.nb_num_particles() == 500
    fb = rx.nb_draw(128, 128)
    assert (fb == 0x00FF00FF).sum() == 5
    assert (fb != 0).sum() > 100
    rx.nb_random_disk(123)
    assert rx.nb_num_particles() == 123
    rx.nb_random_disk(0)
    assert rx.nb_num_particles() == 0


@pytest.mark.parametrize("make", ["disk", "orbits", "clumps", "line", "merge"])
def test_product_quadtree_equals_oracle_tree_node_for_node(rx, ob, make):
    rng = np.random.default_rng(4)
    if make == "disk":
        p = ob.random_disk(2000, 31)
    elif make == "orbits":
        p = ob.stable_orbits(3000, 0.5, 30.0, 32)
    elif make == "clumps":
        c = rng.normal(size=(1500, 2)).astype(np.float32) * 0.01
        c[:750] += 10.0
        p = ob.particles(c[:, 0], c[:, 1], np.zeros(1500), np.zeros(1500), rng.uniform(0.5, 2, 1500))
    elif make == "line":   # degenerate AABB: all y equal (root box has zero height)
        p = ob.particles(np.linspace(-20, 20, 300), np.zeros(300), np.zeros(300), np.zeros(300), np.ones(300))
    else:                  # pairs closer than EPS merge into one exterior node (nbody.rs:249-260)
        x = rng.uniform(-20, 20, 200).astype(np.float32)
        y = rng.uniform(-20, 20, 200).astype(np.float32)
        p = ob.particles(np.concatenate([x, x + np.float32(3e-5)]), np.concatenate([y, y]), np.zeros(400),
                         np.zeros(400), np.ones(400))
    rc, want = ob.bh_tree_dump(p)
    assert rc == 0
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    got = e.bh_tree_dump()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("threads", ["1", "2", "7", "32"])
def test_threaded_quadtree_build_is_result_identical(rx, ob, threads, monkeypatch):
    """n >= 32768 takes the threaded build (top levels sequential, subtrees replayed in parallel):
    node for node identical to the oracle's sequential tree, for any thread count, including
    EPS-merged pairs and a dense clump that forces deep subtrees."""
    monkeypatch.setenv("NBX_HOST_THREADS", threads)
    rng = np.random.default_rng(5)
    n = 40000
    x = rng.normal(size=n).astype(np.float32) * 8
    y = rng.normal(size=n).astype(np.float32) * 8
    x[:4000] = x[:4000] * np.float32(0.001) + np.float32(3.0)          # dense clump
    y[:4000] = y[:4000] * np.float32(0.001) - np.float32(2.0)
    x[5000:5100] = x[6000:6100] + np.float32(2e-5)                     # pairs closer than EPS: merged
    y[5000:5100] = y[6000:6100]
    m = rng.uniform(0.1, 2.0, n).astype(np.float32)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
    rc, want = ob.bh_tree_dump(p)
    assert rc == 0
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    got = e.bh_tree_dump()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("case", ["limit5_random", "limit6_plummer", "corner_first", "sorted_x", "sparse_ring", "two_clumps"])
def test_threaded_build_parallel_top_phase_is_result_identical(rx, ob, case, monkeypatch):
    """The warm-up / freeze / route-fold-scatter scheme (host_tree.cpp) for every bucket depth, for orders that
    leave many top nodes exterior at the freeze, and for very unbalanced buckets: node-for-node the oracle tree."""
    monkeypatch.setenv("NBX_HOST_THREADS", "12")
    rng = np.random.default_rng(11)
    if case == "limit5_random":
        n = 70000
        x = rng.uniform(-40, 40, n); y = rng.uniform(-25, 25, n)
    elif case == "limit6_plummer":
        st = rx.plummer_sphere(270000, dim=2)
        x, y = st["px"], st["py"]; n = len(x)
    elif case == "corner_first":      # the first 8192 bodies sit in one corner: most of the box is exterior at the freeze
        n = 80000
        x = rng.uniform(-40, 40, n); y = rng.uniform(-40, 40, n)
        x[:9000] = rng.uniform(35, 40, 9000); y[:9000] = rng.uniform(35, 40, 9000)
    elif case == "sorted_x":
        n = 66000
        x = np.sort(rng.normal(0, 10, n)); y = rng.normal(0, 10, n)
    elif case == "sparse_ring":
        n = 68000
        a = rng.uniform(0, 2 * np.pi, n); r = 30 + rng.normal(0, 0.01, n)
        x = r * np.cos(a); y = r * np.sin(a)
    else:
        n = 90000
        x = np.concatenate([rng.normal(-20, 0.05, n // 2), rng.normal(20, 3, n - n // 2)])
        y = np.concatenate([rng.normal(5, 0.05, n // 2), rng.normal(-5, 3, n - n // 2)])
    m = rng.uniform(0.1, 2.0, n).astype(np.float32)
    p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
    rc, want = ob.bh_tree_dump(p)
    assert rc == 0
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    got = e.bh_tree_dump()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    monkeypatch.setenv("NBX_HOST_THREADS", "1")
    a = e.bh_flat_dump(False)
    monkeypatch.setenv("NBX_HOST_THREADS", "12")
    b = e.bh_flat_dump(True)
    assert a.tobytes() == b.tobytes()


def test_threaded_build_reports_nonpositive_mass(rx, monkeypatch):
    monkeypatch.setenv("NBX_HOST_THREADS", "4")
    rng = np.random.default_rng(12)
    n = 50000
    m = np.ones(n, np.float32); m[30000] = 0.0
    e = rx.NBodyEngine()
    e.set_particles(rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), np.zeros(n), np.zeros(n), m)
    with pytest.raises(rx.NBodyError) as ei:
        e.bh_tree_dump()
    assert ei.value.code == rx.NBX_ERR_TREE      # nbody.rs:304


def test_threaded_quadtree_reports_depth_panic(rx, monkeypatch):
    monkeypatch.setenv("NBX_HOST_THREADS", "4")
    rng = np.random.default_rng(6)
    n = 40000
    x = rng.uniform(-20, 20, n).astype(np.float32)
    y = rng.uniform(-20, 20, n).astype(np.float32)
    x[0], y[0] = 1e30, 1e30                                           # huge box: two close bodies need > 50 splits
    x[1], y[1], x[2], y[2] = 1.0, 1.0, 1.0003, 1.0
    e = rx.NBodyEngine()
    e.set_particles(x, y, np.zeros(n), np.zeros(n), np.ones(n))
    with pytest.raises(rx.NBodyError) as ei:
        e.bh_tree_dump()
    assert ei.value.code == rx.NBX_ERR_TREE_DEPTH


def test_product_quadtree_reports_reference_panics(rx, ob):
    e = rx.NBodyEngine()
    e.set_particles([0.0, 1e30, 1.0, 1.0003], [0.0, 1e30, 1.0, 1.0], [0] * 4, [0] * 4, [1.0] * 4)
    with pytest.raises(rx.NBodyError) as ei:
        e.bh_tree_dump()
    assert ei.value.code == rx.NBX_ERR_TREE_DEPTH          # nbody.rs:230-232
    e.set_particles([0.0, 1.0], [0.0, 1.0], [0, 0], [0, 0], [1.0, 0.0])
    with pytest.raises(rx.NBodyError) as ei:
        e.bh_tree_dump()
    assert ei.value.code == rx.NBX_ERR_TREE                # nbody.rs:304


def test_reference_slab_split(rx):
    # nbody.rs:426-428
    assert rx.reference_slab(10, 0, 3) == (0, 3)
    assert rx.reference_slab(10, 1, 3) == (3, 6)
    assert rx.reference_slab(10, 2, 3) == (6, 10)
    e = rx.NBodyEngine()
    e.set_shard(2, 3)
    e.set_particles(np.zeros(10), np.zeros(10), np.zeros(10), np.zeros(10), np.ones(10))
    assert e.slab() == (6, 10)
    for n, w in ((262144, 8), (1000, 7), (5, 8)):
        covered = []
        for r in range(w):
            lo, hi = rx.reference_slab(n, r, w)
            covered += list(range(lo, hi))
        assert covered == list(range(n))


def test_plummer_generator_is_deterministic_and_bounded(rx):
    a = rx.plummer_sphere(4096)
    b = rx.plummer_sphere(4096)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    r = np.sqrt(a["px"].astype(np.float64) ** 2 + a["py"] ** 2 + a["pz"] ** 2)
    assert r.max() <= 45.0 + 1e-3 and np.median(r) > 3.0 and np.median(r) < 9.0
    assert abs(a["m"].sum() - 1000.0) < 1e-2
    c = rx.plummer_sphere(4096, dim=2)
    assert not c["pz"].any() and np.array_equal(c["px"], a["px"])
    u = rx.splitmix64_uniform(123, 5)
    import ctypes

    from oracle import binding as ob

    s = ctypes.c_uint64(123)
    assert [float(x) for x in u] == [float(ob.lib().orc_next_f32(ctypes.byref(s))) for _ in range(5)]


def test_threaded_flatten_equals_serial_flatten(rx, ob, monkeypatch):
    """The array the GPU traversal walks: forest build + host-thread flattener == sequential build +
    serial flattener, byte for byte, and its structure is a valid pre-order with skip pointers."""
    p = ob.stable_orbits(60000, 0.5, 30.0, 6)
    e = rx.NBodyEngine()
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    monkeypatch.setenv("NBX_HOST_THREADS", "1")      # sequential build, one contiguous pool
    a = e.bh_flat_dump(False)
    monkeypatch.setenv("NBX_HOST_THREADS", "8")      # forest of bucket pools, threaded flatten
    b = e.bh_flat_dump(True)
    c = e.bh_flat_dump(False)
    assert len(a) > 100000 and a.tobytes() == b.tobytes() and a.tobytes() == c.tobytes()
    assert a["skip"][0] == len(a) and a["interior"][0] == 1
    leaves = a[a["interior"] == 0]
    assert np.all(leaves["m"] > 0)                           # empty exterior nodes are dropped
    idx = np.arange(len(a))
    assert np.all(a["skip"] > idx) and np.all(a["skip"] <= len(a))
    assert np.all(a["skip"][a["interior"] == 0] == idx[a["interior"] == 0] + 1)
    assert abs(float(a["m"][0]) - float(p["m"].astype(np.float64).sum())) < 1e-2 * 60000
    assert len(leaves) <= 60000


def test_checkpoint_roundtrip(rx, tmp_path):
    rng = np.random.default_rng(9)
    n = 777
    a = {k: rng.normal(size=n).astype(np.float32) for k in ("px", "py", "pz", "vx", "vy", "vz")}
    a["m"] = rng.uniform(0.1, 2, n).astype(np.float32)
    e = rx.NBodyEngine()
    e.set_particles(a["px"], a["py"], a["vx"], a["vy"], a["m"], a["pz"], a["vz"])
    path = str(tmp_path / "state.nbx")
    e.save(path)
    assert open(path, "rb").read(8) == b"NBXCKPT1"
    f = rx.NBodyEngine()
    assert f.load(path) == n
    st = f.get_particles()
    for k in a:
        assert_bit_equal(st[k], a[k], k)
    bad = str(tmp_path / "bad.nbx")
    open(bad, "wb").write(b"not a checkpoint")
    with pytest.raises(rx.NBodyError):
        f.load(bad)
    assert f.num_particles() == n           # a failed load leaves the state alone
    # a header that promises more than the file holds (truncated / corrupt) is rejected BEFORE anything is allocated from it
    import struct
    whole = open(path, "rb").read()
    for blob in (whole[:-4], whole + b"\0" * 4, whole[:8] + struct.pack("<ii", 2**31 - 1, 0) + whole[16:], whole[:8] + struct.pack("<ii", -1, 0)):
        open(bad, "wb").write(blob)
        with pytest.raises(rx.NBodyError) as ei:
            f.load(bad)
        assert ei.value.code == rx.NBX_ERR_INVALID and f.num_particles() == n
    e.set_particles([], [], [], [], [])
    e.save(path)
    assert f.load(path) == 0


def test_group_host_side_without_device(rx):
    """nbx_group_*: slab layout and state replication are host logic (no GPU needed)."""
    n = 1000
    rng = np.random.default_rng(2)
    a = {k: rng.normal(size=n).astype(np.float32) for k in ("px", "py", "vx", "vy")}
    m = rng.uniform(0.5, 2, n).astype(np.float32)
    if rx.device_count() == 0:
        g = rx.NBodyGroup([0, 1, 2])          # device ordinals are only checked when a driver is present
        assert g.size() == 3
        g.set_particles(a["px"], a["py"], a["vx"], a["vy"], m)
        assert g.num_particles() == n
        st = g.get_particles()
        for k in a:
            assert_bit_equal(st[k], a[k], k)
        with pytest.raises(rx.NBodyError) as ei:
            g.step_brute_force(0.01)
        assert ei.value.code == rx.NBX_ERR_NO_DEVICE
    with pytest.raises(rx.NBodyError):
        rx.NBodyGroup([0, 0])                 # the same device twice


def test_group_info_threads_and_option_queries_without_device(rx, monkeypatch):
    """Host-side facts of the group added in round 3: how the exchange runs (nothing created yet: kind 'rccl', 0 ranks), the
    enqueue-thread switch (threads start and stop without a device), NBX_GROUP_EXCHANGE=copy, and the option round trips of
    ADVICE r02 (NBX_OPT_BH_TREE accepts -1 again; nbx_query_option tells the value -1 from an error)."""
    from rust_exp_amd.engine import NBX_OPT_BH_TREE, NBX_STAT_DRAW_AMBIGUOUS, NBX_OPT_DRAW_DEVICE

    monkeypatch.delenv("NBX_GROUP_EXCHANGE", raising=False)
    if rx.device_count() == 0:
        g = rx.NBodyGroup([0, 1, 2, 3])
        assert g.info() == {"exchange": "rccl", "rccl_ranks": 0, "enqueue_threads": 0, "fp32_stale": False, "note": ""}
        g.set_enqueue_threads(True)
        assert g.info()["enqueue_threads"] == 4
        g.set_particles(np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(9), np.ones(9))
        with pytest.raises(rx.NBodyError) as ei:      # the worker threads' failure text reaches this thread
            g.step_brute_force(0.01)
        assert ei.value.code == rx.NBX_ERR_NO_DEVICE and "no HIP device" in str(ei.value)
        g.set_enqueue_threads(False)
        assert g.info()["enqueue_threads"] == 0
        g.close()
        monkeypatch.setenv("NBX_GROUP_ENQUEUE", "threads")
        g = rx.NBodyGroup([0, 1])
        assert g.info()["enqueue_threads"] == 2
        g.close()
    monkeypatch.setenv("NBX_GROUP_EXCHANGE", "copy")
    g = rx.NBodyGroup([0, 0, 0])              # engines may share a device with the copy exchange
    assert g.info()["exchange"] == "peer_copy" and g.size() == 3
    g.close()
    e = rx.NBodyEngine()
    for where, val in (("host", 0), ("device", 1), ("auto", -1)):
        e.set_bh_tree(where)
        assert e.query_option(NBX_OPT_BH_TREE) == val
    with pytest.raises(rx.NBodyError):
        e.set_option(NBX_OPT_BH_TREE, 2)
    assert e.query_option(NBX_OPT_DRAW_DEVICE) == -1 and e.get_stat(NBX_STAT_DRAW_AMBIGUOUS) == 0
    with pytest.raises(rx.NBodyError):
        e.query_option(99)
    from rust_exp_amd.engine import NBX_STAT_BH_REFUSAL
    assert e.get_stat(NBX_STAT_BH_REFUSAL) == 0             # no device build has refused anything yet
    with pytest.raises(rx.NBodyError):
        e.get_stat(99)
    # round 5: the read-only words are stats, no longer options (10-12, 17); the measured losers 16 and 19 are gone
    for retired in (10, 11, 12, 16, 17, 19):
        with pytest.raises(rx.NBodyError):
            e.query_option(retired)
        with pytest.raises(rx.NBodyError):
            e.set_option(retired, 1)
    from rust_exp_amd.engine import NBX_OPT_BH_FUSE_KICK, NBX_OPT_BH_WALK
    assert e.query_option(NBX_OPT_BH_WALK) == 1
    assert e.query_option(NBX_OPT_BH_FUSE_KICK) == 1
    for v in (0, 1):
        e.set_option(NBX_OPT_BH_FUSE_KICK, v)
        assert e.query_option(NBX_OPT_BH_FUSE_KICK) == v
    with pytest.raises(rx.NBodyError):
        e.set_option(NBX_OPT_BH_FUSE_KICK, 2)
    with pytest.raises(rx.NBodyError):
        e.query_option(NBX_OPT_BH_FUSE_KICK + 1)


def test_host_worker_pool_serves_concurrent_builds(rx, ob):
    """Several engines building big trees at the same time from different caller threads (ctypes drops the GIL): the
    shared worker pool must neither deadlock nor mix results -- every dump equals the single-threaded one."""
    import threading

    ps = [ob.stable_orbits(40000 + 1000 * k, 0.5, 30.0, 100 + k) for k in range(4)]
    want = []
    for p in ps:
        e = rx.NBodyEngine()
        e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
        want.append(e.bh_flat_dump(False))
    got = [None] * len(ps)
    errs = []

    def work(k):
        try:
            e = rx.NBodyEngine()
            e.set_particles(ps[k]["px"], ps[k]["py"], ps[k]["vx"], ps[k]["vy"], ps[k]["m"])
            for _ in range(4):
                got[k] = e.bh_flat_dump(True)
        except Exception as ex:   # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(ps))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive(), "worker pool deadlock"
    assert not errs, errs
    for k in range(len(ps)):
        assert np.array_equal(got[k], want[k])


def test_bench_host_selection_without_a_gpu():
    """bench.py picks its host from the launch line alone: `--gpus N` without a launcher -> the single-process group (which
    refuses when the box has fewer devices instead of pretending), WORLD_SIZE set -> one process per GPU, and the two cannot
    be mixed up.  No GPU needed: every case stops before the first device call (or at it, loudly: no CPU fallback)."""
    import subprocess
    import sys

    from conftest import ROOT

    def run(args, env_extra=None):
        env = dict(os.environ)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NBX_GROUP_EXCHANGE"):
            env.pop(k, None)
        env.update(env_extra or {})
        return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)

    import rust_exp_amd as rx

    if rx.device_count() == 0:
        r = run(["--gpus", "2", "--bodies", "1024", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
        assert r.returncode != 0 and "HIP device" in r.stderr and r.stdout.strip() == ""
        r = run(["--bodies", "1024", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
        # round 4: a run that dies still prints ONE line -- no value, but the stage it died in (tests/test_bench_contract.py)
        import json

        assert r.returncode == 4 and "no CPU fallback" in r.stderr
        d = json.loads(r.stdout.strip())
        assert d["value"] is None and "no CPU fallback" in d["error"] and d["diagnostics"]["stage"].startswith("first exchange")
    r = run(["--host", "torch", "--gpus", "4", "--bodies", "1024"])
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr
    r = run(["--host", "single", "--gpus", "2"])
    assert r.returncode != 0 and "one GPU" in r.stderr
