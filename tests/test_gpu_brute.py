"""GPU parity: the HIP brute-force path (through the C ABI) against the oracle and the golden
vectors.  strict mode = bit-exact; fast mode = within the stated fp32 tolerance
(SURVEY.md 8(d) / DESIGN.md section 4):
    accelerations   max|dF| / max|F|            <= 1e-5
    1 step          max|dp| <= 1e-5 , max|dv|   <= 1e-5 * max|a| * dt * max(1, sqrt(N)/64)
    10 steps        max|dp| <= 1e-4 , max|dv|   <= 5e-3
    vs fp64         GPU error <= 2 x the f32 oracle's own error
"""
import numpy as np
import pytest

from conftest import fast_tolerances, assert_bit_equal, golden, particles_from

pytestmark = pytest.mark.gpu

BRUTE = ["brute_n2", "brute_n5_orbits", "brute_n64_disk", "brute_n1024_orbits", "brute_n1000_disk"]
DT = 0.01


def load(e, p, three_d=False):
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])


# ---------------------------------------------------------------- strict = bit exact
@pytest.mark.parametrize("name", BRUTE)
def test_strict_matches_golden_bitwise(rx, name):
    g = golden(name)
    e = rx.NBodyEngine(mode="strict")
    e.set_particles(g["in_px"], g["in_py"], g["in_vx"], g["in_vy"], g["in_m"])
    done = 0
    for s in (1, 10):
        while done < s:
            e.step_brute_force(float(g["dt"]))
            done += 1
        st = e.get_particles()
        for k in ("px", "py", "vx", "vy"):
            assert_bit_equal(st[k], g[f"s{s}_{k}"], f"{name} step {s} {k}")


@pytest.mark.parametrize("n,seed,kind", [(1, 1, "disk"), (3, 2, "disk"), (255, 3, "disk"), (256, 4, "orbits"),
                                         (257, 5, "disk"), (4096, 6, "orbits"), (10000, 7, "orbits")])
def test_strict_matches_oracle_bitwise(rx, ob, n, seed, kind):
    p = ob.random_disk(n, seed) if kind == "disk" else ob.stable_orbits(n, 0.5, 30.0, seed)
    e = rx.NBodyEngine(mode="strict")
    load(e, p)
    fx, fy, _ = e.forces()
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    assert_bit_equal(fx, ofx, "fx"); assert_bit_equal(fy, ofy, "fy")
    q = p.copy()
    for _ in range(3):
        e.step_brute_force(DT)
        ob.step_brute_force(q, DT, nthreads=8)
    st = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(st[k], q[k], k)


def test_strict_theta_zero_is_brute_force(rx, ob):
    # nbody.rs:197-200
    p = ob.random_disk(500, 9)
    e = rx.NBodyEngine(mode="strict")
    load(e, p)
    e.step_barnes_hut(0.0, DT, 4)
    q = p.copy()
    ob.step_brute_force(q, DT)
    st = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(st[k], q[k], k)


def test_strict_coincident_and_denormal_inputs(rx, ob):
    p = ob.particles([1.0, 1.0, 5.0, 1e-39, 0.0], [2.0, 2.0, 5.0, 0.0, 1e-40], [0] * 5, [0] * 5,
                     [3.0, 4.0, 1.0, 1e-30, 2.0])
    e = rx.NBodyEngine(mode="strict")
    load(e, p)
    fx, fy, _ = e.forces()
    ofx, ofy = ob.brute_forces(p)
    assert_bit_equal(fx, ofx); assert_bit_equal(fy, ofy)


def test_strict_full_size_slice_65536(rx, ob):
    """BASELINE config #2 size: compare a 2048-target slice of a 65536-body Plummer (z=0) bit for bit."""
    st = rx.plummer_sphere(65536, dim=2)
    p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e = rx.NBodyEngine(mode="strict")
    load(e, p)
    fx, fy, _ = e.forces()
    ofx, ofy = ob.brute_forces(p, 0, 2048, nthreads=8)
    assert_bit_equal(fx[:2048], ofx); assert_bit_equal(fy[:2048], ofy)
    ofx, ofy = ob.brute_forces(p, 65536 - 64, 65536)
    assert_bit_equal(fx[-64:], ofx); assert_bit_equal(fy[-64:], ofy)


@pytest.mark.parametrize("kernel", [1, 8, 16])
@pytest.mark.parametrize("n", [1, 2, 3, 55, 56, 57, 59, 60, 61, 63, 64, 65, 111, 112, 113, 119, 120, 121, 255, 256, 257, 1000, 3333])
def test_every_strict_kernel_is_bit_exact(rx, ob, kernel, n):
    """NBX_OPT_STRICT_KERNEL: workgroups of 16 or 8 waves per 64 targets (term producers handing 60- / 56-source chunks to one
    summing wave) and one thread per body all give the oracle's bits -- at sizes on every side of the chunk sizes, the target
    tile and the 256-record padding."""
    p = ob.random_disk(n, 100 + n)
    e = rx.NBodyEngine(mode="strict")
    e.set_strict_kernel(kernel)
    load(e, p)
    fx, fy, _ = e.forces()
    assert e.last_launch()["variant"] == -kernel
    ofx, ofy = ob.brute_forces(p)
    assert_bit_equal(fx, ofx, "fx"); assert_bit_equal(fy, ofy, "fy")
    q = p.copy()
    for _ in range(2):
        e.step_brute_force(DT)
        ob.step_brute_force(q, DT)
    st = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(st[k], q[k], k)


@pytest.mark.parametrize("kernel", [1, 8, 16])
def test_every_strict_kernel_on_every_slab(rx, ob, kernel):
    """the same kernels on a slab of the targets (the sharded layout): 5 ranks' slabs of 4001 bodies stitch to the oracle's
    forces; the slabs start in the middle of chunks and target tiles"""
    n, world = 4001, 5
    p = ob.stable_orbits(n, 0.5, 30.0, 12)
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    for r in range(world):
        e = rx.NBodyEngine(mode="strict")
        e.set_strict_kernel(kernel)
        e.set_shard(r, world)
        load(e, p)
        lo, hi = e.slab()
        fx, fy, _ = e.forces()
        assert e.last_launch()["variant"] == -kernel
        assert_bit_equal(fx[lo:hi] if len(fx) == n else fx, ofx[lo:hi], f"rank {r} fx")
        assert_bit_equal(fy[lo:hi] if len(fy) == n else fy, ofy[lo:hi], f"rank {r} fy")


def test_strict_kernel_chosen_by_targets_per_gpu(rx):
    """16 waves per workgroup while the 64-target workgroups fit the CUs, 8 up to ~120 000 targets, one thread per body
    beyond when its waves fill the SIMDs evenly (profiles/r02_strict_kernel_sweep.txt)"""
    for n, want in ((10000, -16), (16384, -16), (16385, -8), (65536, -8), (131072, -1), (163840, -8), (262144, -1)):
        st = rx.plummer_sphere(n, dim=2)
        e = rx.NBodyEngine(mode="strict")
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
        e.forces()
        assert e.last_launch()["variant"] == want, (n, e.last_launch())
    with pytest.raises(Exception):
        e.set_strict_kernel(4)


# ---------------------------------------------------------------- fast = stated tolerance
def rel_err(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", BRUTE)
def test_fast_matches_golden_within_tolerance(rx, ob, name):
    g = golden(name)
    e = rx.NBodyEngine(mode="fast")
    e.set_particles(g["in_px"], g["in_py"], g["in_vx"], g["in_vy"], g["in_m"])
    p0 = ob.particles(g["in_px"], g["in_py"], g["in_vx"], g["in_vy"], g["in_m"])
    tp1, tol_v1 = fast_tolerances(ob, p0, float(g["dt"]), 1)      # SURVEY 8(d) bounds from this case's own max|a|
    tp10, tol_v10 = fast_tolerances(ob, p0, float(g["dt"]), 10)
    done = 0
    for s, tp, tv in ((1, tp1, tol_v1), (10, tp10, tol_v10)):
        while done < s:
            e.step_brute_force(float(g["dt"]))
            done += 1
        st = e.get_particles()
        for k, tol in (("px", tp), ("py", tp), ("vx", tv), ("vy", tv)):
            err = np.abs(st[k].astype(np.float64) - g[f"s{s}_{k}"]).max()
            assert err <= tol, f"{name} step {s} {k}: {err} > {tol}"


@pytest.mark.parametrize("bpt", [2, 4])
@pytest.mark.parametrize("jsplit", [1, 3, 8])
def test_fast_accelerations_all_launch_shapes(rx, ob, bpt, jsplit):
    """The LDS-tile sweep (variant 1: the default below 16 384 sources) at every launch shape.  (Rounds 1-4 also swept the
    measured losers 0, 2, 3, 4, 5 here -- 54 cases; removed with those kernels in round 5.)"""
    variant = 1
    p = ob.stable_orbits(4096 + 37, 0.5, 30.0, 11)
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    e = rx.NBodyEngine(mode="fast")
    load(e, p)
    e.set_launch(jsplit=jsplit, bodies_per_thread=bpt, variant=variant)
    fx, fy, fz = e.forces()
    ll = e.last_launch()
    assert ll["jsplit"] == jsplit and ll["bodies_per_thread"] == bpt and ll["dim"] == 2
    assert rel_err(fx, ofx) <= 1e-5 and rel_err(fy, ofy) <= 1e-5
    assert not fz.any()
    # the 3-D kernel with z == 0 must reduce to the 2-D law (dz*dz = +0, s*dz = 0)
    e.set_launch(jsplit=jsplit, bodies_per_thread=bpt, dim=3, variant=variant)
    gx, gy, gz = e.forces()
    assert e.last_launch()["dim"] == 3
    assert rel_err(gx, ofx) <= 1e-5 and rel_err(gy, ofy) <= 1e-5 and not gz.any()


@pytest.mark.parametrize("n", [1, 63, 256, 4096 + 37, 20000])
@pytest.mark.parametrize("jsplit", [0, 1, 3, 8])
def test_wave_split_variants_6_and_7(rx, ob, n, jsplit):
    """k_force_smem_pkw: the four waves of a workgroup share 256 targets and split the source range; partial sums meet in
    LDS in wave order. Variant 6 = general masses; variant 7 = unit-mass sweep (a = m * sum d/(r^2+eps), no per-pair
    multiply), selected only when every mass is equal -- otherwise the request silently runs variant 6.  The unit-mass
    sweep ends at the true body count (padding records would otherwise attract)."""
    p = ob.random_disk(n, 19)
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    scale = max(np.abs(ofx).max(), np.abs(ofy).max(), 1e-30)
    e = rx.NBodyEngine(mode="fast")
    for variant, dim in ((6, 2), (6, 3), (7, 2)):
        load(e, p)
        e.set_launch(jsplit=jsplit, dim=dim, variant=variant)
        fx, fy, fz = e.forces()
        ll = e.last_launch()
        # random masses: a request for 7 runs 6 (a single body trivially has "equal masses")
        assert ll["variant"] == (7 if (variant == 7 and n == 1) else 6) and ll["dim"] == dim and ll["bodies_per_thread"] == 4
        assert (jsplit == 0 or ll["jsplit"] == min(jsplit, (n + 255) // 256)) and ll["grid"] == ll["jsplit"] * ((n + 255) // 256)
        assert np.abs(fx - ofx).max() <= 1e-5 * scale and np.abs(fy - ofy).max() <= 1e-5 * scale and not fz.any()
    # equal masses: the unit-mass sweep really runs, for 2-D and 3-D, and a full step matches the oracle
    q = p.copy()
    q["m"][:] = np.float32(0.37)
    ofx, ofy = ob.brute_forces(q, nthreads=8)
    scale = max(np.abs(ofx).max(), np.abs(ofy).max(), 1e-30)
    for dim in (2, 3):
        load(e, q)
        e.set_launch(jsplit=jsplit, dim=dim, variant=7)
        fx, fy, _ = e.forces()
        assert e.last_launch()["variant"] == 7
        assert np.abs(fx - ofx).max() <= 1e-5 * scale and np.abs(fy - ofy).max() <= 1e-5 * scale, (n, dim)
    ptol, vtol = fast_tolerances(ob, q, DT, 1) if n > 1 else (1e-5, 1e-30)
    e.step_brute_force(DT)
    r = q.copy(); ob.step_brute_force(r, DT, nthreads=8)
    st = e.get_particles()
    assert np.abs(st["px"] - r["px"]).max() <= ptol and np.abs(st["vx"] - r["vx"]).max() <= vtol + 1e-30


def test_unit_mass_sweep_in_3d_against_fp64(rx):
    """Variant 7 on a real 3-D equal-mass system (the bench workload's shape) against an fp64 sum on a sample of targets."""
    st = rx.plummer_sphere(8192)
    e = rx.NBodyEngine()
    e.set_launch(variant=7)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    assert e.last_launch() == dict(e.last_launch(), variant=7, dim=3)
    P = np.stack([st["px"], st["py"], st["pz"]], 1).astype(np.float64)
    m = st["m"].astype(np.float64)
    idx = np.arange(0, 8192, 61)
    d = P[None, :, :] - P[idx, None, :]
    w = m[None, :] / ((d * d).sum(2) + 1e-4 * (1 + 0 * m[None, :]))   # EPS = 1e-4 (f32 0.0001 differs in the 9th digit)
    F = (w[:, :, None] * d).sum(1) * m[idx, None]
    got = np.stack([fx, fy, fz], 1)[idx]
    assert np.abs(got - F).max() <= 1e-5 * np.abs(F).max()


def _with_exceptions(p, k, seed):
    """All bodies at mass 0.37 except k of them (a 1000-mass 'sun' first, then heavier and LIGHTER bodies), at positions
    that include the first and the last index."""
    q = p.copy()
    q["m"][:] = np.float32(0.37)
    n = len(q)
    if k == 0:
        return q, []
    rng = np.random.default_rng(seed)
    where = [0, n - 1][:k] + list(rng.choice(np.arange(1, n - 1), size=max(0, k - 2), replace=False))
    vals = ([1000.0, 0.01] + list(rng.uniform(0.05, 40.0, size=max(0, k - 2))))[:k]
    for i, v in zip(where, vals):
        q["m"][i] = np.float32(v)
    return q, sorted(where)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("k", [0, 1, 17])
def test_unit_mass_sweep_with_exceptional_masses(rx, ob, k, dim):
    """'One common mass + a handful of exceptions' -- the shape of the reference's own nb_stable_orbits (unit planets + a
    1000-mass sun, nbody.rs:85-102): the unit-mass sweep (variant 7) runs over weightless sources at the common mass and
    K2 / the force readout add the k exceptional sources with weight m_j - m_common.  Forces and one step against the
    oracle, 0 / 1 / 17 exceptions."""
    n = 20000
    q, where = _with_exceptions(ob.random_disk(n, 23), k, 5)
    ofx, ofy = ob.brute_forces(q, nthreads=8)
    scale = max(np.abs(ofx).max(), np.abs(ofy).max())
    e = rx.NBodyEngine(mode="fast")
    load(e, q)
    e.set_launch(dim=dim)                       # default kernel choice: >= 16 384 sources -> the wave-split sweep
    fx, fy, fz = e.forces()
    ll = e.last_launch()
    assert ll["variant"] == 7 and ll["dim"] == dim, ll
    assert np.abs(fx - ofx).max() <= 1e-5 * scale and np.abs(fy - ofy).max() <= 1e-5 * scale and not fz.any()
    ptol, vtol = fast_tolerances(ob, q, DT, 1)
    e.step_brute_force(DT)
    r = q.copy(); ob.step_brute_force(r, DT, nthreads=8)
    st = e.get_particles()
    for kk in ("px", "py"):
        assert np.abs(st[kk] - r[kk]).max() <= ptol, (k, kk)
    for kk in ("vx", "vy"):
        assert np.abs(st[kk] - r[kk]).max() <= vtol, (k, kk)
    # the general-mass kernel on the same system agrees with the corrected sweep to the same bound
    e6 = rx.NBodyEngine(mode="fast")
    load(e6, q)
    e6.set_launch(dim=dim, variant=6)
    gx, gy, _ = e6.forces()
    assert e6.last_launch()["variant"] == 6
    assert np.abs(gx - fx).max() <= 2e-5 * scale and np.abs(gy - fy).max() <= 2e-5 * scale


def test_unit_mass_sweep_exceptions_in_3d_and_on_slabs(rx):
    """The same with real z coordinates (fp64 arbiter) and on every slab of a 3-way shard (targets in a slab, exceptional
    sources anywhere)."""
    from conftest import fp64_forces_sample

    n = 24576
    st = rx.plummer_sphere(n)
    m = st["m"].copy()
    exc = [0, 5000, 12288, 20000, n - 1]
    m[exc] = np.float32([1000.0, 3.0, 1e-3, 77.0, 0.5])
    st = dict(st, m=m)
    idx = np.arange(0, n, 97)
    F = fp64_forces_sample(st, idx)
    for world in (1, 3):
        for rank in range(world):
            e = rx.NBodyEngine()
            e.set_shard(rank, world)
            e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
            fx, fy, fz = e.forces()
            assert e.last_launch()["variant"] == 7 and e.last_launch()["dim"] == 3
            lo, hi = e.slab()
            sel = (idx >= lo) & (idx < hi)
            got = np.stack([fx, fy, fz], 1)[idx[sel] - lo]
            assert np.abs(got - F[sel]).max() <= 1e-5 * np.abs(F).max(), (world, rank)


def test_too_many_exceptions_run_the_general_kernel(rx, ob):
    """More exceptional masses than the cap (max(32, n/1024), at most n/64): no common mass, variant 6."""
    n = 20000
    q, _ = _with_exceptions(ob.random_disk(n, 29), 40, 7)
    e = rx.NBodyEngine(mode="fast")
    load(e, q)
    fx, fy, _ = e.forces()
    assert e.last_launch()["variant"] == 6
    ofx, ofy = ob.brute_forces(q, nthreads=8)
    assert np.abs(fx - ofx).max() <= 1e-5 * max(np.abs(ofx).max(), np.abs(ofy).max())


def test_stable_orbits_preset_takes_the_unit_mass_sweep(rx, ob):
    """nb_stable_orbits itself (seeded): 19 999 unit planets + the sun -> variant 7 with one exception, step == oracle
    within the stated tolerance."""
    e = rx.NBodyEngine(mode="fast")
    e.seed(4)
    e.stable_orbits(20000, 0.5, 30.0)
    s0 = e.get_particles()
    p = ob.particles(s0["px"], s0["py"], s0["vx"], s0["vy"], s0["m"])
    e.step_brute_force(DT)
    assert e.last_launch()["variant"] == 7
    ptol, vtol = fast_tolerances(ob, p, DT, 1)
    ob.step_brute_force(p, DT, nthreads=8)
    st = e.get_particles()
    assert np.abs(st["px"] - p["px"]).max() <= ptol and np.abs(st["py"] - p["py"]).max() <= ptol
    assert np.abs(st["vx"] - p["vx"]).max() <= vtol and np.abs(st["vy"] - p["vy"]).max() <= vtol


def test_removed_variants_are_refused(rx, ob):
    """Round 5 removed the K1 variants that every A/B lost (0, 2, 3, 4, 5) and one target per thread: asking for them is an
    error, not a silent substitution."""
    e = rx.NBodyEngine(mode="fast")
    for v in (0, 2, 3, 4, 5, 8):
        with pytest.raises(Exception):
            e.set_launch(variant=v)
    with pytest.raises(Exception):
        e.set_launch(bodies_per_thread=1)
    p = ob.random_disk(4096, 17)
    load(e, p)
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    for v, b in ((1, 2), (1, 4), (6, 0), (7, 0)):
        e.set_launch(bodies_per_thread=b, variant=v)
        fx, fy, _ = e.forces()
        assert rel_err(fx, ofx) <= 1e-5 and rel_err(fy, ofy) <= 1e-5, (v, b)


def test_fast_error_vs_fp64_no_worse_than_twice_the_f32_oracle(rx, ob):
    p = ob.stable_orbits(4096, 0.5, 30.0, 12)
    dfx, dfy = ob.brute_forces_f64(p)
    ofx, ofy = ob.brute_forces(p, nthreads=8)
    e = rx.NBodyEngine(mode="fast")
    load(e, p)
    fx, fy, _ = e.forces()
    scale = np.abs(dfx).max()
    err_gpu = max(np.abs(fx - dfx).max(), np.abs(fy - dfy).max()) / scale
    err_cpu = max(np.abs(ofx - dfx).max(), np.abs(ofy - dfy).max()) / scale
    assert err_gpu <= 2.0 * err_cpu + 1e-7, (err_gpu, err_cpu)


def test_fast_deterministic_run_to_run(rx, ob):
    p = ob.random_disk(5000, 13)
    out = []
    for _ in range(2):
        e = rx.NBodyEngine(mode="fast")
        load(e, p)
        for _ in range(5):
            e.step_brute_force(DT)
        out.append(e.get_particles())
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(out[0][k], out[1][k], k)


def test_fast_3d_against_fp64(rx):
    st = rx.plummer_sphere(8192)
    e = rx.NBodyEngine(mode="fast")
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    assert e.last_launch()["dim"] == 3
    P = np.stack([st["px"], st["py"], st["pz"]], 1).astype(np.float64)
    m = st["m"].astype(np.float64)
    idx = np.arange(0, 8192, 16)
    d = P[None, :, :] - P[idx, None, :]
    w = m[None, :] / ((d * d).sum(-1) + 1e-4)
    F = (w[:, :, None] * d).sum(1) * m[idx, None]
    scale = np.abs(F).max()
    got = np.stack([fx[idx], fy[idx], fz[idx]], 1)
    assert np.abs(got - F).max() / scale <= 1e-5


def test_fast_vs_strict_at_full_size_262144(rx, ob):
    """BASELINE headline size (a CPU step here takes minutes): the fast kernel against the bit-exact
    kernel on the same GPU, both against an fp64 sample, plus Newton's third law as a
    size-independent check.  Tolerance 1e-5 * max(1, sqrt(N)/64): sequential f32 summation error
    grows ~sqrt(N) and the strict (reference-order) sum is the LESS accurate of the two."""
    n = 262144
    st = rx.plummer_sphere(n, dim=2)
    fast = rx.NBodyEngine(mode="fast")
    strict = rx.NBodyEngine(mode="strict")
    for e in (fast, strict):
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    fx, fy, _ = fast.forces()
    sx, sy, _ = strict.forces()
    tol = 1e-5 * max(1.0, np.sqrt(n) / 64.0)
    assert rel_err(fx, sx) <= tol and rel_err(fy, sy) <= tol
    # fp64 arbiter on a 512-target sample: the fast kernel is at least as accurate as the reference order
    p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    dfx, dfy = ob.brute_forces_f64(p, 1000, 1512)
    scale = np.abs(sx).max()
    err_fast = max(np.abs(fx[1000:1512] - dfx).max(), np.abs(fy[1000:1512] - dfy).max()) / scale
    err_strict = max(np.abs(sx[1000:1512] - dfx).max(), np.abs(sy[1000:1512] - dfy).max()) / scale
    assert err_fast <= 1e-5 and err_fast <= 2.0 * err_strict + 1e-7, (err_fast, err_strict)
    # and the strict slice is the oracle's, bit for bit, at this size too
    ofx, ofy = ob.brute_forces(p, 1000, 1064)
    assert_bit_equal(sx[1000:1064], ofx); assert_bit_equal(sy[1000:1064], ofy)
    # sum of all internal forces vanishes (antisymmetry, nbody.rs:174-183)
    tot = np.array([fx.astype(np.float64).sum(), fy.astype(np.float64).sum()])
    assert np.all(np.abs(tot) <= 1e-6 * np.abs(fx.astype(np.float64)).sum())


def test_step_sequence_mixed_with_state_io(rx, ob):
    """set -> step -> get -> set -> step keeps host mirror and device state coherent."""
    p = ob.random_disk(777, 14)
    e = rx.NBodyEngine(mode="strict")
    load(e, p)
    e.step_brute_force(DT)
    mid = e.get_particles()
    e2 = rx.NBodyEngine(mode="strict")
    e2.set_particles(mid["px"], mid["py"], mid["vx"], mid["vy"], mid["m"])
    e.step_brute_force(DT); e2.step_brute_force(DT)
    a, b = e.get_particles(), e2.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(a[k], b[k], k)
    fb = e.draw(64, 64)
    q = p.copy(); ob.step_brute_force(q, DT); ob.step_brute_force(q, DT)
    assert np.array_equal(fb, ob.draw(q, 64, 64))


def test_empty_and_single_body(rx):
    e = rx.NBodyEngine()
    e.set_particles([], [], [], [], [])
    e.step_brute_force(DT)
    assert e.num_particles() == 0
    e.set_particles([1.0], [2.0], [0.5], [-0.5], [3.0])
    e.step_brute_force(DT)
    st = e.get_particles()
    assert st["vx"][0] == np.float32(0.5) and st["px"][0] == np.float32(1.0) + np.float32(DT) * np.float32(0.5)


def test_level1_surface_on_gpu(rx, ob):
    """The six reference symbols exactly as RustNBodyExperiment.hs drives them (process-global)."""
    rx.nb_stable_orbits(1024, 0.5, 30.0)
    assert rx.nb_num_particles() == 1024
    fb0 = rx.nb_draw(256, 256)
    rx.nb_step_brute_force(DT)
    rx.nb_step_barnes_hut(0.0, DT, 1)
    rx.nb_step_barnes_hut(0.85, DT, 1)
    fb1 = rx.nb_draw(256, 256)
    assert rx.nb_num_particles() == 1024
    assert (fb1 == 0x00FF00FF).sum() == 5 and not np.array_equal(fb0, fb1)


def test_profile_records_kernel_time(rx, ob):
    p = ob.random_disk(8192, 15)
    e = rx.NBodyEngine()
    load(e, p)
    e.profile(True)
    for _ in range(4):
        e.step_brute_force(DT)
    ms, cnt = e.profile_read(rx.NBX_K_FORCE)
    assert cnt == 4 and ms > 0
    ms2, cnt2 = e.profile_read(rx.NBX_K_INTEGRATE)
    assert cnt2 == 4 and ms2 > 0 and ms2 < ms


@pytest.mark.parametrize("kernel", [1, 8, 16])
@pytest.mark.parametrize("case", ["edge_of_fast_range", "tiny_masses", "huge_masses", "far_coordinates", "zero_mass", "mixed_16k"])
def test_strict_division_paths_are_bit_exact(rx, ob, case, kernel):
    """The bit-exact kernel divides with the short exact sequence only when the launch has proven that no pair needs the
    scaling / fix-up steps of the IEEE expansion (masses in [1e-10, 1e10], |coordinates| <= 1e5), and with the
    compiler's full expansion otherwise. Both must equal the oracle bit for bit: at the edge of the fast range (mass
    products 1e-20 .. 1e20, separations up to 2.8e5) and on every side of it."""
    rng = np.random.default_rng(31)
    n = 16384 if case == "mixed_16k" else 3000
    x = rng.uniform(-40, 40, n).astype(np.float32)
    y = rng.uniform(-40, 40, n).astype(np.float32)
    m = rng.uniform(0.5, 2.0, n).astype(np.float32)
    if case in ("edge_of_fast_range", "mixed_16k"):
        m[: n // 3] = 1e-10
        m[n // 3: 2 * n // 3] = 1e10
        x[::7] = rng.uniform(-1e5, 1e5, len(x[::7])).astype(np.float32)
        y[::11] = rng.uniform(-1e5, 1e5, len(y[::11])).astype(np.float32)
        x[5] = 1e5; y[5] = -1e5; x[6] = -1e5; y[6] = 1e5          # the largest separation the guard admits
        x[7] = x[8]; y[7] = np.nextafter(y[8], np.float32(1e9))    # and the smallest: one ulp apart
    elif case == "tiny_masses":
        m[::3] = 1e-15; m[1::3] = 3e-38
    elif case == "huge_masses":
        m[::3] = 1e12; m[1::3] = 1e18
    elif case == "far_coordinates":
        x[::5] *= 1e4; y[::9] *= 1e5
    else:
        m[::4] = 0.0
    p = ob.particles(x, y, rng.normal(0, 1, n), rng.normal(0, 1, n), m)
    e = rx.NBodyEngine(mode="strict")
    e.set_strict_kernel(kernel)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    fx, fy, _ = e.forces(0.0)
    wx, wy = ob.brute_forces(p, 0, n, nthreads=8)
    assert_bit_equal(fx, wx, case + " fx"); assert_bit_equal(fy, wy, case + " fy")
    q = p.copy()
    for _ in range(2):
        e.step_brute_force(DT)
        ob.step_brute_force(q, DT)
    st = e.get_particles()
    with np.errstate(invalid="ignore"):
        for k in ("px", "py", "vx", "vy"):
            assert np.array_equal(st[k].view(np.uint32), q[k].view(np.uint32)), f"{case} {k}"


def test_mode_switch_on_a_live_engine_and_half_source_size(rx, ob):
    """NBX_OPT_FORCE_MODE may change between steps of one engine: each step then equals the same step of an engine
    that was in that mode all along (fast within tolerance, strict bit for bit); the fp16 source copy reports
    8 bytes per padded body."""
    p = ob.stable_orbits(5000, 0.5, 30.0, 12)
    e = rx.NBodyEngine(mode="fast")
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    e.set_mode("strict")
    e.step_brute_force(DT)
    q = p.copy(); ob.step_brute_force(q, DT)
    st = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        assert_bit_equal(st[k], q[k], k)
    e.set_mode("fast")
    e.step_brute_force(DT)
    ptol, vtol = fast_tolerances(ob, q, DT, 1)      # from the state this step starts from
    ob.step_brute_force(q, DT)
    st = e.get_particles()
    assert np.abs(st["px"] - q["px"]).max() <= ptol * max(1.0, np.abs(q["px"]).max())
    assert np.abs(st["vx"] - q["vx"]).max() <= vtol
    e.set_source_precision(16)
    e.step_brute_force(DT)
    assert e.half_sources_bytes() == ((5000 + 255) // 256) * 256 * 8
    from rust_exp_amd.engine import NBX_OPT_FORCE_MODE, NBX_OPT_SOURCE_PRECISION
    assert e.get_option(NBX_OPT_FORCE_MODE) == 0 and e.get_option(NBX_OPT_SOURCE_PRECISION) == 16


def test_default_launch_shapes(rx, ob):
    """The launch heuristic at the sizes the measurements were made on (profiles/r02_k1_wave_split_sweep.txt,
    r02_small_n_variants.txt): a change of these defaults should be a decision, not an accident."""
    def shape(n, equal_mass=True, shard=None, dim=3):
        st = rx.plummer_sphere(n, dim=dim)
        m = st["m"] if equal_mass else np.linspace(0.5, 2.0, n).astype(np.float32)
        e = rx.NBodyEngine()
        if shard:
            e.set_shard(0, shard)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], m, st["pz"], st["vz"])
        e.forces()
        ll = e.last_launch()
        return ll["variant"], ll["jsplit"], ll["grid"], ll["dim"]

    assert shape(262144) == (7, 8, 8192, 3)                       # headline: unit-mass wave-split sweep, 8 partial slabs
    assert shape(262144, equal_mass=False) == (6, 8, 8192, 3)     # unequal masses: the same kernel with the m_j multiply
    assert shape(65536) == (7, 64, 16384, 3)                      # config #2
    assert shape(262144, shard=8) == (7, 64, 8192, 3)             # one GPU of config #3: 32 768 targets keep the chip full
    assert shape(16384, dim=2) == (7, 16, 1024, 2)                # first size on the wave-split kernel
    assert shape(8192)[0] == 1 and shape(4096) == (1, 16, 128, 3)  # LDS tiles below; tiny systems: one tile per workgroup
