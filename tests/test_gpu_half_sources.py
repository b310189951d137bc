"""GPU parity for BASELINE config #5's numerics: fp16 SOURCE copy, fp32 targets/accumulators/state.
Tolerance class: (a) against an fp64 model that uses the SAME fp16-rounded sources
the kernel is exact to fp32 rounding (1e-5 of max|F|); (b) against the true fp32 result the error is that
of quantising source coordinates to 11 bits: median relative acceleration error < 5e-3."""
import numpy as np
import pytest

from conftest import fast_tolerances

pytestmark = pytest.mark.gpu


def model_f64(st, idx, dim):
    """F_i = m_i * sum_{j != i} m_j^h (p_j^h - p_i) / (|p_j^h - p_i|^2 + eps), sources rounded to fp16."""
    P = np.stack([st["px"], st["py"], st["pz"]], 1).astype(np.float64)
    Ph = np.stack([st["px"], st["py"], st["pz"]], 1).astype(np.float16).astype(np.float64)
    mh = st["m"].astype(np.float16).astype(np.float64)
    if dim == 2:
        P[:, 2] = 0; Ph[:, 2] = 0
    d = Ph[None, :, :] - P[idx, None, :]
    w = mh[None, :] / ((d * d).sum(-1) + 1e-4)
    w[np.arange(len(idx)), idx] = 0.0            # self term excluded (the kernel subtracts it)
    return (w[:, :, None] * d).sum(1) * st["m"].astype(np.float64)[idx, None]


@pytest.mark.parametrize("kind,n,bpt,jsplit", [("plummer", 8192, 2, 0), ("plummer", 8192, 4, 3), ("galaxies", 6000, 2, 5),
                                               ("galaxies", 6000, 4, 1)])
def test_half_sources_exact_to_fp32_rounding_against_fp64_model(rx, kind, n, bpt, jsplit):
    st = rx.plummer_sphere(n) if kind == "plummer" else rx.two_galaxies(n)
    dim = 3 if kind == "plummer" else 2
    e = rx.NBodyEngine()
    e.set_source_precision(16)
    e.set_launch(jsplit=jsplit, bodies_per_thread=bpt)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    assert e.last_launch()["dim"] == dim
    idx = np.arange(0, n, 7)
    F = model_f64(st, idx, dim)
    got = np.stack([fx[idx], fy[idx], fz[idx]], 1)
    assert np.isfinite(got).all()
    assert np.abs(got - F).max() <= 1e-5 * np.abs(F).max()


@pytest.mark.parametrize("kind,n,jsplit,masses", [("plummer", 20000, 0, "equal"), ("plummer", 16384, 5, "random"),
                                                  ("galaxies", 24576, 0, "preset"), ("galaxies", 20001, 3, "preset")])
def test_half_sources_on_the_wave_split_kernels_against_fp64_model(rx, kind, n, jsplit, masses):
    """Round 3: from 16 384 sources on, K4 runs on the wave-split kernels -- the half4 copy widened to
    float4 once per step, sources through the scalar cache, every target's interaction with its own fp16 image taken out
    again; 'one common mass + exceptions' (the two galaxy cores) takes the unit-mass sweep (variant 18), anything else the
    general one (17).  Same bound as the LDS-tile K4 kernel: exact to fp32 rounding against an fp64 model that uses the
    SAME fp16-rounded sources."""
    st = rx.plummer_sphere(n) if kind == "plummer" else rx.two_galaxies(n)
    if masses == "random":
        st = dict(st, m=np.random.default_rng(3).uniform(0.1, 1.5, n).astype(np.float32))
    dim = 3 if kind == "plummer" else 2
    e = rx.NBodyEngine()
    e.set_source_precision(16)
    e.set_launch(jsplit=jsplit)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    ll = e.last_launch()
    assert ll["dim"] == dim and ll["variant"] == (17 if masses == "random" else 18), ll
    idx = np.arange(0, n, 41)
    if kind == "galaxies":
        idx = np.unique(np.r_[idx, 0, n // 2])          # the two cores: exceptional sources AND targets
    F = model_f64(st, idx, dim)
    got = np.stack([fx[idx], fy[idx], fz[idx]], 1)
    assert np.isfinite(got).all()
    assert np.abs(got - F).max() <= 1e-5 * np.abs(F).max()
    # a step on it stays finite and moves the bodies; the fp32 state keeps its masses
    e.step_brute_force(0.01)
    p = e.get_particles()
    assert np.isfinite(p["px"]).all() and np.array_equal(p["m"], st["m"])


def test_half_sources_accuracy_class_vs_fp32(rx):
    st = rx.two_galaxies(32768)
    a = rx.NBodyEngine()
    a.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    b = rx.NBodyEngine()
    b.set_source_precision(16)
    b.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    fx, fy, _ = a.forces()
    hx, hy, _ = b.forces()
    rel = np.hypot(hx - fx, hy - fy) / (np.hypot(fx, fy) + 1e-20)
    assert np.median(rel) < 5e-3
    # and a short run stays close in position (10 steps, dt 0.01)
    for _ in range(10):
        a.step_brute_force(0.01); b.step_brute_force(0.01)
    pa, pb = a.get_particles(), b.get_particles()
    assert np.median(np.hypot(pa["px"] - pb["px"], pa["py"] - pb["py"])) < 5e-2   # measured 8e-3 (orbital speed ~32)
    # masses and the fp32 state itself are untouched by the fp16 copy
    assert np.array_equal(pb["m"], st["m"])


def test_half_sources_sharded_slabs_stitch(rx, ob):
    """Every rank's nbx_step_local in fp16-source mode (own slab fp32, sources = fp16 copy of all bodies)
    reproduces the unsharded fp16-source step to fp32 rounding."""
    n, world = 16384, 8
    st = rx.two_galaxies(n)
    ref = rx.NBodyEngine()
    ref.set_source_precision(16)
    ref.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    ref.step_brute_force(0.01)
    want = ref.get_particles()
    # both sides sweep the SAME fp16-rounded sources in fp32: they differ by launch shape only, i.e. by twice the fast
    # mode's bound for this case (from the oracle's max|a| on this state)
    ptol, vtol = fast_tolerances(ob, ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"]), 0.01, 1)
    for r in range(world):
        e = rx.NBodyEngine()
        e.set_source_precision(16)
        e.set_shard(r, world)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
        e.step_local(0.01)
        lo, hi = e.slab()
        got = e.get_particles()
        assert np.abs(got["px"][lo:hi] - want["px"][lo:hi]).max() <= 2 * ptol
        assert np.abs(got["vx"][lo:hi] - want["vx"][lo:hi]).max() <= 2 * vtol, vtol


def test_switching_precision_on_a_live_engine(rx):
    st = rx.plummer_sphere(4096)
    e = rx.NBodyEngine()
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    e.step_brute_force(0.01)
    e.set_source_precision(16)          # builds the fp16 copy from the live device state
    e.step_brute_force(0.01)
    e.set_source_precision(32)
    e.step_brute_force(0.01)
    p = e.get_particles()
    assert np.isfinite(p["px"]).all() and np.abs(p["px"] - st["px"]).max() > 0


def test_half_copy_stays_coherent_across_barnes_hut_and_strict_steps(rx):
    """Steps that do not use the fp16 copy (Barnes-Hut, bit-exact brute force) still refresh it, so a later
    fp16-source step sees the current positions."""
    st = rx.two_galaxies(4096)
    a = rx.NBodyEngine(); a.set_source_precision(16)
    a.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    a.step_barnes_hut(0.5, 0.01, 1)
    mid = a.get_particles()
    b = rx.NBodyEngine(); b.set_source_precision(16)
    b.set_particles(mid["px"], mid["py"], mid["vx"], mid["vy"], mid["m"])
    fa = a.forces(); fb = b.forces()
    assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1])
