"""GPU: BASELINE.json's configurations at their FULL sizes, through size-independent properties
(a CPU step at these sizes takes minutes to hours).

  #2  65 536-body Plummer, brute force fp32, 1 GPU      -> test_gpu_brute.py::test_strict_full_size_slice_65536 + here
  #3  262 144 bodies, 8 slabs                            -> every rank's slab kernel stitched on one GPU
  #4  1 048 576 bodies Barnes-Hut theta = 0.5            -> force error vs all-pairs, host tree and device tree
  #5  524 288-body two-galaxy, fp16 sources / fp32 acc.  -> accuracy class vs fp32, self-image correction
"""
import numpy as np
import pytest

from conftest import fast_tolerances

pytestmark = pytest.mark.gpu


def test_config2_plummer_65536_fast_vs_fp64_sample_and_step_invariants(rx, ob):
    st = rx.plummer_sphere(65536)
    e = rx.NBodyEngine()
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    P = np.stack([st["px"], st["py"], st["pz"]], 1).astype(np.float64)
    m = st["m"].astype(np.float64)
    idx = np.arange(0, 65536, 128)
    d = P[None, :, :] - P[idx, None, :]
    w = m[None, :] / ((d * d).sum(-1) + 1e-4)
    F = (w[:, :, None] * d).sum(1) * m[idx, None]
    got = np.stack([fx[idx], fy[idx], fz[idx]], 1)
    assert np.abs(got - F).max() <= 1e-5 * np.abs(F).max()
    # total internal force vanishes (antisymmetry of nbody.rs:174-183) -> total momentum is conserved by a step
    tot = np.array([fx.astype(np.float64).sum(), fy.astype(np.float64).sum(), fz.astype(np.float64).sum()])
    assert np.all(np.abs(tot) <= 1e-6 * np.abs(fx.astype(np.float64)).sum())
    for _ in range(5):
        e.step_brute_force(0.01)
    p = e.get_particles()
    mom = (p["m"].astype(np.float64)[:, None] * np.stack([p["vx"], p["vy"], p["vz"]], 1)).sum(0)
    assert np.all(np.abs(mom) <= 1e-4 * (p["m"].astype(np.float64) * np.abs(p["vx"])).sum() + 1e-6)


def _sample_targets(lo, hi, count=512):
    return lo + (np.arange(count, dtype=np.int64) * (hi - lo)) // count


def _fp64_forces_fp16_sources(st, idx, chunk=16):
    """F_i = m_i * sum_{j != i} m_j^h (p_j^h - p_i) / (|p_j^h - p_i|^2 + 1e-4) in float64, 2-D: the pair law of nbody.rs:174-183
    with every SOURCE (position and mass) rounded to fp16 as config #5 stores it; targets keep their fp32 state.  [len(idx), 2]."""
    P = np.stack([st["px"], st["py"]], 1).astype(np.float64)
    Ph = np.stack([st["px"], st["py"]], 1).astype(np.float16).astype(np.float64)
    mh = st["m"].astype(np.float16).astype(np.float64)
    m = st["m"].astype(np.float64)
    idx = np.asarray(idx)
    out = np.zeros((len(idx), 2))
    for a in range(0, len(idx), chunk):
        ii = idx[a:a + chunk]
        d = Ph[None, :, :] - P[ii, None, :]
        w = mh[None, :] / ((d * d).sum(-1) + 1e-4)
        w[np.arange(len(ii)), ii] = 0.0          # a body does not attract itself (the kernel takes its own image out again)
        out[a:a + chunk] = (w[:, :, None] * d).sum(1) * m[ii, None]
    return out


@pytest.mark.parametrize("masses", ["equal", "random"])
def test_headline_kernel_at_the_headline_shape_against_fp64(rx, masses):
    """The kernel bench.py times, at the shape it times it: N = 262 144, dim 3, DEFAULT launch =
    k_force_smem_pkw, S = 8, 8 192 workgroups -- variant 7 (every Plummer body has the same mass) and variant 6 (random
    masses, what `general_masses` in the bench line times).  512 targets spread over the whole range x ALL 262 144 sources
    against a float64 sum (nbody.rs:174-183 with z), bound 1e-5 * max|F|."""
    from conftest import fp64_forces_sample

    n = 262144
    st = rx.plummer_sphere(n)
    if masses == "random":
        st = dict(st, m=np.random.default_rng(11).uniform(0.1, 1.5, n).astype(np.float32))   # nb_random_disk's range, nbody.rs:62
    e = rx.NBodyEngine()
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    fx, fy, fz = e.forces()
    ll = e.last_launch()
    assert (ll["variant"], ll["jsplit"], ll["grid"], ll["dim"], ll["block"]) == (7 if masses == "equal" else 6, 8, 8192, 3, 256), ll
    idx = _sample_targets(0, n)
    F = fp64_forces_sample(st, idx)
    got = np.stack([fx[idx], fy[idx], fz[idx]], 1).astype(np.float64)
    err = np.abs(got - F).max() / np.abs(F).max()
    assert err <= 1e-5, (masses, err)
    # and the step built on it: one kick-drift from those accelerations, against the float64 forces of the sample
    e.step_brute_force(0.01)
    assert e.last_launch() == ll
    p = e.get_particles()
    a = F / st["m"][idx].astype(np.float64)[:, None]
    vnew = np.stack([st["vx"][idx], st["vy"][idx], st["vz"][idx]], 1) + 0.01 * a
    pnew = np.stack([st["px"][idx], st["py"][idx], st["pz"][idx]], 1) + 0.01 * vnew
    amax = float(np.max(np.sqrt((a * a).sum(1))))
    ptol, vtol = fast_tolerances(None, None, 0.01, 1, amax=amax, n=n)
    assert np.abs(np.stack([p["vx"][idx], p["vy"][idx], p["vz"][idx]], 1) - vnew).max() <= vtol
    assert np.abs(np.stack([p["px"][idx], p["py"][idx], p["pz"][idx]], 1) - pnew).max() <= ptol


def test_config3_262144_eight_slabs_against_fp64(rx):
    """One GPU's share of config #3 (32 768 targets x 262 144 sources; first, middle and last slab of the reference split
    nbody.rs:426-428) at the launch shape an 8-GPU run uses, each against the float64 sum on 512 of its targets -- an
    external check per slab shape (round 2 compared fast slabs with the fast unsharded step), then the slab's own step."""
    from conftest import fp64_forces_sample

    n, world = 262144, 8
    st = rx.plummer_sphere(n)
    for r in (0, 3, 7):
        e = rx.NBodyEngine()
        e.set_shard(r, world)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
        lo, hi = e.slab()
        assert (lo, hi) == (r * 32768, (r + 1) * 32768)
        fx, fy, fz = e.forces()
        ll = e.last_launch()
        assert (ll["variant"], ll["jsplit"], ll["grid"], ll["dim"]) == (7, 64, 8192, 3), ll   # the j-split keeps the chip full
        idx = _sample_targets(lo, hi)
        F = fp64_forces_sample(st, idx)
        got = np.stack([fx[idx - lo], fy[idx - lo], fz[idx - lo]], 1).astype(np.float64)
        assert np.abs(got - F).max() <= 1e-5 * np.abs(F).max(), r
        e.step_local(0.01)
        p = e.get_particles()
        a = F / st["m"][idx].astype(np.float64)[:, None]
        vnew = 0.01 * a                                     # the Plummer workload starts at rest
        pnew = np.stack([st["px"][idx], st["py"][idx], st["pz"][idx]], 1) + 0.01 * vnew
        ptol, vtol = fast_tolerances(None, None, 0.01, 1, amax=float(np.max(np.sqrt((a * a).sum(1)))), n=n)
        assert np.abs(np.stack([p["vx"][idx], p["vy"][idx], p["vz"][idx]], 1) - vnew).max() <= vtol, r
        assert np.abs(np.stack([p["px"][idx], p["py"][idx], p["pz"][idx]], 1) - pnew).max() <= ptol, r
        # bodies outside the slab are untouched by a local step
        out = np.r_[0:lo, hi:n]
        assert np.array_equal(p["px"][out], st["px"][out])


@pytest.mark.parametrize("tree", ["host", "device"])
def test_config4_barnes_hut_1m_theta_half_force_error(rx, tree):
    n = 1048576
    st = rx.plummer_sphere(n, dim=2)
    e = rx.NBodyEngine()
    e.set_bh_tree(tree)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    bx, by, _ = e.forces(0.5)
    fx, fy, _ = e.forces(0.0)
    rel = np.hypot(bx - fx, by - fy) / (np.hypot(fx, fy) + 1e-20)
    assert np.median(rel) < 1e-3 and np.percentile(rel, 99) < 2e-2
    wk = e.bh_work(0.5)
    assert 100 < wk["node_visits"] / n < 5000              # O(log N / theta^2) per body, not O(N)
    e.step_barnes_hut(0.5, 0.01, 1)
    p = e.get_particles()
    assert np.isfinite(p["px"]).all() and np.abs(p["px"]).max() < 60


def test_config4_barnes_hut_1m_theta_half_against_the_oracle(rx, ob):
    """BASELINE config #4 at its FULL size pinned to the oracle directly (the oracle does a 1 M-body
    Barnes-Hut step in ~1 s on the host cores): the bit-exact mode's step (threaded host build with device-side routing,
    wave-uniform strict walk) equals orc_step_barnes_hut (nbody.rs:186-480) bit for bit on all 1 048 576 bodies; the fast
    mode's forces (host tree, the default) are within 2e-5 * max|F| of orc_bh_forces (nbody.rs:333-377)."""
    n, theta, dt = 1048576, 0.5, 0.01
    st = rx.plummer_sphere(n, dim=2)
    p = ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    rc, ofx, ofy = ob.bh_forces(p, theta, nthreads=16)
    assert rc == 0
    f = rx.NBodyEngine(mode="fast")
    f.set_bh_tree("host")
    f.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    fx, fy, _ = f.forces(theta)
    scale = max(np.abs(ofx).max(), np.abs(ofy).max())
    assert np.abs(fx - ofx).max() <= 2e-5 * scale and np.abs(fy - ofy).max() <= 2e-5 * scale
    # the fast mode's DEFAULT at this size builds the tree on the device (exact node sums, rounded once): against the fp64
    # arbiter it is within the walk's fp32 rounding, and what separates it from the oracle is the oracle's own drift
    # (the reference folds a million masses in f32, nbody.rs:303-320: its root holds 1000.6 for a true 1000.0)
    rc2, ex, ey = ob.bh_forces_exact(p, theta, nthreads=16)
    assert rc2 == 0
    d = rx.NBodyEngine(mode="fast")
    d.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    dx, dy, _ = d.forces(theta)
    from rust_exp_amd.engine import NBX_STAT_BH_LAST_TREE
    assert d.get_stat(NBX_STAT_BH_LAST_TREE) == 1
    dev_arb = np.maximum(np.abs(dx - ex), np.abs(dy - ey)) / scale
    orc_arb = np.maximum(np.abs(ofx - ex), np.abs(ofy - ey)) / scale
    dev_orc = np.maximum(np.abs(dx - ofx), np.abs(dy - ofy)) / scale
    q = lambda a: float(np.percentile(a, 99.9))   # noqa: E731  (the other 0.1 %: flipped opening decisions, see test_gpu_bh_device_tree.py)
    assert q(dev_arb) <= 2e-5 and dev_arb.max() <= 2e-3, (q(dev_arb), dev_arb.max())
    assert q(dev_orc) <= q(orc_arb) + 2e-5 and dev_orc.max() <= orc_arb.max() + 2e-3, (q(dev_orc), q(orc_arb))
    e = rx.NBodyEngine(mode="strict")
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    sx, sy, _ = e.forces(theta)
    assert np.array_equal(sx.view(np.uint32), ofx.view(np.uint32)) and np.array_equal(sy.view(np.uint32), ofy.view(np.uint32))
    q = p.copy()
    for _ in range(2):
        e.step_barnes_hut(theta, dt, 1)
        assert ob.step_barnes_hut(q, theta, dt, 16) == 0
    got = e.get_particles()
    for k in ("px", "py", "vx", "vy"):
        bad = np.nonzero(got[k].view(np.uint32) != q[k].view(np.uint32))[0]
        assert bad.size == 0, (k, bad.size, bad[:5])


def test_config5_two_galaxies_524288_fp16_sources(rx):
    n = 524288
    st = rx.two_galaxies(n)
    a = rx.NBodyEngine()
    a.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    b = rx.NBodyEngine()
    b.set_source_precision(16)
    b.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    fx, fy, _ = a.forces()
    hx, hy, _ = b.forces()
    # EXTERNAL check at full size, at the launch shape bench.py times for this config: the unit-mass sweep over
    # the widened fp16 copy with its two exceptional sources, S = 4, 8 192 workgroups -- against a float64 sum that uses the SAME
    # fp16-rounded sources (tests/test_gpu_half_sources.py's model, chunked), on 512 targets spread over the whole range plus both
    # galaxy cores (exceptional sources AND targets), each against ALL 524 288 sources.  Bound: fp32 rounding, 1e-5 of max|F|.
    ll = b.last_launch()
    assert (ll["variant"], ll["jsplit"], ll["grid"], ll["block"], ll["dim"]) == (18, 4, 8192, 256, 2), ll
    idx = np.unique(np.r_[_sample_targets(0, n), 0, n // 2])
    F = _fp64_forces_fp16_sources(st, idx)
    got = np.stack([hx[idx], hy[idx]], 1).astype(np.float64)
    err = np.abs(got - F).max() / np.abs(F).max()
    assert err <= 1e-5, err
    rel = np.hypot(hx - fx, hy - fy) / (np.hypot(fx, fy) + 1e-20)
    assert np.median(rel) < 5e-3
    # without the self-image correction every body would feel ~50*m of spurious self force: the galaxy cores
    # (m = 1000) would be off by O(1); they are not
    for core in (0, n // 2):
        assert np.hypot(hx[core] - fx[core], hy[core] - fy[core]) <= 2e-2 * np.hypot(fx[core], fy[core])
    # one rank's slab of the 8-way shard in fp16-source mode
    e = rx.NBodyEngine()
    e.set_source_precision(16)
    e.set_shard(5, 8)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e.step_local(0.01)
    b.step_brute_force(0.01)
    lo, hi = e.slab()
    got, want = e.get_particles(), b.get_particles()
    # two fast results (different source splits): each within the SURVEY 8(d) bound computed from this case's max|a|
    amax = float(np.max(np.hypot(fx.astype(np.float64), fy.astype(np.float64)) / st["m"]))
    ptol, vtol = fast_tolerances(None, None, 0.01, 1, amax=amax, n=n)
    assert np.abs(got["px"][lo:hi] - want["px"][lo:hi]).max() <= 2 * ptol, ptol
    assert np.abs(got["vx"][lo:hi] - want["vx"][lo:hi]).max() <= 2 * vtol, vtol
