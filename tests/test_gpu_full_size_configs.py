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


def test_config3_262144_eight_slabs_stitch_to_the_unsharded_step(rx):
    n, world = 262144, 8
    st = rx.plummer_sphere(n)
    ref = rx.NBodyEngine()
    ref.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    # tolerance from the data; at this size max|a| comes from the engine's own force evaluation (the oracle
    # needs minutes for 6.9e10 pairs) -- it only scales the bound. Two fast results are compared, each within the bound
    # of the oracle: hence the factor 2.
    fx, fy, fz = ref.forces()
    amax = float(np.max(np.sqrt(fx.astype(np.float64) ** 2 + fy.astype(np.float64) ** 2 + fz.astype(np.float64) ** 2) / st["m"]))
    ptol, vtol = fast_tolerances(None, None, 0.01, 1, amax=amax, n=n)
    ref.step_brute_force(0.01)
    want = ref.get_particles()
    for r in (0, 3, 7):      # first, middle, last slab (each is 32 768 targets x 262 144 sources)
        e = rx.NBodyEngine()
        e.set_shard(r, world)
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
        e.step_local(0.01)
        lo, hi = e.slab()
        assert (lo, hi) == (r * 32768, (r + 1) * 32768)
        got = e.get_particles()
        for k in ("px", "py", "pz"):
            assert np.abs(got[k][lo:hi] - want[k][lo:hi]).max() <= 2 * ptol, (r, k)
        for k in ("vx", "vy", "vz"):
            assert np.abs(got[k][lo:hi] - want[k][lo:hi]).max() <= 2 * vtol, (r, k, vtol)
        ll = e.last_launch()
        assert ll["grid"] >= 2048 and ll["dim"] == 3      # the j-split keeps the chip full with 32 768 targets


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
    from rust_exp_amd.engine import NBX_OPT_BH_LAST_TREE
    assert d.get_option(NBX_OPT_BH_LAST_TREE) == 1
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
