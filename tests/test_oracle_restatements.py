"""The C oracle's Barnes-Hut step against a second, independent restatement of nbody.rs:186-480 (oracle/nbody_bh_py.py:
recursive Python, numpy.float32 scalars): bit for bit, including EPS merges, the velocity kill box and the panics."""
import numpy as np
import pytest

from oracle import nbody_bh_py as bhpy


def run_both(ob, p, theta, steps):
    q = p.copy()
    a = {k: p[k].astype(np.float32).copy() for k in ("px", "py", "vx", "vy", "m")}
    for _ in range(steps):
        assert ob.step_barnes_hut(q, theta, 0.01, 1) == 0
        bhpy.step_barnes_hut(a["px"], a["py"], a["vx"], a["vy"], a["m"], theta, 0.01)
    for k in ("px", "py", "vx", "vy"):
        assert np.array_equal(q[k].view(np.uint32), a[k].view(np.uint32)), k


@pytest.mark.parametrize("n,theta,steps", [(2, 0.5, 3), (5, 0.85, 3), (64, 0.5, 2), (300, 0.85, 2), (300, 0.3, 1)])
def test_c_oracle_equals_python_restatement_on_presets(ob, n, theta, steps):
    run_both(ob, ob.random_disk(n, 5 + n), theta, steps)
    if n > 1:
        run_both(ob, ob.stable_orbits(n, 0.5, 30.0, 9 + n), theta, steps)


def test_c_oracle_equals_python_restatement_with_merges_and_kill_box(ob):
    rng = np.random.default_rng(3)
    x = rng.uniform(-20, 20, 120).astype(np.float32)
    y = rng.uniform(-20, 20, 120).astype(np.float32)
    x = np.concatenate([x, x[:30] + np.float32(3e-5), x[:5], [70.0, -80.0]]).astype(np.float32)   # sub-EPS pairs, duplicates, outside +-55
    y = np.concatenate([y, y[:30], y[:5], [1.0, 2.0]]).astype(np.float32)
    n = len(x)
    p = ob.particles(x, y, rng.normal(0, 1, n), rng.normal(0, 1, n), rng.uniform(0.5, 2.0, n))
    run_both(ob, p, 0.6, 2)


def test_both_refuse_what_the_reference_panics_on(ob):
    # two bodies 1e-3 apart inside a huge box: the split chain exceeds depth 50?  no -- use the classic: > 50 levels needs
    # separations far below f32 resolution of the box, which the subdivision assert catches first; a zero mass trips add_mass
    p = ob.particles([0.0, 1.0, 2.0], [0.0, 1.0, 0.5], [0, 0, 0], [0, 0, 0], [1.0, 0.0, 1.0])
    assert ob.step_barnes_hut(p.copy(), 0.5, 0.01, 1) != 0
    a = {k: p[k].astype(np.float32).copy() for k in ("px", "py", "vx", "vy", "m")}
    with pytest.raises(bhpy.TreePanic):
        bhpy.step_barnes_hut(a["px"], a["py"], a["vx"], a["vy"], a["m"], 0.5, 0.01)


@pytest.mark.parametrize("shape", [(64, 48), (101, 37), (200, 200)])
def test_c_oracle_draw_equals_python_restatement(ob, shape):
    """nb_draw (nbody.rs:482-617): the C oracle against oracle/nbody_draw_py.py, pixel for pixel -- aspect handling,
    truncating casts, octant tails, saturation under heavy overlap, out-of-view bodies, the magenta cross."""
    from oracle import nbody_draw_py as drawpy

    w, h = shape
    p = ob.stable_orbits(2500, 0.5, 30.0, 21)
    p["px"][:300] = 0.125; p["py"][:300] = -0.25      # 300 bodies on one pixel: saturation
    p["px"][300:330] *= 5.0                            # outside the viewport
    p["vx"][330:340] = 0.0; p["vy"][330:340] = 0.0     # atan2(0, 0)
    want = drawpy.draw(p["px"], p["py"], p["vx"], p["vy"], w, h)
    got = np.asarray(ob.draw(p, w, h)).reshape(h, w)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,seed", [(1, 3), (2, 4), (500, 5), (3000, 0xDEADBEEFCAFEF00D)])
def test_c_oracle_presets_equal_python_restatement(ob, n, seed):
    """nb_random_disk / nb_stable_orbits (nbody.rs:39-104): f32 construction from the bit stream, Range sampling, draw
    order and arithmetic -- the C oracle against oracle/nbody_presets_py.py, bit for bit."""
    from oracle import nbody_presets_py as prepy

    for got, want in ((ob.random_disk(n, seed), prepy.random_disk(n, seed)),
                      (ob.stable_orbits(n, 0.5, 30.0, seed), prepy.stable_orbits(n, 0.5, 30.0, seed))):
        assert len(got) == len(want)
        for j, k in enumerate(("px", "py", "vx", "vy", "m")):
            assert np.array_equal(got[k].view(np.uint32), want[:, j].view(np.uint32)), k


def test_fp64_barnes_hut_arbiter_c_equals_python_restatement(ob):
    """orc_bh_forces_exact (the fp64 arbiter the device-tree tests lean on) against its own second restatement in Python
    (oracle/nbody_bh_py.py::bh_forces_exact): same tree, exact node sums, fp64 walk -- equal to fp64 rounding.  And the arbiter
    really is what the f32 restatement approximates: they agree to f32 accuracy on a small system (no drift to speak of)."""
    from oracle import nbody_bh_py as bp

    rng = np.random.default_rng(12)
    for n, theta in ((300, 0.5), (700, 0.85), (40, 0.3)):
        x = rng.normal(0, 10, n).astype(np.float32); y = rng.normal(0, 10, n).astype(np.float32)
        x[: n // 10] = x[n // 10: 2 * (n // 10)] + np.float32(3e-5)        # a few sub-EPS pairs: merged blobs are leaves
        y[: n // 10] = y[n // 10: 2 * (n // 10)]
        m = rng.uniform(0.5, 2.0, n).astype(np.float32)
        p = ob.particles(x, y, np.zeros(n), np.zeros(n), m)
        rc, ex, ey = ob.bh_forces_exact(p, theta, nthreads=3)
        assert rc == 0
        qx, qy = bp.bh_forces_exact(x, y, m, theta)
        scale = max(np.abs(qx).max(), np.abs(qy).max())
        assert np.abs(ex - qx).max() <= 1e-12 * scale and np.abs(ey - qy).max() <= 1e-12 * scale
        rc, fx, fy = ob.bh_forces(p, theta, nthreads=2)
        err = np.maximum(np.abs(fx - ex), np.abs(fy - ey)) / scale
        assert np.percentile(err, 99) <= 2e-6 and err.max() <= 2e-3      # a flipped opening decision at most
