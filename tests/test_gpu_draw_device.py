"""GPU: nb_draw on the device against the oracle's nb_draw.
Body pixels and every tail whose velocity direction is not within an ulp of an octant boundary are
bit-identical; the saturating per-channel add is order independent, so atomics give exact counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def eng(rx, p):
    e = rx.NBodyEngine()
    e.set_draw_device(True)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    return e


@pytest.mark.parametrize("shape", [(512, 512), (64, 48), (101, 37)])
def test_device_draw_exact_on_octant_safe_velocities(rx, ob, shape):
    w, h = shape
    rng = np.random.default_rng(1)
    n = 20000
    # velocities on exact octant directions and v = 0 (atan2 exact: 0, pi/4.., pi): no boundary ambiguity
    dirs = np.array([(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (0, 0)], np.float32)
    v = dirs[rng.integers(0, 9, n)] * rng.uniform(0.5, 4.0, (n, 1)).astype(np.float32)
    p = ob.particles(rng.normal(0, 20, n), rng.normal(0, 20, n), v[:, 0], v[:, 1], np.ones(n))
    p["px"][:100] = 70.0          # outside the viewport: dropped (nbody.rs:559)
    p["px"][100:200] = 0.25       # heavy overlap on one pixel: saturation (nbody.rs:611-614)
    p["py"][100:200] = 0.25
    want = ob.draw(p, w, h)
    got = eng(rx, p).draw(w, h)
    if not np.array_equal(got, want):
        # exact-direction tails can still differ if device atan2f is off by an ulp at pi/4 multiples:
        assert (got != want).sum() <= 0, (got != want).sum()


def test_device_draw_matches_host_draw_up_to_tail_boundaries(rx, ob):
    p = ob.stable_orbits(100000, 0.5, 30.0, 3)
    want = ob.draw(p, 512, 512)
    e = eng(rx, p)
    got = e.draw(512, 512)
    diff = (got != want).sum()
    assert diff <= 8, diff                       # a boundary tail moves 1 count between 2 pixels
    assert (got == 0x00FF00FF).sum() == 5        # centre cross
    # after stepping, the device draw sees the live device state without a state download
    e.step_barnes_hut(0.85, 0.01, 1)
    q = p.copy(); ob.step_barnes_hut(q, 0.85, 0.01, 1)
    got2 = e.draw(512, 512)
    e.set_draw_device(False)
    host2 = e.draw(512, 512)
    assert (got2 != host2).sum() <= 8
    assert not np.array_equal(got2, got)


def test_device_draw_empty_and_tiny(rx):
    e = rx.NBodyEngine()
    e.set_draw_device(True)
    e.set_particles([], [], [], [], [])
    fb = e.draw(16, 16)
    assert (fb == 0x00FF00FF).sum() == 5 and (fb != 0).sum() == 5
    e.set_particles([0.0], [0.0], [1.0], [0.0], [1.0])
    fb = e.draw(2, 2)                             # w,h < 3: no cross (guarded)
    assert (fb == 0x00FF00FF).sum() == 0
