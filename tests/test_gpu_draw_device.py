"""GPU: nb_draw on the device against the oracle's nb_draw (nbody.rs:482-617): PIXEL-IDENTICAL,
tails included.  The device decides a tail's octant itself unless the reference's f32 expression
((8*atan2f(vy,vx)/(2*pi)+8) as i32 % 8, nbody.rs:541-542) sits within 1e-5 of one of its steps; those few particles are
handed to the host, which evaluates the expression with its own atan2f (draw.hip).  The saturating per-channel add is order
independent, so atomics give exact counts."""
import numpy as np
import pytest

from rust_exp_amd.engine import NBX_STAT_DRAW_AMBIGUOUS

pytestmark = pytest.mark.gpu


def eng(rx, p, device=True):
    e = rx.NBodyEngine()
    e.set_draw_device(device)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    return e


@pytest.mark.parametrize("shape", [(512, 512), (64, 48), (101, 37)])
def test_device_draw_exact_on_octant_directions(rx, ob, shape):
    """Velocities exactly on the 8 octant directions and v = 0: axis-aligned ones are decided on the device, exact diagonals
    sit ON a step of the expression and go to the host (what atan2f returns for 3*pi/4 decides the pixel)."""
    w, h = shape
    rng = np.random.default_rng(1)
    n = 20000
    dirs = np.array([(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (0, 0)], np.float32)
    v = dirs[rng.integers(0, 9, n)] * rng.uniform(0.5, 4.0, (n, 1)).astype(np.float32)
    p = ob.particles(rng.normal(0, 20, n), rng.normal(0, 20, n), v[:, 0], v[:, 1], np.ones(n))
    p["px"][:100] = 70.0          # outside the viewport: dropped (nbody.rs:559)
    p["px"][100:200] = 0.25       # heavy overlap on one pixel: saturation (nbody.rs:611-614)
    p["py"][100:200] = 0.25
    p["vx"][200:210] = -0.0       # signed zeros: atan2f(+-0, -0) = +-pi
    p["vy"][200:205] = 0.0
    p["vy"][205:210] = -0.0
    want = ob.draw(p, w, h)
    e = eng(rx, p)
    got = e.draw(w, h)
    assert np.array_equal(got, want), int((got != want).sum())
    n_diag = int(((np.abs(p["vx"]) == np.abs(p["vy"])) & (p["vx"] != 0)).sum())
    assert e.get_stat(NBX_STAT_DRAW_AMBIGUOUS) == n_diag > 1000


def test_device_draw_adversarial_directions_near_every_octant_step(rx, ob):
    """Directions a few ulps either side of every multiple of 45 degrees, at many magnitudes, plus huge / tiny / infinite /
    NaN components: the band hands them to the host, everything else is decided on the device; every pixel equals the oracle."""
    rng = np.random.default_rng(5)
    vx, vy = [], []
    for k in range(8):
        a0 = k * np.pi / 4
        for off in np.concatenate([np.linspace(-3e-5, 3e-5, 241), rng.normal(0, 2e-7, 200)]):
            r = np.float64(10.0 ** rng.uniform(-3, 3))
            vx.append(r * np.cos(a0 + off)); vy.append(r * np.sin(a0 + off))
    # one-ulp neighbours of the exact diagonals and axes
    for sx, sy in ((1, 1), (-1, 1), (-1, -1), (1, -1)):
        for m in (0.37, 1.0, 31.62, 1e-20, 1e20):
            b = np.float32(m)
            for dx in (-2, -1, 0, 1, 2):
                for dy in (-2, -1, 0, 1, 2):
                    x = np.float32(b).view(np.uint32) + np.uint32(dx) if dx >= 0 else np.float32(b).view(np.uint32) - np.uint32(-dx)
                    y = np.float32(b).view(np.uint32) + np.uint32(dy) if dy >= 0 else np.float32(b).view(np.uint32) - np.uint32(-dy)
                    vx.append(sx * float(np.uint32(x).view(np.float32))); vy.append(sy * float(np.uint32(y).view(np.float32)))
    for m in (1e-45, 1e-38, 1.0, 3e38):      # subnormal tilt off an axis: 8a/2pi rounds back onto the integer
        for sx in (1, -1):
            vx += [sx * 1.0, sx * 1.0, m, -m]; vy += [m, -m, sx * 1.0, sx * 1.0]
    vx += [np.inf, -np.inf, np.inf, 1.0, np.nan, 1.0, 0.0, np.inf]
    vy += [np.inf, np.inf, -1.0, -np.inf, 1.0, np.nan, np.nan, 0.0]
    vx = np.asarray(vx, np.float32); vy = np.asarray(vy, np.float32)
    n = len(vx)
    p = ob.particles(rng.uniform(-45, 45, n), rng.uniform(-45, 45, n), vx, vy, np.ones(n))
    want = ob.draw(p, 256, 256)
    e = eng(rx, p)
    got = e.draw(256, 256)
    assert np.array_equal(got, want), int((got != want).sum())
    amb = e.get_stat(NBX_STAT_DRAW_AMBIGUOUS)
    assert 500 < amb < n                       # the band caught the near-step directions, and only part of the set is in it


def test_device_draw_equals_oracle_on_100000_random_bodies_and_is_the_default_for_large_systems(rx, ob):
    """np.array_equal on the 100 000-body case; then the default engine (no option set) takes the device
    path for >= 4096 resident bodies and the host path below -- same pixels either way."""
    p = ob.stable_orbits(100000, 0.5, 30.0, 3)
    want = ob.draw(p, 512, 512)
    e = eng(rx, p)
    got = e.draw(512, 512)
    assert np.array_equal(got, want)
    assert (got == 0x00FF00FF).sum() == 5        # centre cross
    q = ob.random_disk(100000, 4)
    assert np.array_equal(eng(rx, q).draw(512, 512), ob.draw(q, 512, 512))
    # after stepping, the device draw sees the live device state without a state download
    e.step_barnes_hut(0.85, 0.01, 1)
    got2 = e.draw(512, 512)
    e.set_draw_device(False)
    host2 = e.draw(512, 512)
    assert np.array_equal(got2, host2) and not np.array_equal(got2, got)
    st = e.get_particles()
    assert np.array_equal(got2, ob.draw(ob.particles(st["px"], st["py"], st["vx"], st["vy"], st["m"]), 512, 512))
    # default engine: by size
    d = rx.NBodyEngine()
    d.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    d.step_brute_force(0.0)                      # state resident on the GPU (a zero-length step changes nothing)
    assert np.array_equal(d.draw(512, 512), want) and d.get_stat(NBX_STAT_DRAW_AMBIGUOUS) >= 0      # ran on the device
    small = rx.NBodyEngine()
    small.set_particles(p["px"][:4000], p["py"][:4000], p["vx"][:4000], p["vy"][:4000], p["m"][:4000])
    small.step_brute_force(0.0)
    assert np.array_equal(small.draw(512, 512), ob.draw(p[:4000], 512, 512)) and small.get_stat(NBX_STAT_DRAW_AMBIGUOUS) == -1
    from rust_exp_amd.engine import NBX_OPT_DRAW_DEVICE
    assert d.get_option(NBX_OPT_DRAW_DEVICE) == -1


def test_device_draw_empty_and_tiny(rx):
    e = rx.NBodyEngine()
    e.set_draw_device(True)
    e.set_particles([], [], [], [], [])
    fb = e.draw(16, 16)
    assert (fb == 0x00FF00FF).sum() == 5 and (fb != 0).sum() == 5
    e.set_particles([0.0], [0.0], [1.0], [0.0], [1.0])
    fb = e.draw(2, 2)                             # w,h < 3: no cross (guarded)
    assert (fb == 0x00FF00FF).sum() == 0
