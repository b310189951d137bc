"""GPU: long runs are chaotic, so they are checked through invariants of the integrator + force law rather
than against a trajectory: total momentum (exact pairwise antisymmetry of nbody.rs:174-183)
and the energy-like quantity of the 1/r law,  E = sum 1/2 m v^2 + sum_{i<j} 1/2 m_i m_j ln(r_ij^2 + eps),
which the kick-drift (symplectic Euler) scheme of nbody.rs:153-160 keeps bounded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def energy_and_momentum(st):
    m = st["m"].astype(np.float64)
    x = st["px"].astype(np.float64); y = st["py"].astype(np.float64)
    vx = st["vx"].astype(np.float64); vy = st["vy"].astype(np.float64)
    kin = 0.5 * (m * (vx * vx + vy * vy)).sum()
    dx = x[:, None] - x[None, :]; dy = y[:, None] - y[None, :]
    r2 = dx * dx + dy * dy + 1e-4
    pot = 0.25 * (m[:, None] * m[None, :] * np.log(r2)).sum() - 0.25 * (m * m * np.log(1e-4)).sum()   # i<j pairs
    return kin, pot, np.array([(m * vx).sum(), (m * vy).sum()])


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_momentum_and_energy_over_many_steps(rx, ob, mode):
    p = ob.random_disk(1024, 71)
    e = rx.NBodyEngine(mode=mode)
    e.set_particles(p["px"], p["py"], p["vx"], p["vy"], p["m"])
    k0, u0, mom0 = energy_and_momentum(e.get_particles())
    scale = (p["m"].astype(np.float64) * np.hypot(p["vx"], p["vy"])).sum()
    es = []
    for block in range(8):
        for _ in range(250):
            e.step_brute_force(0.0005)
        k, u, mom = energy_and_momentum(e.get_particles())
        es.append(k + u)
        assert np.all(np.abs(mom - mom0) <= 2e-5 * scale), (block, mom - mom0)      # f32 rounding only
    es = np.array(es)
    # bounded, not drifting: every sample within 1 % of the initial energy (|K| ~ |U| scale)
    assert np.abs(es - (k0 + u0)).max() <= 1e-2 * (abs(k0) + abs(u0)), (es, k0 + u0)


def test_barnes_hut_long_run_stays_bounded(rx, ob):
    """2000 Barnes-Hut steps of the default scene (10 000 stable orbits, theta 0.85, dt 0.01): bodies stay inside
    the kill box (nbody.rs:466-471 pulls escapers back) and nothing turns NaN. (Barnes-Hut forces are not pairwise
    antisymmetric, so the 1000-mass sun random-walks a few units under 10 000 unit-mass planets: measured 4.5.)"""
    e = rx.NBodyEngine()
    e.seed(3)
    e.stable_orbits(10000, 0.5, 30.0)
    for _ in range(2000):
        e.step_barnes_hut(0.85, 0.01, 1)
    st = e.get_particles()
    assert np.isfinite(st["px"]).all() and np.isfinite(st["vx"]).all()
    assert np.hypot(st["px"][0], st["py"][0]) < 15.0
    r = np.hypot(st["px"][1:], st["py"][1:])
    assert np.median(r) < 35 and (r > 80).mean() < 0.01
