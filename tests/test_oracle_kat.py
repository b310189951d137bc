"""CPU: known-answer tests implied by the SOURCE semantics of rs-src/nbody.rs.
They pin the oracle where no reference fixture exists."""
import numpy as np

from conftest import assert_bit_equal

EPS = np.float32(0.0001)


def test_force_expression_order(ob):
    # nbody.rs:174-183 evaluated by hand in f32
    px1, py1, m1, px2, py2, m2 = map(np.float32, (0.25, -1.5, 3.0, 2.0, 0.75, 0.5))
    dx = px2 - px1
    dy = py2 - py1
    d2 = dx * dx + dy * dy
    f = m1 * m2 / (d2 + EPS)
    fx, fy = ob.force(px1, py1, m1, px2, py2, m2)
    assert fx == f * dx and fy == f * dy


def test_force_is_antisymmetric_bitwise(ob):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.uniform(-30, 30, 4).astype(np.float32)
        m = rng.uniform(0.1, 1000, 2).astype(np.float32)
        f12 = ob.force(a[0], a[1], m[0], a[2], a[3], m[1])
        f21 = ob.force(a[2], a[3], m[1], a[0], a[1], m[0])
        assert f12[0] == -f21[0] and f12[1] == -f21[1]


def test_force_magnitude_is_one_over_r_not_inverse_square(ob):
    # un-normalised direction (nbody.rs:171-183): |F| = m1 m2 r / (r^2 + eps) ~ 1/r
    f1, _ = ob.force(0, 0, 1, 1.0, 0, 1)
    f2, _ = ob.force(0, 0, 1, 2.0, 0, 1)
    assert abs(f1 / f2 - 2.0) < 1e-3


def test_coincident_distinct_bodies_exert_zero_force(ob):
    p = ob.particles([1.0, 1.0, 5.0], [2.0, 2.0, 5.0], [0, 0, 0], [0, 0, 0], [3.0, 4.0, 1.0])
    fx, fy = ob.brute_forces(p)
    # bodies 0 and 1 sit on the same point: their mutual term is f*0 = 0, only body 2 pulls
    e0 = ob.force(1.0, 2.0, 3.0, 5.0, 5.0, 1.0)
    assert fx[0] == e0[0] and fy[0] == e0[1]


def test_two_equal_bodies_equal_and_opposite(ob):
    p = ob.particles([-1.0, 1.0], [0.0, 0.0], [0, 0], [0, 0], [2.0, 2.0])
    ob.step_brute_force(p, 0.01)
    assert p["vx"][0] == -p["vx"][1] and p["vy"][0] == 0 and p["vy"][1] == 0
    assert p["px"][0] == -p["px"][1]


def test_momentum_change_sums_to_rounding(ob):
    p = ob.random_disk(300, 5)
    v0 = np.array([p["vx"], p["vy"]], dtype=np.float64)
    m = p["m"].astype(np.float64)
    ob.step_brute_force(p, 0.01)
    dv = np.array([p["vx"], p["vy"]], dtype=np.float64) - v0
    mom = (dv * m).sum(axis=1)
    scale = np.abs(dv * m).sum(axis=1)
    assert np.all(np.abs(mom) < 1e-5 * scale)


def test_integrator_is_kick_then_drift_with_new_velocity(ob):
    # nbody.rs:153-160: v += (dt*F)/m ; p += dt * v_new
    p = ob.particles([0.0, 3.0], [0.0, 4.0], [1.0, 0.0], [2.0, 0.0], [2.0, 5.0])
    fx, fy = ob.brute_forces(p)
    dt = np.float32(0.01)
    vx = p["vx"] + (dt * fx) / p["m"]
    vy = p["vy"] + (dt * fy) / p["m"]
    px = p["px"] + dt * vx
    py = p["py"] + dt * vy
    ob.step_brute_force(p, float(dt))
    assert_bit_equal(p["vx"], vx); assert_bit_equal(p["vy"], vy)
    assert_bit_equal(p["px"], px); assert_bit_equal(p["py"], py)


def test_circular_orbit_speed_independent_of_radius(ob):
    # 1/r law: orbital speed sqrt(G M) at ANY radius (nbody.rs:88,100-101)
    for r in (2.0, 10.0, 30.0):
        p = ob.particles([0.0, r], [0.0, 0.0], [0.0, 0.0], [0.0, np.sqrt(1000.0)], [1000.0, 1e-6])
        for _ in range(200):
            ob.step_brute_force(p, 0.001)
        rr = np.hypot(p["px"][1] - p["px"][0], p["py"][1] - p["py"][0])
        assert abs(rr - r) / r < 2e-2


def test_theta_zero_is_brute_force_bit_for_bit(ob):
    p0 = ob.random_disk(200, 9)
    a, b = p0.copy(), p0.copy()
    ob.step_brute_force(a, 0.01)
    assert ob.step_barnes_hut(b, 0.0, 0.01, 3) == 0      # nbody.rs:197-200
    assert np.array_equal(a, b)


def test_bh_tiny_theta_approaches_brute_force(ob):
    # keep every body inside the +-55 kill box and farther apart than EPS: only summation order differs
    p0 = ob.random_disk(256, 10)
    fx, fy = ob.brute_forces(p0)
    rc, bx, by = ob.bh_forces(p0, 1e-6)
    assert rc == 0
    scale = np.abs(fx).max()
    assert np.abs(bx - fx).max() < 2e-5 * scale and np.abs(by - fy).max() < 2e-5 * np.abs(fy).max()


def test_bh_velocity_kill_outside_55(ob):
    # nbody.rs:466-471 -- Barnes-Hut only
    p0 = ob.particles([0.0, 56.0, -10.0], [0.0, 0.0, 54.9], [0.0, 1.0, 1.0], [0.0, 1.0, 1.0], [1000.0, 1.0, 1.0])
    a, b = p0.copy(), p0.copy()
    ob.step_barnes_hut(a, 0.5, 0.01, 1)
    assert a["vx"][1] == 0 and a["vy"][1] == 0
    assert a["vx"][2] != 0
    ob.step_brute_force(b, 0.01)
    assert b["vx"][1] != 0      # brute force has no kill box


def test_bh_merges_particles_closer_than_eps(ob):
    # nbody.rs:249-260: |dx|,|dy| < EPS -> merged into one exterior node
    p = ob.particles([1.0, 1.00005, -3.0], [1.0, 1.00005, 2.0], [0] * 3, [0] * 3, [1.0, 2.0, 1.0])
    rc, st = ob.bh_tree_stats(p)
    assert rc == 0 and st["leaves"] == 2
    assert abs(st["m"] - 4.0) < 1e-6


def test_bh_depth_panic(ob):
    # two bodies 2e-4 apart in a unit-wide box separate only after ~13 splits; farther than EPS so no merge.
    # With a huge box the required depth exceeds 50 -> the reference panics (nbody.rs:230-232)
    p = ob.particles([0.0, 1e30, 1.0, 1.0003], [0.0, 1e30, 1.0, 1.0], [0] * 4, [0] * 4, [1.0] * 4)
    rc, _ = ob.bh_tree_stats(p)
    assert rc == ob.ORC_PANIC_DEPTH


def test_bh_nonpositive_mass_panics(ob):
    p = ob.particles([0.0, 1.0], [0.0, 1.0], [0, 0], [0, 0], [1.0, 0.0])
    rc, _ = ob.bh_tree_stats(p)
    assert rc == ob.ORC_PANIC_MASS       # nbody.rs:304


def test_bh_nthreads_zero_updates_nobody_and_does_not_panic(ob):
    """nbody.rs:424-428: the division by nthreads sits inside the (0..nthreads).map closure; with nthreads <= 0 the iterator
    is empty, no worker runs, the state is untouched.  The tree is still built before that (:380-417): its asserts can fire."""
    p = ob.random_disk(10, 1)
    q = p.copy()
    for t in (0, -3):
        assert ob.step_barnes_hut(q, 0.5, 0.01, t) == 0
        assert np.array_equal(q.view(np.uint8), p.view(np.uint8))
    bad = ob.particles([0.0, 1.0], [0.0, 1.0], [0, 0], [0, 0], [1.0, 0.0])
    assert ob.step_barnes_hut(bad, 0.5, 0.01, 0) == ob.ORC_PANIC_MASS      # the build's assert (:304) still fires
    assert ob.step_barnes_hut(q, 0.0, 0.01, 0) == 0 and not np.array_equal(q["px"], p["px"])   # theta == 0 never looks at nthreads


def test_stable_orbits_preset_shape(ob):
    p = ob.stable_orbits(1000, 0.5, 30.0, 2)
    assert len(p) == 1000
    assert tuple(p[0]) == (0.0, 0.0, 0.0, 0.0, 1000.0)          # nbody.rs:93
    r = np.hypot(p["px"][1:], p["py"][1:])
    assert r.min() >= 0.5 - 1e-4 and r.max() <= 30.0 + 1e-4
    sp = np.hypot(p["vx"][1:], p["vy"][1:])
    assert np.allclose(sp, np.sqrt(1000.0), rtol=1e-5)          # :88
    # tangential: v . p = 0
    assert np.abs(p["vx"][1:] * p["px"][1:] + p["vy"][1:] * p["py"][1:]).max() < 1e-2
    assert np.all(p["m"][1:] == 1.0)
    assert len(ob.stable_orbits(0, 1, 2, 3)) == 1               # the sun is always pushed, 0..n-1 is empty
    assert len(ob.stable_orbits(1, 1, 2, 3)) == 1


def test_random_disk_preset_shape(ob):
    p = ob.random_disk(5000, 3)
    r = np.hypot(p["px"], p["py"])
    assert r.max() <= 23.0 + 1e-4                               # nbody.rs:55-56
    assert 14.5 < np.median(r) < 17.5                           # sqrt-law: median = 23/sqrt(2)
    assert p["vx"].min() >= -3.5 and p["vx"].max() < 3.5        # :47
    assert p["m"].min() >= 0.1 and p["m"].max() < 1.5           # :48
    assert len(ob.random_disk(0, 1)) == 0 and len(ob.random_disk(-5, 1)) == 0


def test_next_f32_is_top_24_bits(ob):
    import ctypes as C

    s = C.c_uint64(123)
    vals = [ob.lib().orc_next_f32(C.byref(s)) for _ in range(1000)]
    assert all(0.0 <= v < 1.0 for v in vals)
    assert all(float(v) * 16777216.0 == int(float(v) * 16777216.0) for v in vals)


def test_draw_colours_cross_and_saturation(ob):
    assert ob.lib().orc_rgb_to_abgr32(255, 215, 130, 0.3) == 0x0027404C     # nbody.rs:520
    assert ob.lib().orc_rgb_to_abgr32(255, 215, 130, 0.25) == 0x0020353F    # :521
    assert ob.lib().orc_add_abgr32(0x00F0F0F0, 0x00202020) == 0x00FFFFFF    # :611-614 saturate per channel
    # 5 bodies on one pixel moving east: body pixel saturates R after 4 overlaps (4*76 > 255)
    p = ob.particles([10.0] * 5, [10.0] * 5, [1.0] * 5, [0.0] * 5, [1.0] * 5)
    fb = ob.draw(p, 100, 100)
    x, y = int((10 + 50) * 1.0), int((10 + 50) * 1.0)
    assert fb[y, x] & 0xFF == 255
    assert fb[y, x] == ((min(255, 5 * 39) << 16) | (min(255, 5 * 64) << 8) | 255)
    assert fb[y, x - 1] == ((5 * 32) << 16 | (min(255, 5 * 53)) << 8 | min(255, 5 * 63))  # tail one pixel west (E octant)
    # magenta centre cross, 5 pixels (nbody.rs:571-577)
    cross = [(50, 50), (50, 51), (51, 50), (50, 49), (49, 50)]
    assert all(fb[r, c] == 0x00FF00FF for r, c in cross)
    assert (fb == 0x00FF00FF).sum() == 5


def test_draw_truncation_and_bounds(ob):
    # x in (-1,0) truncates toward zero into column 0 (`as i32`), out-of-viewport bodies are dropped
    p = ob.particles([-50.5, 60.0, -49.5], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0] * 3)
    fb = ob.draw(p, 100, 100)
    assert fb[50, 0] == 2 * 0x0027404C      # -50.5 -> x=-0.5 -> column 0 ; -49.5 -> 0.5 -> column 0
    assert (fb != 0).sum() == 5 + 1 + 1     # cross + body pixel + tail pixel (N octant: one row up, y-1)


def test_bh_depth_panic_counts_two_per_level_of_a_split_chain(ob):
    """nbody.rs:230-232 tests a COUNTER, not the level of a node: the re-insert of a split (nbody.rs:278-281) starts one above the
    node it descends from, so the counter grows by two per level while a leaf is split down.  A body that arrives at a leaf at
    level d0 and ends at level d1 drives it to d0 + 2 (d1 - d0) <= 2 d1: a tree whose leaves are all at level <= 25 cannot panic,
    deeper ones can although no node is anywhere near level 50.  (The device build has no such counter: in the bit-exact mode it
    hands every tree with a leaf below level 25 to the host build -- bh_build.hip kWhyDepthPanic.)"""
    rng = np.random.default_rng(12)
    panics_seen = 0
    for case in range(60):
        half = float(rng.choice([50.0, 2e3, 2e4, 1e6]))
        gap = float(rng.choice([1.5e-4, 4e-4, 3e-3, 0.05]))
        n0 = 200
        x = np.concatenate([[1.0, 1.0 + gap], rng.uniform(-half, half, n0)]).astype(np.float32)
        y = np.concatenate([[1.0, 1.0], rng.uniform(-half, half, n0)]).astype(np.float32)
        n = len(x)
        order = rng.permutation(n)
        p = ob.particles(x[order], y[order], np.zeros(n), np.zeros(n), np.ones(n))
        rc, st = ob.bh_tree_stats(p)
        # level at which the two close bodies part: the descent of nbody.rs:289-300 / :324-331 in f32
        f = np.float32
        x1, y1, x2, y2 = f(x.min()), f(y.min()), f(x.max()), f(y.max())
        level = 0
        while level < 60:
            cx, cy = f((x1 + x2) * f(0.5)), f((y1 + y2) * f(0.5))
            qa = (2 if y[0] < cy else 0) + (0 if x[0] < cx else 1)
            qb = (2 if y[1] < cy else 0) + (0 if x[1] < cx else 1)
            if qa != qb:
                break
            if y[0] < cy: y2 = cy
            else: y1 = cy
            if x[0] < cx: x2 = cx
            else: x1 = cx
            level += 1
        leaf_level = level + 1
        if rc != 0:
            panics_seen += 1
            assert leaf_level > 25, (half, gap, leaf_level)          # a panic needs a leaf below level 25 ...
            assert leaf_level < 50                                    # ... and none anywhere near level 50
        else:
            assert st["depth"] >= leaf_level - 1
    assert panics_seen >= 3
