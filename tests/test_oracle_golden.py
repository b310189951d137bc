"""CPU: the oracle (oracle/nbody_oracle.c) against the committed golden vectors and against the
independent numpy restatement. These pins stand in for reference tests, which do not exist
(SURVEY.md section 4)."""
import numpy as np
import pytest

from conftest import assert_bit_equal, golden, particles_from

BRUTE = ["brute_n2", "brute_n5_orbits", "brute_n64_disk", "brute_n1024_orbits", "brute_n1000_disk"]
BH = [f"bh_{c}_t{t}" for c in ("n64_disk", "n1024_orbits", "n1000_disk") for t in ("0p5", "0p85")]


@pytest.mark.parametrize("name", BRUTE)
def test_oracle_brute_matches_golden(ob, name):
    g = golden(name)
    p = particles_from(ob, g)
    done = 0
    for s in (1, 10):
        while done < s:
            assert ob.step_brute_force(p, float(g["dt"])) == 0
            done += 1
        for k in ("px", "py", "vx", "vy"):
            assert_bit_equal(p[k], g[f"s{s}_{k}"], f"{name} step {s} {k}")


@pytest.mark.parametrize("name", BH)
def test_oracle_bh_matches_golden(ob, name):
    g = golden(name)
    p = particles_from(ob, g)
    rc, fx, fy = ob.bh_forces(p, float(g["theta"]))
    assert rc == 0
    assert_bit_equal(fx, g["f0_x"], name + " fx")
    assert_bit_equal(fy, g["f0_y"], name + " fy")
    done = 0
    for s in (1, 10):
        while done < s:
            assert ob.step_barnes_hut(p, float(g["theta"]), float(g["dt"]), 1) == 0
            done += 1
        for k in ("px", "py", "vx", "vy"):
            assert_bit_equal(p[k], g[f"s{s}_{k}"], f"{name} step {s} {k}")


@pytest.mark.parametrize("n,seed", [(3, 11), (100, 12), (513, 13)])
def test_oracle_equals_numpy_restatement(ob, n, seed):
    from oracle import nbody_numpy as onp

    p0 = ob.random_disk(n, seed)
    p = p0.copy()
    px, py, vx, vy = (p0[k].copy() for k in ("px", "py", "vx", "vy"))
    for _ in range(3):
        ob.step_brute_force(p, 0.01)
        px, py, vx, vy = onp.step_brute_force(px, py, vx, vy, p0["m"], 0.01)
    for k, a in (("px", px), ("py", py), ("vx", vx), ("vy", vy)):
        assert_bit_equal(p[k], a, k)


def test_oracle_threaded_brute_is_bit_identical(ob):
    p0 = ob.random_disk(777, 21)
    a, b = p0.copy(), p0.copy()
    ob.step_brute_force(a, 0.01)
    assert ob.step_brute_force(b, 0.01, nthreads=5) == 0
    assert np.array_equal(a, b)


def test_oracle_bh_threads_do_not_change_results(ob):
    p0 = ob.stable_orbits(500, 0.5, 30.0, 4)
    a, b = p0.copy(), p0.copy()
    ob.step_barnes_hut(a, 0.85, 0.01, 1)
    ob.step_barnes_hut(b, 0.85, 0.01, 7)
    assert np.array_equal(a, b)


def test_oracle_draw_matches_golden(ob):
    g = golden("draw_n1024_orbits")
    p = particles_from(ob, g)
    assert np.array_equal(ob.draw(p, 64, 48), g["fb_64x48"])
    assert np.array_equal(ob.draw(p, 512, 512), g["fb_512x512"])


def test_oracle_presets_match_golden(ob):
    g = golden("presets")
    d = np.array(ob.random_disk(257, 7)).view(np.float32).reshape(-1, 5)
    o = np.array(ob.stable_orbits(100, 0.5, 30.0, 9)).view(np.float32).reshape(-1, 5)
    assert np.array_equal(d.view(np.uint32), g["disk_seed7_n257"].view(np.uint32))
    assert np.array_equal(o.view(np.uint32), g["orbits_seed9_n100"].view(np.uint32))
