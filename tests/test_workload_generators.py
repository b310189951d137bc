"""The benchmark workload generators behind the C ABI (nbx_plummer_sphere / nbx_two_galaxies; SURVEY.md 8(d): "so C++
and numpy agree bit-for-bit") against the committed golden values (tests/golden/workload_*: produced by the numpy
restatement alone) and against that restatement run now.  Host-side entry points: no GPU needed."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal

KEYS = ("px", "py", "pz", "vx", "vy", "vz", "m")


def _digest(st):
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(st[k], dtype="<f4").tobytes())
    return h.hexdigest()


def _lib_plummer(rx, n, seed, dim):
    e = rx.NBodyEngine()
    e.plummer_sphere(n, seed, dim)
    assert e.num_particles() == n
    return e.get_particles()


def _lib_galaxies(rx, n, seed):
    e = rx.NBodyEngine()
    e.two_galaxies(n, seed)
    assert e.num_particles() == n
    return e.get_particles()


@pytest.mark.parametrize("name,make", [
    ("workload_plummer_n1000_dim3", lambda rx: _lib_plummer(rx, 1000, 0x5EED0001, 3)),
    ("workload_plummer_n1000_dim2", lambda rx: _lib_plummer(rx, 1000, 0x5EED0001, 2)),
    ("workload_plummer_n257_seed7", lambda rx: _lib_plummer(rx, 257, 7, 3)),
    ("workload_two_galaxies_n1000", lambda rx: _lib_galaxies(rx, 1000, 0x5EED0002)),
    ("workload_two_galaxies_n7_seed3", lambda rx: _lib_galaxies(rx, 7, 3)),
])
def test_library_generators_equal_the_committed_golden_values(rx, name, make):
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = make(rx)
    for k in KEYS:
        assert_bit_equal(got[k], want[k], f"{name} {k}")


def test_library_generators_equal_the_golden_digests_at_the_benchmark_sizes(rx):
    """BASELINE configs #2 / #3 / #4 (Plummer 65 536, 262 144, 1 048 576) and #5 (two galaxies, 524 288)."""
    dig = json.load(open(os.path.join(GOLDEN, "workload_digests.json")))
    for n, dim in ((65536, 3), (262144, 3), (262144, 2), (1048576, 2)):
        assert _digest(_lib_plummer(rx, n, 0x5EED0001, dim)) == dig[f"plummer_n{n}_dim{dim}_seed0x5EED0001"], (n, dim)
    assert _digest(_lib_galaxies(rx, 524288, 0x5EED0002)) == dig["two_galaxies_n524288_seed0x5EED0002"]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 256, 4097])
def test_library_generators_equal_the_numpy_restatement_run_now(rx, n):
    a, b = _lib_plummer(rx, n, 12345, 3), rx.plummer_sphere(n, seed=12345, dim=3)
    for k in KEYS:
        assert_bit_equal(a[k], b[k], f"plummer n={n} {k}")
    a, b = _lib_galaxies(rx, n, 99), rx.two_galaxies(n, seed=99)
    for k in KEYS:
        assert_bit_equal(a[k], b[k], f"two_galaxies n={n} {k}")


def test_workload_shapes(rx):
    st = _lib_plummer(rx, 4096, 0x5EED0001, 3)
    r = np.sqrt(st["px"].astype(np.float64) ** 2 + st["py"] ** 2 + st["pz"] ** 2)
    assert r.max() <= 45.0 + 1e-3 and 3.0 < np.median(r) < 9.0          # inside the +-55 kill box of nbody.rs:466-471
    assert not st["vx"].any() and len(set(st["m"].tolist())) == 1 and abs(st["m"].sum() - 1000.0) < 1e-2
    g = _lib_galaxies(rx, 4096, 0x5EED0002)
    assert g["m"][0] == 1000.0 and g["m"][2048] == 1000.0 and (np.delete(g["m"], [0, 2048]) == 1.0).all()
    assert g["px"][0] == -15.0 and g["px"][2048] == 15.0 and not g["pz"].any()
    # planets on circular orbits around their core (speed sqrt(1000) at any radius: the reference's 1/r law, nbody.rs:88)
    rel = np.hypot(g["vx"][1:2048] - 3.0, g["vy"][1:2048] + 1.0)
    assert np.allclose(rel, np.sqrt(1000.0), rtol=1e-5)


def test_workload_generators_reject_bad_arguments(rx):
    e = rx.NBodyEngine()
    with pytest.raises(rx.NBodyError):
        e.plummer_sphere(-1)
    with pytest.raises(rx.NBodyError):
        e.plummer_sphere(16, dim=4)
    with pytest.raises(rx.NBodyError):
        e.two_galaxies(-5)
