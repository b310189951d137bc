import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import rust_exp_amd

    if rust_exp_amd.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding

    binding.lib()
    return binding


@pytest.fixture(scope="session")
def rx():
    """The product package (HIP library binding)."""
    import rust_exp_amd

    rust_exp_amd.lib()
    return rust_exp_amd


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, what
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a[bad[:5]]} vs {b[bad[:5]]}"


def particles_from(ob, d, prefix="in_"):
    return ob.particles(d[prefix + "px"], d[prefix + "py"], d[prefix + "vx"], d[prefix + "vy"], d["in_m"])
