#!/usr/bin/env python3
"""Hand-run leak probe (pytest does not collect it): engines and groups are created, used on every path (fast / bit-exact,
host / device tree, fp16 sources, device draw, checkpoints) and destroyed a few hundred times; device and host memory must
not creep.   python tests/leak_probe.py [rounds]"""
import ctypes
import os
import resource
import sys
import tempfile

import numpy as np

os.environ.setdefault("NBX_GROUP_EXCHANGE", "copy")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def device_free():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def one_round(rng, k):
    n = int(rng.choice([1000, 40000, 90000]))
    x = rng.normal(0, 10, n).astype(np.float32); y = rng.normal(0, 10, n).astype(np.float32)
    v = rng.normal(0, 1, (2, n)).astype(np.float32); m = rng.uniform(0.5, 2, n).astype(np.float32)
    for mode in ("fast", "strict"):
        e = rx.NBodyEngine(mode=mode)
        e.set_bh_tree("device" if k % 2 else "host")
        e.set_draw_device(bool(k % 3 == 0))
        if mode == "fast" and k % 4 == 0:
            e.set_source_precision(16)
        e.set_particles(x, y, v[0], v[1], m)
        e.step_barnes_hut(0.6, 0.01, 1); e.step_brute_force(0.01) if n <= 40000 else None
        e.draw(256, 256); e.forces(0.5)
        with tempfile.TemporaryDirectory() as d:
            e.save(os.path.join(d, "c")); e.load(os.path.join(d, "c"))
        e.get_particles()
        e.close()
    g = rx.NBodyGroup([0, 0, 0], mode="fast")
    g.set_particles(x, y, v[0], v[1], m)
    g.step_barnes_hut(0.6, 0.01, 1); g.synchronize(); g.get_particles(); g.close()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(0)
    for k in range(10):
        one_round(rng, k)                      # warm-up: pools, caches, lazily loaded code objects
    d0, r0 = device_free(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for k in range(rounds):
        one_round(rng, k)
    d1, r1 = device_free(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("leak probe: %d rounds, device free %+.1f MB, host max RSS %+.1f MB" % (rounds, (d1 - d0) / 1e6, (r1 - r0) / 1e3))


if __name__ == "__main__":
    main()
