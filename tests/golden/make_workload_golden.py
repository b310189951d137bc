#!/usr/bin/env python3
"""Generates tests/golden/workload_*.npz and workload_digests.json -- golden values of the benchmark workload
generators of SURVEY.md 8(d) (Plummer sphere, two-galaxy collision; the build's own, not in the reference).

The values come from the NUMPY restatement (rust-exp_amd/presets.py: splitmix64 -> 24-bit f32 samples, float64
arithmetic, one rounding to f32); tests/test_workload_generators.py holds the library's C-ABI generators
(nbx_plummer_sphere / nbx_two_galaxies, host_ops.cpp) to them bit for bit.  Small cases are stored whole, the
benchmark sizes as SHA-256 digests of the little-endian f32 arrays.  Run from the repo root:
    python tests/golden/make_workload_golden.py
"""
import hashlib
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# presets.py alone (numpy only): the library must not be involved in producing its own golden values
spec = importlib.util.spec_from_file_location("presets", os.path.join(ROOT, "rust-exp_amd", "presets.py"))
presets = importlib.util.module_from_spec(spec)
spec.loader.exec_module(presets)

KEYS = ("px", "py", "pz", "vx", "vy", "vz", "m")


def digest(st):
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(st[k], dtype="<f4").tobytes())
    return h.hexdigest()


def main():
    np.savez_compressed(os.path.join(HERE, "workload_plummer_n1000_dim3.npz"), **presets.plummer_sphere(1000, dim=3))
    np.savez_compressed(os.path.join(HERE, "workload_plummer_n1000_dim2.npz"), **presets.plummer_sphere(1000, dim=2))
    np.savez_compressed(os.path.join(HERE, "workload_plummer_n257_seed7.npz"), **presets.plummer_sphere(257, seed=7))
    np.savez_compressed(os.path.join(HERE, "workload_two_galaxies_n1000.npz"), **presets.two_galaxies(1000))
    np.savez_compressed(os.path.join(HERE, "workload_two_galaxies_n7_seed3.npz"), **presets.two_galaxies(7, seed=3))
    dig = {}
    for n, dim in ((65536, 3), (262144, 3), (262144, 2), (1048576, 2)):   # BASELINE configs #2, #3, #4
        dig[f"plummer_n{n}_dim{dim}_seed0x5EED0001"] = digest(presets.plummer_sphere(n, dim=dim))
    dig["two_galaxies_n524288_seed0x5EED0002"] = digest(presets.two_galaxies(524288))   # config #5
    with open(os.path.join(HERE, "workload_digests.json"), "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", sorted(dig))


if __name__ == "__main__":
    main()
