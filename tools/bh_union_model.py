#!/usr/bin/env python3
"""Runs tools/bh_union_model.c on the flattened quadtree of a Plummer disc built by the library's reference-faithful host build (no
GPU needed): turns, child visits and lane occupancy of the child-group walk when W = 1 ... 256 Morton-adjacent bodies share one
walk.  Usage: python tools/bh_union_model.py [n] [theta] > profiles/r04_bh_walk_union_model_n<N>.json"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
    theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    e = rx.NBodyEngine()
    if n == 10000:
        e.seed(1); e.stable_orbits(n, 0.5, 30.0)      # the reference's own scene
    else:
        e.plummer_sphere(n, dim=2)
    flat = e.bh_flat_dump(True)
    tmp = tempfile.mkdtemp(prefix="nbx_union_")
    exe = os.path.join(tmp, "bh_union_model")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", os.path.join(ROOT, "tools", "bh_union_model.c"), "-o", exe, "-lm"])
    path = os.path.join(tmp, "nodes.bin")
    flat.tofile(path)
    sys.stdout.write(subprocess.check_output([exe, path, str(len(flat)), str(theta)]).decode())


if __name__ == "__main__":
    main()
