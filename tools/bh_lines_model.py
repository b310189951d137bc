#!/usr/bin/env python3
"""Runs tools/bh_lines_model.c on the flattened quadtree of BASELINE config #4 (1 048 576-body Plummer disc, theta = 0.5), built by
the library's reference-faithful host build (no GPU needed): which node records / 128-byte lines each XCD-sized eighth of the
bodies touches, and how the visits spread over the tree's depth.  Usage: python tools/bh_lines_model.py [n] [theta] > profiles/r03_bh_walk_lines_model.json"""
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
    e.plummer_sphere(n, dim=2)
    flat = e.bh_flat_dump(True)
    tmp = tempfile.mkdtemp(prefix="nbx_lines_")
    exe = os.path.join(tmp, "bh_lines_model")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", os.path.join(ROOT, "tools", "bh_lines_model.c"), "-o", exe, "-lm"])
    path = os.path.join(tmp, "nodes.bin")
    flat.tofile(path)
    sys.stdout.write(subprocess.check_output([exe, path, str(len(flat)), str(theta)]).decode())


if __name__ == "__main__":
    main()
