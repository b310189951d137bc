#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- the committed golden vectors for the N-body hot path.

The reference (rs-src/nbody.rs) ships no tests/fixtures and cannot be compiled in this image
(Rust), so these vectors come from the line-faithful C restatement oracle/nbody_oracle.c and are
cross-checked here, bit for bit, against the independent numpy-float32 restatement
oracle/nbody_numpy.py before being written ("parity unpinned" by the reference itself; see
DESIGN.md section 3).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from oracle import nbody_numpy as onp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DT = 0.01  # RustNBodyExperiment.hs:45


def fields(p):
    return {k: np.array(p[k]) for k in ("px", "py", "vx", "vy", "m")}


def brute_case(name, p0, steps_list):
    rec = {"in_" + k: v for k, v in fields(p0).items()}
    p = p0.copy()
    px, py, vx, vy = (p0[k].copy() for k in ("px", "py", "vx", "vy"))
    done = 0
    for s in sorted(steps_list):
        while done < s:
            assert ob.step_brute_force(p, DT) == 0
            px, py, vx, vy = onp.step_brute_force(px, py, vx, vy, p0["m"], DT)
            done += 1
        # pin: C oracle == numpy restatement, bit for bit
        for k, a in (("px", px), ("py", py), ("vx", vx), ("vy", vy)):
            assert np.array_equal(a.view(np.uint32), p[k].view(np.uint32)), (name, s, k)
        for k, v in fields(p).items():
            if k != "m":
                rec[f"s{s}_{k}"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), dt=np.float32(DT), **rec)
    print("wrote", name, len(p0))


def bh_case(name, p0, theta, steps_list):
    rec = {"in_" + k: v for k, v in fields(p0).items()}
    p = p0.copy()
    done = 0
    for s in sorted(steps_list):
        while done < s:
            assert ob.step_barnes_hut(p, theta, DT, 1) == 0
            done += 1
        for k, v in fields(p).items():
            if k != "m":
                rec[f"s{s}_{k}"] = v
    rc, fx, fy = ob.bh_forces(p0, theta)
    assert rc == 0
    rec["f0_x"], rec["f0_y"] = fx, fy
    np.savez_compressed(os.path.join(OUT, name + ".npz"), dt=np.float32(DT), theta=np.float32(theta), **rec)
    print("wrote", name, len(p0))


def main():
    # brute force: tiny hand cases + the reference's two presets (seeded)
    two = ob.particles([-1.0, 1.0], [0.0, 0.0], [0.0, 0.0], [0.5, -0.5], [2.0, 2.0])
    brute_case("brute_n2", two, [1, 10])
    brute_case("brute_n5_orbits", ob.stable_orbits(5, 5.0, 40.0, 5), [1, 10])      # hs:87 preset
    brute_case("brute_n64_disk", ob.random_disk(64, 64), [1, 10])
    brute_case("brute_n1024_orbits", ob.stable_orbits(1024, 0.5, 30.0, 1), [1, 10])  # BASELINE config #1
    brute_case("brute_n1000_disk", ob.random_disk(1000, 3), [1, 10])                # ragged (not a tile multiple)
    # Barnes-Hut
    for theta in (0.5, 0.85):
        tag = str(theta).replace(".", "p")
        bh_case(f"bh_n64_disk_t{tag}", ob.random_disk(64, 64), theta, [1, 10])
        bh_case(f"bh_n1024_orbits_t{tag}", ob.stable_orbits(1024, 0.5, 30.0, 1), theta, [1, 10])
        bh_case(f"bh_n1000_disk_t{tag}", ob.random_disk(1000, 3), theta, [1, 10])
    # draw
    p = ob.stable_orbits(1024, 0.5, 30.0, 1)
    np.savez_compressed(os.path.join(OUT, "draw_n1024_orbits.npz"), fb_64x48=ob.draw(p, 64, 48),
                        fb_512x512=ob.draw(p, 512, 512), **{"in_" + k: v for k, v in fields(p).items()})
    # presets (seeded generator; distribution only is reference-pinned)
    np.savez_compressed(os.path.join(OUT, "presets.npz"),
                        disk_seed7_n257=np.array(ob.random_disk(257, 7)).view(np.float32).reshape(-1, 5),
                        orbits_seed9_n100=np.array(ob.stable_orbits(100, 0.5, 30.0, 9)).view(np.float32).reshape(-1, 5))
    print("done")


if __name__ == "__main__":
    main()
