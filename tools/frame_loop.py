#!/usr/bin/env python3
"""The reference's per-frame call sequence, driven through the six level-1 symbols exactly as
hs-src/RustNBodyExperiment.hs does (withExperiment :42-48, experimentDraw :50-62, status line :63-80):

    nb_stable_orbits 10000 0.5 30.0
    every frame:  timeIt(nb_step_barnes_hut theta dt nthreads) ; nb_draw 512 512 fb
    status:       "%i Steps, %.1fSPS/%.2fms | %s Bodies" with the MEDIAN of the last 30 step times

Prints that status line for the default scene and the other key-bound scenes (Q/W/E, hs:85-87).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NB_SEED", "1")
import rust_exp_amd as rx  # noqa: E402


def run(scene, init, theta=0.85, dt=0.01, nthreads=1, frames=60):
    init()
    times, draw_times = [], []
    for _ in range(frames):
        t0 = time.perf_counter()
        rx.nb_step_barnes_hut(theta, dt, nthreads)          # hs:55-57 (wall-clocked by the caller)
        t1 = time.perf_counter()
        fb = rx.nb_draw(512, 512)                           # hs:58-60
        t2 = time.perf_counter()
        times.append(t1 - t0); draw_times.append(t2 - t1)
    avg = float(np.median(times[-30:]))                     # hs:44,:65
    n = rx.nb_num_particles()
    bodies = f"{n // 1000}K" if n > 999 else str(n)
    print(f"[{scene}] {frames} Steps, {1 / avg:.1f}SPS/{avg * 1000:.2f}ms | {bodies} Bodies | Time Step: {dt:.4f} | "
          f"Theta: {theta:.2f} | Threads: {nthreads} | nb_draw {np.median(draw_times) * 1000:.2f}ms | lit pixels {(fb != 0).sum()}")


if __name__ == "__main__":
    run("Q stable orbits 10K (default)", lambda: rx.nb_stable_orbits(10000, 0.5, 30.0))
    run("W random disk 10K", lambda: rx.nb_random_disk(10000))
    run("E 5 bodies", lambda: rx.nb_stable_orbits(5, 5.0, 40.0))
    run("theta 0 -> brute force 10K", lambda: rx.nb_stable_orbits(10000, 0.5, 30.0), theta=0.0)
    run("stable orbits 1M", lambda: rx.nb_stable_orbits(1000000, 0.5, 30.0), frames=12)
