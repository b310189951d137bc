#!/usr/bin/env python3
"""Barnes-Hut leg (BASELINE config #4 and the reference's own published scenario).

  * N = 1 048 576, theta = 0.5: host quadtree build (reference-faithful, stays on the host per the
    north_star) + HIP traversal/eval on 1 GPU; ms/step split host-build / eval; force error vs an
    all-pairs sample.
  * N = 10 000 nb_stable_orbits(0.5, 30), theta = 0.85, dt = 0.01: the scenario behind the only
    numbers the reference publishes (screenshot.png: ~30.75 ms/step, 1 thread, 2016 Mac);
    GPU step vs the oracle on this box's host CPU, median of 30 like RustNBodyExperiment.hs:44,65.
Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def main():
    out = {}
    big = int(os.environ.get("BH_N", "1048576"))
    st = rx.plummer_sphere(big, dim=2)
    e = rx.NBodyEngine(mode="fast")
    e.set_bh_tree("host")                                      # the reference-faithful insertion build (round 2: no longer the default)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
    e.step_barnes_hut(0.5, 0.01, 1); e.synchronize()          # warm-up (allocations)
    e.profile(True); e.profile_reset(); e.bh_host_timing()

    def step():
        e.step_barnes_hut(0.5, 0.01, 1); e.synchronize()

    med, ts = timed(step, 5)
    ms, cnt = e.profile_read(rx.NBX_K_BH_EVAL)
    ims, _ = e.profile_read(rx.NBX_K_INTEGRATE)
    out["bh_1m"] = {"bodies": big, "theta": 0.5, "ms_per_step_median": med * 1e3, "eval_kernel_ms": ms / cnt,
                    "integrate_kernel_ms": ims / cnt, "host_tree_and_copies_ms": med * 1e3 - ms / cnt - ims / cnt,
                    "steps_timed": len(ts)}
    out["bh_1m"]["host_phases"] = e.bh_host_timing()
    e.profile(False)
    # force error of theta=0.5 vs all-pairs on this state
    bx, by, _ = e.forces(0.5)
    fx, fy, _ = e.forces(0.0)
    rel = np.hypot(bx - fx, by - fy) / (np.hypot(fx, fy) + 1e-20)
    out["bh_1m"]["force_rel_err_median"] = float(np.median(rel))
    out["bh_1m"]["force_rel_err_p99"] = float(np.percentile(rel, 99))

    # the same step with the quadtree built on the device (bh_build.hip): no host round trip
    cur = e.get_particles()            # the state the host-tree errors above were measured on
    d = rx.NBodyEngine(mode="fast")
    d.set_bh_tree("device")
    d.set_particles(cur["px"], cur["py"], cur["vx"], cur["vy"], cur["m"])
    dx, dy, _ = d.forces(0.5)
    rel = np.hypot(dx - fx, dy - fy) / (np.hypot(fx, fy) + 1e-20)
    relh = np.hypot(dx - bx, dy - by) / (np.hypot(bx, by) + 1e-20)
    dev_err = {"force_rel_err_median": float(np.median(rel)), "force_rel_err_p99": float(np.percentile(rel, 99)),
               "vs_host_tree_rel_median": float(np.median(relh)), "vs_host_tree_rel_p999": float(np.percentile(relh, 99.9))}
    d.step_barnes_hut(0.5, 0.01, 1); d.synchronize()
    d.profile(True); d.profile_reset(); d.bh_host_timing()

    def dstep():
        d.step_barnes_hut(0.5, 0.01, 1); d.synchronize()

    dmed, dts = timed(dstep, 10)
    dms, dcnt = d.profile_read(rx.NBX_K_BH_EVAL)
    ht = d.bh_host_timing()
    out["bh_1m_device_tree"] = {"ms_per_step_median": dmed * 1e3, "eval_kernel_ms": dms / dcnt, "tree_build_ms": ht["build_ms"],
                                "nodes": ht["nodes"], "steps_timed": len(dts)}
    d.profile(False)
    out["bh_1m_device_tree"].update(dev_err)
    from rust_exp_amd.engine import NBX_OPT_BH_WAVE
    d.set_option(NBX_OPT_BH_WAVE, 0)
    d.profile(True); d.profile_reset()
    for _ in range(5):
        dstep()
    pms, pcnt = d.profile_read(rx.NBX_K_BH_EVAL)
    out["bh_1m_device_tree"]["eval_kernel_ms_per_lane_walk"] = pms / pcnt
    d.profile(False)
    d.set_option(NBX_OPT_BH_WAVE, 1)
    wk = d.bh_work(0.5)
    kms = out["bh_1m_device_tree"]["eval_kernel_ms"]
    out["bh_1m_device_tree"]["work"] = {
        "node_visits_per_body": wk["node_visits"] / big, "pair_evals_per_body": wk["pair_evals"] / big,
        "node_bytes_per_launch": wk["node_visits"] * 32.0,
        "node_read_rate_GBps": wk["node_visits"] * 32.0 / (kms * 1e-3) / 1e9,
        "visits_per_second": wk["node_visits"] / (kms * 1e-3)}

    if not os.environ.get("BH_NO_CPU"):
        import bench   # the CPU-baseline legs live in bench.py (the only non-test code that may run the oracle)

        ms1, rc = bench.cpu_baseline_barnes_hut(st, 0.5, 0.01, 16, 1)
        out["bh_1m_cpu_oracle_16threads"] = {"ms_per_step": ms1, "rc": rc,
                                             "note": "oracle restatement of nbody.rs:186-480, serial tree build + 16 traversal threads (the caller's maximum, hs:94-97)"}
    # the reference's published scenario
    e2 = rx.NBodyEngine(mode="fast")
    e2.seed(1); e2.stable_orbits(10000, 0.5, 30.0)
    s0 = e2.get_particles()

    def step2():
        e2.step_barnes_hut(0.85, 0.01, 1); e2.synchronize()

    step2()
    med2, _ = timed(step2, 30)
    out["bh_10k_gpu"] = {"bodies": 10000, "theta": 0.85, "ms_per_step_median_of_30": med2 * 1e3}
    if not os.environ.get("BH_NO_CPU"):
        import bench

        med3, _ = bench.cpu_baseline_barnes_hut(s0, 0.85, 0.01, 1, 31)
        out["bh_10k_cpu_oracle_1thread"] = {"ms_per_step_median_of_30": med3,
                                            "reference_published_ms": 30.75, "note": "screenshot.png, unknown 2016 Mac"}
        threads = min(bench.effective_cores(), 16)
        med4, _ = bench.cpu_baseline_barnes_hut(s0, 0.85, 0.01, threads, 31)
        out["bh_10k_cpu_oracle_16threads"] = {"ms_per_step_median_of_30": med4, "threads": threads}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
