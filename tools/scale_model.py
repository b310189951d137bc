#!/usr/bin/env python3
"""What one GPU can say about the 2/4/8-GPU runs it cannot make: the time of ONE rank's share of a G-way shard of the headline
workload (`bench.py --shard-of G`: rank 0's slab x all 262 144 sources, K1 + K2, measured here), plus a stated allowance for the
per-step all-gather (not measured: no multi-GPU box was reachable), gives the step time and the speed-up a G-GPU run would
reach if RCCL behaves.  Prints one JSON object; the driver's SCALE run is the measurement, this is the expectation."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLGATHER_US = {2: 40.0, 4: 50.0, 8: 60.0}   # allowance: 4 MiB gathered over xGMI is ~10-30 us of wire time + launch/sync latency


def line(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-traffic", "--steps", "40", "--warmup", "10"] + extra,
                         capture_output=True, text=True, timeout=600)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    base = line([])
    res = {"single_gpu": {"ms_per_step": base["ms_per_step"], "value": base["value"], "frac": base["roofline"]["frac"]},
           "allgather_allowance_us": ALLGATHER_US, "shards": {}}
    for g in (2, 4, 8):
        d = line(["--shard-of", str(g)])
        step = d["ms_per_step"] + ALLGATHER_US[g] * 1e-3
        res["shards"][str(g)] = {"rank_ms_per_step_measured": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_avg_ms"],
                                 "per_gpu_frac": d["roofline"]["frac"], "launch": d["config"]["launch"],
                                 "expected_ms_per_step": step, "expected_speedup": base["ms_per_step"] / step,
                                 "expected_value": 262144.0 * 262143.0 / (step * 1e-3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
