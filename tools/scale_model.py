#!/usr/bin/env python3
"""What one GPU can say about the 2/4/8-GPU runs it cannot make: the time of ONE rank's share of a G-way shard of the headline
workload (`bench.py --shard-of G`: rank 0's slab x all 262 144 sources, K1 + K2, measured here) plus the MEASURED software
floor of one exchange on this box (tools/exchange_latency.py -> profiles/r03_exchange_latency.json: an RCCL collective with one
rank, and the peer-copy exchange between G engines sharing the GPU).  That is an UPPER BOUND on the speed-up a G-GPU run can
reach -- wire time over xGMI (4 MiB gathered: ~10-30 us) and the skew between ranks come on top -- not an expectation
(round 2 used guessed 40/50/60 us).  Prints one JSON object; the driver's SCALE run is the measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR_FILE = os.path.join(ROOT, "profiles", "r03_exchange_latency.json")


def exchange_floor_us():
    """per G: the larger of the measured RCCL one-rank collective latency and nothing else (the copy exchange's figure is
    reported beside it); falls back to measuring now when the profile file is absent"""
    if os.path.exists(FLOOR_FILE):
        d = json.load(open(FLOOR_FILE))
    else:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_latency.py")], capture_output=True, text=True, timeout=1800)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    rccl = max(d["rccl_world1"].get("exchange_us_per_engine", [0.0]))
    copy = {g: max(d.get(f"copy_{g}", {}).get("exchange_us_per_engine", [0.0])) for g in (2, 4, 8)}
    return rccl, copy


def line(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-traffic", "--steps", "40", "--warmup", "10"] + extra,
                         capture_output=True, text=True, timeout=600)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    base = line([])
    rccl_us, copy_us = exchange_floor_us()
    res = {"single_gpu": {"ms_per_step": base["ms_per_step"], "value": base["value"], "frac": base["roofline"]["frac"]},
           "exchange_floor_us": {"rccl_one_rank_collective": rccl_us, "peer_copy_engines_sharing_one_gpu": copy_us,
                                 "meaning": "measured software floor on this box; a real exchange adds wire time and rank skew"},
           "shards": {}}
    for g in (2, 4, 8):
        d = line(["--shard-of", str(g)])
        step = d["ms_per_step"] + rccl_us * 1e-3
        res["shards"][str(g)] = {"rank_ms_per_step_measured": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_avg_ms"],
                                 "per_gpu_frac": d["roofline"]["frac"], "launch": d["config"]["launch"],
                                 "bound_ms_per_step": step, "speedup_upper_bound": base["ms_per_step"] / step,
                                 "value_upper_bound": 262144.0 * 262143.0 / (step * 1e-3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
