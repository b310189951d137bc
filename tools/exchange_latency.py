#!/usr/bin/env python3
"""What ONE GPU can measure of the per-step exchange of an N-GPU group (VERDICT r02 next #1d): the software floor.

  rccl_world1   a group of ONE device exchanging through RCCL: ncclCommInitAll(1) + one in-place ncclAllGather of the whole
                (x,y,z,m) array per step -- the launch + completion latency of an RCCL collective on this box, no wire time
  copy_G        G engines sharing the device, NBX_GROUP_EXCHANGE=copy: per engine G-1 event waits + G-1 hipMemcpyPeerAsync pulls
                of N/G float4 + one event record per step -- the enqueue + event latency of the fallback exchange (the copies
                are device-local here: HBM speed, not xGMI)

Per exchange: microseconds between the HIP events the library records around it on each engine's stream
(NBX_K_EXCHANGE), and the host wall time of a zero-dt step minus the same step without exchange.  These are LOWER bounds
for a real multi-GPU exchange (which adds wire time and the skew between ranks): tools/scale_model.py uses them as such.
Prints one JSON object.   Usage: python tools/exchange_latency.py [N]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, os.environ["NBX_ROOT"])
import numpy as np
import rust_exp_amd as rx
G, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
st = rx.plummer_sphere(n)
g = rx.NBodyGroup([0] * G)
g.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
for _ in range(5):
    g.step_brute_force(0.0)          # dt = 0: bodies stay, every kernel and the exchange still run
g.synchronize()
engines = [g.engine(i) for i in range(G)]
for e in engines:
    e.profile(True); e.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    g.step_brute_force(0.0)
g.synchronize()
t1 = time.perf_counter()
ex = [e.profile_read(rx.NBX_K_EXCHANGE) for e in engines]
k1 = [e.profile_read(rx.NBX_K_FORCE) for e in engines]
k2 = [e.profile_read(rx.NBX_K_INTEGRATE) for e in engines]
print("RESULT " + json.dumps({"G": G, "n": n, "reps": reps, "info": g.info(),
      "exchange_us_per_engine": [1e3 * ms / max(c, 1) for ms, c in ex],
      "k1_ms_per_engine": [ms / max(c, 1) for ms, c in k1], "k2_ms_per_engine": [ms / max(c, 1) for ms, c in k2],
      "wall_ms_per_step": 1e3 * (t1 - t0) / reps}))
"""


def run(G, n, reps, env_extra):
    env = dict(os.environ, NBX_ROOT=ROOT, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD, str(G), str(n), str(reps)], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stdout + r.stderr)[-1500:]}
    return json.loads(lines[-1][7:])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    out = {"n": n, "what": "software floor of the per-step exchange, one GPU (see the docstring of tools/exchange_latency.py)"}
    out["rccl_world1"] = run(1, n, 200, {})
    for G in (2, 4, 8):
        out[f"copy_{G}"] = run(G, n, 100, {"NBX_GROUP_EXCHANGE": "copy"})
        out[f"copy_{G}_threads"] = run(G, n, 100, {"NBX_GROUP_EXCHANGE": "copy", "NBX_GROUP_ENQUEUE": "threads"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
