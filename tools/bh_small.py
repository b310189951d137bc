import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx
out = {}
for n in (10000, 100000):
    for where in ("host", "device"):
        e = rx.NBodyEngine(); e.set_bh_tree(where); e.seed(1); e.stable_orbits(n, 0.5, 30.0)
        for _ in range(3): e.step_barnes_hut(0.85, 0.01, 1)
        e.synchronize(); e.bh_host_timing(); e.profile(True); e.profile_reset()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); e.step_barnes_hut(0.85, 0.01, 1); e.synchronize(); ts.append(time.perf_counter() - t0)
        ms, cnt = e.profile_read(rx.NBX_K_BH_EVAL)
        out[f"{n}_{where}"] = {"median_ms": float(np.median(ts)) * 1e3, "eval_ms": ms / cnt, "host": e.bh_host_timing()}
print(json.dumps(out))
