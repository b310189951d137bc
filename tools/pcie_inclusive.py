#!/usr/bin/env python3
"""What the boundary costs when the caller hands over HOST buffers: set_particles (H2D of the whole state),
get_particles (D2H) and nb_draw's state download, next to the resident step time. A step itself moves no
PCIe bytes; bench.py's `value` is the resident rate."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rust_exp_amd as rx
out = {}
for n in (10000, 262144, 1048576):
    st = rx.plummer_sphere(n)
    e = rx.NBodyEngine()
    def setp():
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"]); e.forces() if False else None
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    e.step_brute_force(0.01); e.synchronize()
    reps = 9

    def med(fn):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    t_set_step = med(lambda: (e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"]), e.step_brute_force(0.01), e.synchronize()))
    t_step = med(lambda: (e.step_brute_force(0.01), e.synchronize()))
    t_step_get = med(lambda: (e.step_brute_force(0.01), e.get_particles()))
    e.set_draw_device(False)
    e.draw(512, 512)                      # first call of either path allocates its buffers: keep that out of the timing
    t_step_draw = med(lambda: (e.step_brute_force(0.01), e.draw(512, 512)))
    e.set_draw_device(True)
    e.draw(512, 512)
    t_step_ddraw = med(lambda: (e.step_brute_force(0.01), e.draw(512, 512)))
    inter = n * (n - 1.0)
    out[str(n)] = {"step_ms": t_step * 1e3, "set+step_ms": t_set_step * 1e3, "step+get_ms": t_step_get * 1e3,
                   "step+host_draw_ms": t_step_draw * 1e3, "step+device_draw_ms": t_step_ddraw * 1e3,
                   "resident_rate": inter / t_step, "rate_with_upload_every_step": inter / t_set_step,
                   "rate_with_download_every_step": inter / t_step_get}
print(json.dumps(out))
