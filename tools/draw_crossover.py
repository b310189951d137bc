import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_exp_amd as rx
for n in (64, 256, 512, 1000, 2000, 4000, 8000):
    st = rx.plummer_sphere(n, dim=2)
    row = {"n": n}
    for name, dev in (("host", False), ("device", True)):
        e = rx.NBodyEngine()
        e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"])
        e.set_draw_device(dev)
        for _ in range(3):
            e.step_barnes_hut(0.85, 0.01, 1); e.draw(512, 512)
        t = []
        for _ in range(30):
            e.step_barnes_hut(0.85, 0.01, 1); e.synchronize()
            t0 = time.perf_counter(); e.draw(512, 512); t.append(time.perf_counter() - t0)
        row[name + "_draw_ms"] = round(float(np.median(t)) * 1e3, 4)
    print(json.dumps(row), flush=True)
