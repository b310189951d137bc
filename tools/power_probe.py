#!/usr/bin/env python3
"""Board power while K1 runs: is the all-pairs kernel power-limited?  

Runs the brute-force step of the bench workload (N = 262 144 Plummer, 3-D) back to back for >= SECONDS seconds and samples,
from a thread, every source of socket power this box exposes:
  * `rocm-smi --showpower --showclocks --json`                              -- as fast as the tool answers
    (round 3: the amdgpu hwmon power1_input series is gone -- it read a flat 248 W beside rocm-smi's 1350 W, the wrong
    sensor; only the board's power cap is still taken from hwmon)
  * `amd-smi metric --power --clock --json`                                 -- idem
Prints one JSON object: average / max watts per source, the power cap if the box reports one, interactions/s,
joules per interaction, and the shader clock the tools saw.  Usage: python tools/power_probe.py [SECONDS] [--variant V] [--idle]
"""
import glob
import json
import os
import shutil
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_exp_amd as rx  # noqa: E402


def hwmon_nodes():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        rec = {"dir": d}
        for name in ("power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                rec[name] = p
        out.append(rec)
    return out


def read_int(path):
    try:
        return int(open(path).read())
    except (OSError, ValueError):
        return None


def find_numbers(obj, want):
    """every numeric leaf of a parsed JSON tree whose key path mentions one of `want` (lower case)"""
    found = []

    def walk(o, path):
        if isinstance(o, dict):
            for k, v in o.items():
                walk(v, path + "/" + str(k).lower())
        elif isinstance(o, list):
            for v in o:
                walk(v, path)
        else:
            if any(w in path for w in want):
                try:
                    found.append((path, float(str(o).split()[0])))
                except (ValueError, IndexError):
                    pass
    walk(obj, "")
    return found


class ToolSampler(threading.Thread):
    def __init__(self, name, cmd):
        super().__init__(daemon=True)
        self.name_, self.cmd, self.rows, self.stop_ = name, cmd, [], threading.Event()
        self.raw_first = None

    def run(self):
        while not self.stop_.is_set():
            t = time.perf_counter()
            try:
                r = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20)
                txt = r.stdout
                if self.raw_first is None:
                    self.raw_first = txt[:4000]
                starts = [i for i in (txt.find("{"), txt.find("[")) if i >= 0]
                js = json.loads(txt[min(starts):])
                self.rows.append((t, js))
            except Exception:   # noqa: BLE001 (a probe: any tool failure just means no sample)
                pass
            self.stop_.wait(0.05)


def main():
    argv = [a for a in sys.argv[1:]]
    seconds = float(argv[0]) if argv and not argv[0].startswith("-") else 6.0
    variant = int(argv[argv.index("--variant") + 1]) if "--variant" in argv else -1
    idle = "--idle" in argv
    n = 262144
    st = rx.plummer_sphere(n)
    e = rx.NBodyEngine()
    e.set_launch(variant=variant)
    e.set_particles(st["px"], st["py"], st["vx"], st["vy"], st["m"], st["pz"], st["vz"])
    for _ in range(3):
        e.step_brute_force(0.01)
    e.synchronize()
    nodes = hwmon_nodes()
    node = nodes[0] if nodes else {}
    samplers = []
    if shutil.which("rocm-smi"):
        samplers.append(ToolSampler("rocm-smi", ["rocm-smi", "--showpower", "--showclocks", "--json"]))
    if shutil.which("amd-smi"):
        samplers.append(ToolSampler("amd-smi", ["amd-smi", "metric", "--power", "--clock", "--json"]))
    for s in samplers:
        s.start()
    t0 = time.perf_counter()
    steps = 0
    if idle:
        time.sleep(seconds)
    else:
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                e.step_brute_force(0.01)
            e.synchronize()
            steps += 10
    t1 = time.perf_counter()
    for s in samplers:
        s.stop_.set()
    for s in samplers:
        s.join(timeout=30)
    inter = float(n) * (n - 1) * steps
    out = {"workload": f"plummer N={n} 3-D brute force, back-to-back steps for {t1 - t0:.2f} s", "steps": steps,
           "interactions_per_s": inter / (t1 - t0) if steps else 0.0, "launch": e.last_launch()}
    if "power1_cap" in node:
        out["power_cap_w"] = (read_int(node["power1_cap"]) or 0) * 1e-6
    if "power1_cap_max" in node:
        out["power_cap_max_w"] = (read_int(node["power1_cap_max"]) or 0) * 1e-6
    # samples of the first 0.5 s (ramp) are skipped
    for s in samplers:
        pw, ck = [], []
        for (t, js) in s.rows:
            if not (t0 + 0.5 <= t <= t1):
                continue
            pw += [v for (p, v) in find_numbers(js, ("power",)) if 5.0 < v < 5000.0 and "cap" not in p and "limit" not in p]
            ck += [v for (p, v) in find_numbers(js, ("sclk", "gfx")) if 100.0 < v < 5000.0]
        rec = {"samples": len(s.rows), "first_output": (s.raw_first or "")[:1500]}
        if pw:
            rec.update({"avg_w": float(np.mean(pw)), "max_w": float(np.max(pw)),
                        "joules_per_interaction": float(np.mean(pw)) * (t1 - t0) / inter if steps else None})
        if ck:
            rec.update({"clock_mhz_avg": float(np.mean(ck)), "clock_mhz_min": float(np.min(ck)), "clock_mhz_max": float(np.max(ck))})
        out[s.name_] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
