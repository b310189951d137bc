#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes under gpurun_out/pmc_* into profiles/ (per round).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes (TCC slots), unit = KiB, and on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of a wide (16 B/lane) coalesced streaming read -> doubled here; WRITE_SIZE is used as
reported (uncalibrated per the guide; it matches the kernel's store bytes exactly here).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    p = os.path.join(SRC, d, "p_counter_collection.csv")
    if not os.path.exists(p):
        return acc, dur
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc, dur


allc = collections.defaultdict(dict)
durs = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_grbm", "pmc_tcc"):
    acc, dur = load(d)
    for k, v in acc.items():
        for c, x in v.items():
            allc[k][c] = sum(x) / len(x)
        durs.setdefault(k, sum(dur[k]) / len(dur[k]))

out = {"source": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python bench.py --no-cpu-baseline --no-traffic --steps 5 --warmup 1 "
                 "(tools/gpu_session.sh pmc / k1pmc)",
       "corrections": "FETCH_SIZE and WRITE_SIZE in separate passes, unit KiB; gfx950: FETCH_SIZE x 2 for wide coalesced reads "
                      "(MI355X_MICROARCH.md, HBM section); GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = kernel cycles); "
                      "SQ_ACTIVE_INST_VALU counts quad-cycles (x 4 / 1024 SIMDs / kernel cycles = valu_busy_frac)",
       "kernels": {}}
for k, c in allc.items():
    rec = dict(c)
    if "FETCH_SIZE" in c:
        rec["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0   # gfx950 x2 correction
    if "WRITE_SIZE" in c:
        rec["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes_per_launch" in rec and "hbm_write_bytes_per_launch" in rec:
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
    if "GRBM_GUI_ACTIVE" in c and k in durs:
        rec["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / durs[k]   # counter is summed over the 8 XCDs
        cyc_per_simd = c["GRBM_GUI_ACTIVE"] / 8.0
        if "SQ_ACTIVE_INST_VALU" in c:
            # SQ_ACTIVE_INST_* count quad-cycles (guide, per-instruction table); 1024 SIMDs
            rec["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc_per_simd
        if "SQ_INSTS_VALU" in c:
            rec["cycles_per_valu_inst_per_simd"] = cyc_per_simd / (c["SQ_INSTS_VALU"] / 1024.0)
    if "TCC_HIT_sum" in c:
        rec["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    rec["avg_duration_ns_profiled"] = durs.get(k)
    out["kernels"][k] = rec

force = [k for k in out["kernels"] if "k_force" in k]
if force:
    out["k_force_tile_hbm_bytes_per_launch"] = out["kernels"][force[0]].get("hbm_bytes_per_launch")
    out["dominant_kernel"] = force[0]
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
json.dump({"k_force_tile_hbm_bytes_per_launch": out.get("k_force_tile_hbm_bytes_per_launch"),
           "dominant_kernel": out.get("dominant_kernel"), "from": f"profiles/{tag}_pmc_summary.json"},
          open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
