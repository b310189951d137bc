#!/bin/bash
# PMC counters of the Barnes-Hut walk kernel at 1 M bodies (is it vector-issue, scalar-issue or latency bound?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_FLAT" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/bhpmc_$i -o p --output-format csv -- python $OLDPWD/bench.py --workload bh --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/bhpmc_$i.err
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/bhpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_bh_eval" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    rec = {n: sum(v) / len(v) for n, v in c.items()}
    rec["avg_duration_ns"] = sum(dur[k]) / len(dur[k])
    if "GRBM_GUI_ACTIVE" in rec:
        cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
        rec["clock_ghz"] = cyc / rec["avg_duration_ns"]
        if "SQ_ACTIVE_INST_VALU" in rec: rec["valu_busy_frac_per_simd"] = 4.0 * rec["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc
        if "SQ_ACTIVE_INST_SCA" in rec: rec["scalar_busy_frac_per_cu"] = 4.0 * rec["SQ_ACTIVE_INST_SCA"] / 256.0 / cyc
        if "SQ_INST_CYCLES_SALU" in rec: rec["salu_cycles_frac_per_cu"] = 4.0 * rec["SQ_INST_CYCLES_SALU"] / 256.0 / cyc
    out[k] = rec
json.dump(out, open("gpurun_out/r02_bh_walk_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
