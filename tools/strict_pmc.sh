#!/bin/bash
# PMC counters of the bit-exact all-pairs kernel at N = 262 144 (or BODIES=..., KERNEL=0|1|8|16)
BODIES=${BODIES:-262144}; export BODIES
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACCUM_PREV_HIRES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/stpmc_$i -o p --output-format csv -- python $OLDPWD/bench.py --mode strict --strict-kernel ${KERNEL:-0} --bodies $BODIES --no-cpu-baseline --no-traffic --steps 3 --warmup 1 > /dev/null 2> $O/stpmc_$i.err
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections, json, os
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/stpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_force_strict" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    rec = {n: sum(v) / len(v) for n, v in c.items()}
    rec["avg_duration_ns"] = sum(dur[k]) / len(dur[k])
    pairs = float(os.environ['BODIES']) ** 2
    if "SQ_INSTS_VALU" in rec: rec["valu_per_64_pairs"] = rec["SQ_INSTS_VALU"] / (pairs / 64)
    if "SQ_INSTS_LDS" in rec: rec["lds_per_64_pairs"] = rec["SQ_INSTS_LDS"] / (pairs / 64)
    if "GRBM_GUI_ACTIVE" in rec:
        cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
        rec["clock_ghz"] = cyc / rec["avg_duration_ns"]
        # busy fraction of the SIMDs that have work: a launch with fewer workgroups than CUs leaves the rest idle
        wgs = float(os.environ["BODIES"]) / (256.0 if "pc" not in k else 64.0)
        simds = 4.0 * min(256.0, wgs)
        rec["simds_with_work"] = simds
        if "SQ_ACTIVE_INST_VALU" in rec: rec["valu_busy_frac"] = 4.0 * rec["SQ_ACTIVE_INST_VALU"] / simds / cyc
        if "SQ_INSTS_VALU" in rec: rec["cycles_per_valu_inst_per_simd"] = cyc / (rec["SQ_INSTS_VALU"] / 1024.0)
    out[k] = rec
json.dump(out, open("gpurun_out/%s_strict_pmc_%s.json" % (os.environ.get("TAG", "r02"), os.environ["BODIES"]), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
