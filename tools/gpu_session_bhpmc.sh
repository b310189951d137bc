#!/bin/bash
# PMC passes over the Barnes-Hut device-tree step (one counter group per run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python $R/tools/bh_steps.py device 6"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/bhpmc_sq -o p -- $B > $R/gpurun_out/bhpmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/bhpmc_grbm -o p -- $B > $R/gpurun_out/bhpmc_grbm.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, json
out = {}
for tag in ("sq", "grbm"):
    for f in glob.glob(f"gpurun_out/bhpmc_{tag}/**/p_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open("gpurun_out/bhpmc_summary.json", "w"), indent=1)
PY
rm -rf gpurun_out/bhpmc_sq gpurun_out/bhpmc_grbm
