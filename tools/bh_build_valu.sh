#!/bin/bash
# Instruction issue of every kernel of a Barnes-Hut step at 1 M bodies: VALU / SALU / memory instructions per wave, waves, busy
# cycles (rocprofv3 --pmc, one pass per group).  One JSON object on stdout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_bv_$i
  rocprofv3 --kernel-trace --pmc $g -d /tmp/pmc_bv_$i -o p --output-format csv -- python $R/bench.py --workload bh --no-cpu-baseline --no-traffic --no-accuracy --steps 8 --warmup 3 --steady-seconds 0 > /dev/null 2>&1
  i=$((i+1))
done
python - <<'PY'
import csv, glob, json, collections
out = collections.OrderedDict()
for f in glob.glob("/tmp/pmc_bv_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nbx::", "")
        if "rocprim" in name or "amd_rocclr" in name: continue
        per[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in per.items():
        v = v[len(v) // 3:]
        out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
for k, d in out.items():
    w = d.get("SQ_WAVES", 0)
    if w:
        d["valu_per_wave"] = round(d.get("SQ_INSTS_VALU", 0) / w, 1)
        d["salu_per_wave"] = round(d.get("SQ_INSTS_SALU", 0) / w, 1)
print(json.dumps(out, indent=1))
PY
