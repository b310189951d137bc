#!/bin/bash
# Kernel trace + PMC counters of the Barnes-Hut traversal kernels, both walks (NBX_OPT_BH_WALK 0 / 1), one size.
#   bash tools/bh_walk_pmc.sh [bodies] [theta] [tag]     -> gpurun_out/<tag>_bh_walk_{stats_walkW.csv,pmc.json}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
N=${1:-1048576}; TH=${2:-0.5}; TAG=${3:-r04}
cd /tmp
for W in ${WALKS:-0 2 1}; do
  rm -rf $O/bhw_stats_$W
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/bhw_stats_$W -o p --output-format csv -- python $R/bench.py --workload bh --bodies $N --theta $TH --bh-walk $W --no-cpu-baseline --no-traffic --steady-seconds 0 --steps 10 --warmup 3 > /dev/null 2> $O/bhw_stats_$W.err
  find $O/bhw_stats_$W -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_bh_walk_stats_n${N}_walk$W.csv \;
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ"; do
    i=$((i+1)); rm -rf $O/bhw_pmc_${W}_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/bhw_pmc_${W}_$i -o p --output-format csv -- python $R/bench.py --workload bh --bodies $N --theta $TH --bh-walk $W --no-cpu-baseline --no-traffic --steady-seconds 0 --steps 3 --warmup 1 > /dev/null 2> $O/bhw_pmc_${W}_$i.err
  done
done
cd $R
python - "$N" "$TAG" <<'PY'
import csv, glob, collections, json, sys
N, TAG = sys.argv[1], sys.argv[2]
out = {}
for W in (0, 1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/bhw_pmc_{W}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_bh_" not in k or "count" in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in acc.items():
        rec = {n: sum(v) / len(v) for n, v in c.items()}
        rec["avg_duration_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
        if "GRBM_GUI_ACTIVE" in rec:
            cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
            rec["kernel_cycles"] = cyc
            if "SQ_ACTIVE_INST_VALU" in rec: rec["valu_busy_frac_per_simd"] = 4.0 * rec["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc
            if "SQ_ACTIVE_INST_SCA" in rec: rec["scalar_busy_frac_per_simd"] = 4.0 * rec["SQ_ACTIVE_INST_SCA"] / 1024.0 / cyc
            if "SQ_INST_CYCLES_SALU" in rec: rec["salu_cycles_frac_per_simd"] = 4.0 * rec["SQ_INST_CYCLES_SALU"] / 1024.0 / cyc
        if rec.get("SQ_INSTS_SMEM"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH"):
                if n in rec: rec[n + "_per_smem"] = rec[n] / rec["SQ_INSTS_SMEM"]
        if rec.get("SQ_WAVE_CYCLES") and rec.get("SQ_WAIT_INST_ANY"): rec["wait_inst_any_share"] = rec["SQ_WAIT_INST_ANY"] / rec["SQ_WAVE_CYCLES"]
        if rec.get("SQC_DCACHE_REQ"): rec["scalar_cache_hit_rate"] = rec.get("SQC_DCACHE_HITS", 0.0) / rec["SQC_DCACHE_REQ"]
        if "SQ_WAVE_CYCLES" in rec and "kernel_cycles" in rec: rec["wave_slots_occupied_frac"] = 4.0 * rec["SQ_WAVE_CYCLES"] / (1024.0 * 8.0) / rec["kernel_cycles"]
        if "FETCH_SIZE" in rec: rec["hbm_read_bytes"] = rec["FETCH_SIZE"] * 1024 * 2     # KiB -> B, x2 gfx950 correction (MI355X_MICROARCH.md)
        if "WRITE_SIZE" in rec: rec["hbm_write_bytes"] = rec["WRITE_SIZE"] * 1024
        out[f"walk{W}:{k}"] = rec
json.dump(out, open(f"gpurun_out/{TAG}_bh_walk_pmc_n{N}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
