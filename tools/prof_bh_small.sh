#!/bin/bash
# per-kernel times of the Barnes-Hut step of a small system (rocprofv3 --kernel-trace --stats), both tree classes
#   bash tools/prof_bh_small.sh [bodies] [tag]   -> gpurun_out/<tag>_bh_small_kernel_stats_<fold>_<bodies>.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
N=${1:-10000}; TAG=${2:-r04}
cd /tmp
for FOLD in exact reference; do
  rm -rf $O/bhs_$FOLD
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/bhs_$FOLD -o p --output-format csv -- python $R/tools/bh_small_steps.py $N 200 $FOLD > /dev/null 2> $O/bhs_$FOLD.err
  find $O/bhs_$FOLD -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_bh_small_kernel_stats_${FOLD}_$N.csv \;
  python - "$O/${TAG}_bh_small_kernel_stats_${FOLD}_$N.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[1].split("/")[-1])
for r in rows:
    if int(r["Calls"]) >= 100:
        print("   %-52s calls %5s  avg %7.1f us" % (r["Name"].split("(")[0].replace("void ", "")[-52:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
