cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && NBX_LONG_STEPS=300 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st7 -o p --output-format csv -- python $OLDPWD/tools/bh_warm_long_run.py random_disk:65536 > $OLDPWD/$O/s27_long.json 2>/dev/null); f=$(find /tmp/st7 -name '*kernel_stats.csv' | head -1); python tools/kstats.py $f | head -16; cut -c1-300 $O/s27_long.json
f=$(find /tmp/st7 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith("nbx::k_chain(")]
print("k_chain per step (every 10th):", " ".join("%.0f" % x for x in v[::10]))
PY
NBX_LOG_CHAINS=1 NBX_LONG_STEPS=300 timeout 300 python - <<'PY' 2>&1 | awk 'NR%25==1' | cut -c1-200
import sys, os
sys.path.insert(0, os.getcwd())
import rust_exp_amd as rx
from rust_exp_amd.engine import NBX_OPT_BH_ASYNC
e = rx.NBodyEngine(); e.set_option(NBX_OPT_BH_ASYNC, 0); e.seed(11); e.random_disk(65536)
for s in range(300): e.step_barnes_hut(0.5, 0.01, 1)
PY
