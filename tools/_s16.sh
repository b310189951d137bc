cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st5 -o p --output-format csv -- python $OLDPWD/tools/bh_reference_fold_async_probe.py > $OLDPWD/$O/s16_probe.log 2>&1); tail -2 $O/s16_probe.log | cut -c1-400
f=$(find /tmp/st5 -name '*kernel_stats.csv' | head -1); python tools/kstats.py $f | head -12
f=$(find /tmp/st5 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r) for r in rows]
for d, r in sorted(big, key=lambda x: -x[0])[:8]:
    print("%.2f ms" % (d / 1e6), r["Kernel_Name"][:60], "at %.1f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6))
# gaps
prev = None
gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and s - prev[0] > 50e6: gaps.append(((s - prev[0]) / 1e6, prev[1], r["Kernel_Name"][:40], (s - t0) / 1e6))
    prev = (e, r["Kernel_Name"][:40])
for g in gaps[:10]: print("gap %.1f ms after %s before %s at %.1f ms" % g)
PY
