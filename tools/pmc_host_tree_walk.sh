cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --traffic-child --workload bh --bh-tree host --no-cpu-baseline --no-traffic > /dev/null 2>&1
  python - <<PY
import csv,glob
rows=[]
for f in glob.glob("/tmp/pmc_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="$c" and ("k_bh_walk_groups" in r["Kernel_Name"] or "k_bh_groups(" in r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:30], float(r["Counter_Value"])))
for r in sorted(rows): print("$c", r)
PY
done
