#!/bin/bash
# HBM bytes per launch of every kernel of a Barnes-Hut step at 1 M bodies (device tree, warm sort): rocprofv3 --pmc FETCH_SIZE and
# WRITE_SIZE in separate passes, KiB -> bytes, FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md), averaged over the
# warm launches.  One JSON object on stdout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bt_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_bt_$c -o p --output-format csv -- python $R/bench.py --workload bh --no-cpu-baseline --no-traffic --steps 12 --warmup 3 --steady-seconds 0 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = collections.OrderedDict()
for c, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    per = collections.defaultdict(list)
    for f in glob.glob("/tmp/pmc_bt_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nbx::", "")
                if "rocprim" in name or "amd_rocclr" in name: continue
                per[name].append(float(r["Counter_Value"]) * scale)
    for k, v in per.items():
        v = v[len(v) // 3:]            # the warm launches
        out.setdefault(k, {})["read_MB" if c == "FETCH_SIZE" else "write_MB"] = round(sum(v) / len(v) / 1e6, 2)
        out[k]["launches"] = len(v)
print(json.dumps(out, indent=1))
PY
