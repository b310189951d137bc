#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: name (truncated), calls, average us, total ms."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"].replace("\n", " ")
    for pre in ("void ", "nbx::"):
        if name.startswith(pre): name = name[len(pre):]
    if "rocprim" in name:
        name = "rocprim:" + ("block_merge#2" if "lambda(auto:1)#2" in name else "block_merge#1" if "merge_sort_block_merge" in name else "block_sort" if "block_sort" in name else name[:40])
    print("%-46s %6s %10.1f us %9.3f ms" % (name[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
