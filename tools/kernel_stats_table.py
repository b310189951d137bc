import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
