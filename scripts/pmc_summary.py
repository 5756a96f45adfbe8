"""Summarise rocprofv3 --pmc counter_collection.csv files: per (kernel name, counter), calls / total / average."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not files:
        print("no counter file under", d)
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (r["Kernel_Name"][:60], r.get("Counter_Name", "?"))
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    print(d)
    for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        print("  %-60s %-14s calls=%6d total=%.5g avg=%.6g" % (k, c, n, v, v / n))
