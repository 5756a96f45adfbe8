"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, calls / total / average of the counter."""
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
    cname = "?"
    with open(files[0]) as fh:
        for r in csv.DictReader(fh):
            cname = r.get("Counter_Name", cname)
            k = r["Kernel_Name"][:44]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    print(d, cname)
    for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("  %-44s calls=%6d total=%.5g avg=%.6g" % (k, c, v, v / c))
