"""Summarise a rocprofv3 --kernel-trace CSV of scripts/perf_sbr.py: per kernel name totals, and for the LAST sy2sb call the
duration of every launch class as a function of the panel index (binned), plus stream-idle gaps. Usage: trace_sy2sb.py trace.csv out.json"""
import csv
import json
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Stream_Id", r.get("Queue_Id", 0)) or 0),
                     int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0)))
rows.sort()
tot = defaultdict(lambda: [0, 0])
for s, e, k, q, g in rows:
    k = k.split("(")[0]
    tot[k][0] += 1
    tot[k][1] += e - s
out = {"totals_ms": {k: [c, round(t / 1e6, 3)] for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]}}
# the last run of sy2sb: from the last sbr_pad/first gram64<true> ... take launches between the last two sbr_chase kernels
chase = [i for i, r in enumerate(rows) if "sbr_chase" in r[2]]
if len(chase) >= 2:
    seg = rows[chase[-2] + 1: chase[-1]]
    t0, t1 = seg[0][0], seg[-1][1]
    out["segment_ms"] = round((t1 - t0) / 1e6, 3)
    # classify
    cls = defaultdict(list)
    for s, e, k, q, g in seg:
        name = k.split("(")[0].replace("void scl::", "").replace("scl::", "")
        cls[name].append((s - t0, e - s, g))
    summ = {}
    for name, lst in cls.items():
        n = len(lst)
        bins = []
        for b in range(8):
            part = lst[b * n // 8: (b + 1) * n // 8]
            if part:
                bins.append(round(sum(d for _, d, _ in part) / len(part) / 1e3, 1))
        summ[name] = {"calls": n, "total_ms": round(sum(d for _, d, _ in lst) / 1e6, 2), "avg_us_by_octile": bins}
    out["last_sy2sb"] = summ
    # busy time of the union of all kernels (any stream) inside the segment
    ev = sorted((s, e) for s, e, *_ in seg)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    out["last_sy2sb_union_busy_ms"] = round(busy / 1e6, 3)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:3000])
