"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) rocpd SQLite result: name, calls, total ms, average us, share.
Usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc"))
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for nm, c, t, mn, mx in rows:
    short = nm.replace("(anonymous namespace)::", "").split("(")[0][-70:]
    lines.append(f'"{short}",{c},{t},{t / c:.1f},{100.0 * t / tot:.2f},{mn},{mx}')
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
