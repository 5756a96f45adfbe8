#!/bin/bash
# HBM traffic of ONE two-stage eigensolve of order 30 016 with 15 008 vectors on the shipped build: separate FETCH_SIZE / WRITE_SIZE
# passes (MI355X_MICROARCH.md: they do not fit one pass), the program directly after `--`, counters collected only for the kernels
# that move the bytes (--kernel-include-regex): round 3's unfiltered passes serialised ~60 000 dispatches under the profiler and did
# not finish inside 10 minutes (round 4's first call: rc 124 for both) -- they were slow, not hung.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
export LOW_HALF=1 TWO_STAGE=1 REPS=1
REGEX='sbr_q2_apply|gemm_split_kernel|gemm_nt_big|sbr_chase_mb|tri_stein|gemm_kernel|split_image|sbr_q2_build'
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  t0=$(date +%s)
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX" --output-format csv -d /tmp/pmc_$c -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_$c.log 2>&1
  echo "pmc $c rc=$? $(( $(date +%s) - t0 )) s" >> /root/repo/$O/summary.txt
done
cd /root/repo
unset LOW_HALF TWO_STAGE REPS
python3 - <<'PY' > gpurun_out/r4c/pmc_eig_summary.txt 2>&1
import collections, csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_{c}/*/*counter_collection.csv") + glob.glob(f"/tmp/pmc_{c}/*counter_collection.csv")
    if not fs:
        print("no counter file for", c)
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    tot[c] = agg
    print(c, "(counter units: KB)")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-60s calls=%6d total=%.5g avg=%.6g" % (k, n, v, v / n))
    print("  SUM over the profiled kernels: %.6g KB" % sum(v for _, v in agg.values()))
PY
cat $O/pmc_eig_summary.txt
# kernel trace of one band reduction + chase on this build, by octile of the panel index (what is left of the dense -> band stage)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_sbr -- python3 /root/repo/scripts/perf_sbr.py 30016 > /root/repo/$O/perf_sbr.log 2>&1
cd /root/repo
TR=$(find /tmp/trace_sbr -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python3 scripts/trace_sy2sb.py $TR $O/sy2sb_trace_summary_30016.json > $O/trace_sy2sb.log 2>&1
tail -n 3 $O/perf_sbr.log
cat $O/summary.txt
