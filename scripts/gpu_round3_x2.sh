#!/bin/bash
# final measurements of round 3 (second pass, after the split-fp16 products): bench line, kernel stats of the same command, PMC passes, cfg2
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3x2
mkdir -p $O
ulimit -c 0
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_cfg4_final.json 2> $O/bench_cfg4_final.err; echo "bench cfg4 rc=$?" >> $O/summary.txt
tail -c 600 $O/bench_cfg4_final.json; echo
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg4 -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --strict-fp32 off > /root/repo/$O/bench_cfg4_under_rocprof.json 2> /root/repo/$O/bench_cfg4_under_rocprof.err
cd /root/repo
echo "rocprof rc=$?" >> $O/summary.txt
DB=$(find /tmp/prof_cfg4 -name "*.db" | head -1)
CSV=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1)
if [ -n "$CSV" ]; then cp $CSV $O/cfg4_kernel_stats.csv; elif [ -n "$DB" ]; then python3 scripts/rocpd_stats.py $DB $O/cfg4_kernel_stats.csv > /dev/null; fi
head -n 12 $O/cfg4_kernel_stats.csv
export LOW_HALF=1 TWO_STAGE=1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_eig_f -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_eig_w -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_write.log 2>&1
cd /root/repo
grep "rep=1" $O/pmc_fetch.log | tail -1
unset LOW_HALF TWO_STAGE
python3 scripts/pmc_summary.py /tmp/pmc_eig_f /tmp/pmc_eig_w > $O/pmc_eig_summary.txt 2>&1
python3 - <<'PY' >> gpurun_out/r3x2/pmc_eig_summary.txt 2>&1
import csv, glob
for d, nm in (("/tmp/pmc_eig_f", "FETCH_SIZE"), ("/tmp/pmc_eig_w", "WRITE_SIZE")):
    fs = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    tot = 0.0; n = 0
    for r in csv.DictReader(open(fs[0])):
        tot += float(r["Counter_Value"]); n += 1
    print(f"{nm}: {n} dispatches, sum {tot:.6g} (counter units) over the whole run = 2 eigendecompositions + 2 Gram products")
PY
tail -n 22 $O/pmc_eig_summary.txt
timeout 600 python bench.py --config cfg2 --steps 3 --warmup 1 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?" >> $O/summary.txt
tail -c 300 $O/bench_cfg2.json; echo
cat $O/summary.txt
