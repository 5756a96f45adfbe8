#!/bin/bash
# What bounds the image-fed second back-transformation (variants 8 / 9: 300 ms against 385 for variant 7, not the 2x its instruction
# count suggests)? Floors with the products or the DMA switched off, kernel times of the build, and an A/B of whole sclens() calls.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
ulimit -c 0
for dbg in 0 1 2 3; do
  SCLENS_HIP_Q2_DBG=$dbg timeout 600 python scripts/q2_variants.py 30016 15008 9 2>&1 | grep "variant 9:" > $O/q2_dbg$dbg.log; echo "dbg $dbg: $(cat $O/q2_dbg$dbg.log)"
done
SCLENS_HIP_Q2_DBG=1 timeout 600 python scripts/q2_variants.py 30016 15008 8 2>&1 | grep "variant 8:" > $O/q2_v8_dbg1.log; echo "v8 dbg 1: $(cat $O/q2_v8_dbg1.log)"
export LOW_HALF=1 TWO_STAGE=1
cd /tmp
SCLENS_HIP_Q2_VARIANT=9 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eig -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/prof_eig.log 2>&1
cd /root/repo
CSV=$(find /tmp/prof_eig -name "*kernel_stats.csv" | head -1)
[ -n "$CSV" ] && cp $CSV $O/eig_kernel_stats_v9.csv && head -n 16 $O/eig_kernel_stats_v9.csv
SCLENS_HIP_Q2_VARIANT=9 timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_v9.log; echo "all vectors v9: $(cat $O/eig_all_v9.log)"
unset LOW_HALF TWO_STAGE
for v in 7 9 7 9; do
  SCLENS_HIP_Q2_VARIANT=$v timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_v$v.json 2> $O/bench_v$v.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_v$v.json").read().strip().splitlines()[-1])
    print("variant $v:", d["sclens_wall_s"], [q["wall_s"] for q in d["observed"]["decisions_per_step"]], d["observed"]["phase_s_rank0_last_step"])
except Exception as e:
    print("variant $v: no result", e)
PY
done
