#!/bin/bash
# round 5, eighth GPU call: Q2 with passes of 12 blocks (variant 19, one workgroup per CU) against 16; per-sweep trace of the partial
# eigensolver on the ensemble of cfg4 (how many sweeps, which degrees, residuals after the first sweep)
set -x
O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/q2_variants.py 30016 15008 16 19 > $O/q2_half.log 2>&1; tail -6 $O/q2_half.log
SCLENS_HIP_OPTIONS="debug=1" SCLENS_BENCH_DETAIL=$O/detail_debug.json timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_debug.json 2> $O/bench_debug.err
grep -c chefsi $O/bench_debug.err; grep "chefsi" $O/bench_debug.err | cut -c1-400 > $O/chefsi_trace.log; head -30 $O/chefsi_trace.log
du -sh $O
