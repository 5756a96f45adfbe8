#!/bin/bash
# round 6, call i: fp32 second back-transformation with passes of 12 / 16 blocks
O=gpurun_out/r6i; mkdir -p $O
export TMPDIR=/tmp
for b in 12 16; do
  SCLENS_HIP_OPTIONS=precision=0,q2_fp32_blocks=$b LOW_HALF=1 PRINT_HASH=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig_fp32_q2_blocks$b.log 2>&1; tail -4 $O/perf_eig_fp32_q2_blocks$b.log | head -2
done
