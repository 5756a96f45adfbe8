#!/bin/bash
# round 6, call k: tunables of the fp32 eigensolve (rank-256 updates from a lower order, panels per block reflector of Q1)
O=gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
for opts in "precision=0" "precision=0,sy2sb_delay_min=4096" "precision=0,sy2sb_delay_min=64" "precision=0,q1_group=4" "precision=0,q1_group=8" "precision=0,sy2sb_lookahead=0"; do
  echo "== $opts"
  SCLENS_HIP_OPTIONS=$opts LOW_HALF=1 PRINT_HASH=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1\|crc32" | cut -c1-260
done > $O/fp32_eig_tunables.log 2>&1
cat $O/fp32_eig_tunables.log
