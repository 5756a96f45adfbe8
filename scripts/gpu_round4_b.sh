#!/bin/bash
# Second GPU call of round 4: retune the thresholds that were set while the trailing updates still ran on the fp32 pipes.
# One eigendecomposition of order 30 016 with 15 008 vectors per setting (scripts/perf_eig.py; ~20 s each incl. start-up).
cd /root/repo
export TMPDIR=/tmp LOW_HALF=1 TWO_STAGE=1
O=gpurun_out/r4b
mkdir -p $O
run() {  # name, then VAR=value pairs
  local name=$1; shift
  env "$@" timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_$name.log
  echo "$name $(cat $O/eig_$name.log)"
}
run base
for d in 8192 12288 24576; do run delay$d SCLENS_HIP_SY2SB_DELAY_MIN=$d; done
run nodelay SCLENS_HIP_SY2SB_NO_DELAY=1
for s in 2048 8192; do run split$s SCLENS_HIP_SY2SB_SPLIT=$s; done
run scales2 SCLENS_HIP_SY2SB_SPLIT_SCALES=2
for q in 512 2048; do run q1split$q SCLENS_HIP_Q1_SPLIT=$q; done
for g in 192 224; do run chase$g SCLENS_HIP_CHASE_WGS=$g; done
run steinpf32 SCLENS_HIP_STEIN_PF=32
