#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3ee
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sbr.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -n 4
for q in 0 1024; do
  SCLENS_HIP_Q1_SPLIT=$q LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_q1split$q.log; echo "q1split=$q $(cat $O/eig_q1split$q.log)"
done
LOW_HALF=1 TWO_STAGE=1 timeout 300 python scripts/perf_eig.py 4000 8000 4000 2>&1 | tail -n 3
SCLENS_HIP_Q1_SPLIT=0 SCLENS_HIP_SY2SB_SPLIT=0 SCLENS_HIP_Q2_VARIANT=3 LOW_HALF=1 TWO_STAGE=1 timeout 300 python scripts/perf_eig.py 4000 8000 4000 2>&1 | tail -n 3
