#!/bin/bash
# round 6, call h: the fp32 second back-transformation with passes of 8 blocks against 4 (context option q2_fp32_blocks); the auto-choice test
O=gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
for b in 4 8; do
  SCLENS_HIP_OPTIONS=precision=0,q2_fp32_blocks=$b LOW_HALF=1 PRINT_HASH=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig_fp32_q2_blocks$b.log 2>&1; tail -5 $O/perf_eig_fp32_q2_blocks$b.log
done
timeout 900 python -m pytest tests/test_gpu_gram_sparse.py -x -q -k "chosen" > $O/pytest_auto.log 2>&1; tail -3 $O/pytest_auto.log
timeout 900 python -m pytest tests/test_gpu_sbr.py -x -q > $O/pytest_sbr.log 2>&1; tail -3 $O/pytest_sbr.log
