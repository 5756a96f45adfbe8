#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> $O/summary.txt
SCLENS_HIP_Q1G=8 timeout 900 python -m pytest tests/test_gpu_sbr.py -x -q > $O/pytest_sbr_q1g8.log 2>&1; echo "sbr q1g8 rc=$?" >> $O/summary.txt
for g in 4 8; do
  SCLENS_HIP_Q1G=$g LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/perf_q1g$g.log
done
SCLENS_HIP_Q1G=8 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/perf_q1g8_allvec.log
cat $O/perf*.log $O/summary.txt; tail -2 $O/pytest*.log
