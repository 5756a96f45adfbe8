#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3q
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; echo "sbr+golden rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_sbr.log
for g in 0 64; do
  SCLENS_HIP_CHASE_WGS=$([ $g = 0 ] && echo 100000 || echo $g) SCLENS_HIP_CHASE_PROF=1 LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 > $O/prof_g$g.log 2>&1
  grep -A9 "sbr_chase_mb" $O/prof_g$g.log | tail -n 10; grep "rep=1" $O/prof_g$g.log
done
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_gpu.log
cat $O/summary.txt
