#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; echo "sbr+golden rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_sbr.log
for mb in 0 1; do
  SCLENS_HIP_CHASE_MB=$mb LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=" > $O/eig_mb$mb.log; echo "eig mb=$mb rc=$?" >> $O/summary.txt
  cat $O/eig_mb$mb.log
done
for g in 64 128 200; do
  SCLENS_HIP_CHASE_WGS=$g SCLENS_HIP_CHASE_MB=1 LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_mb1_g$g.log
  echo "G=$g"; cat $O/eig_mb1_g$g.log
done
cat $O/summary.txt
