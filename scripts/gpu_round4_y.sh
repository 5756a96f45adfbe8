#!/bin/bash
# first back-transformation: Z split in registers by the W1 kernel (no image pass); band reduction: the Z maximum from the kernel that
# writes Z; tests, A/B, eigensolve
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p $O
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; rc=$?; echo "pytest sbr rc=$rc" >> $O/summary.txt; tail -n 6 $O/pytest_sbr.log
sb() { echo "$1: $(env $2 timeout 300 python scripts/perf_sbr.py 30016 2>&1 | tail -n 1)"; }
{ sb default A=1; sb zmax_off SCLENS_HIP_SY2SB_ZMAX=0; sb default A=1; sb zmax_off SCLENS_HIP_SY2SB_ZMAX=0; } 2>&1 | tee $O/sy2sb_zmax.log
for w in 1 2 1 2; do
  echo "eig W1_SPLIT=$w: $(SCLENS_HIP_Q1_W1_SPLIT=$w timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee $O/eig_w1.log
echo "eig all vectors: $(timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep 'rep=1')" | tee -a $O/eig_w1.log
cat $O/summary.txt
