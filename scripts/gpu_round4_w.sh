#!/bin/bash
# second back-transformation: window traffic issued behind the products (+ one more group in flight with two images ahead)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4w
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back or two_stage_eigenvectors or two_stage_solver or switches" > $O/pytest_q2.log 2>&1; rc=$?; echo "pytest q2 rc=$rc" >> $O/summary.txt; tail -n 8 $O/pytest_q2.log
SCLENS_HIP_Q2_PROF=1 timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 > $O/q2_prof.log 2>&1; echo "prof rc=$?" >> $O/summary.txt
timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 > $O/q2_times.log 2>&1
SCLENS_HIP_Q2_DBG=8 timeout 600 python scripts/q2_variants.py 30016 15008 10 15 > $O/q2_times_earlywin.log 2>&1
timeout 600 python scripts/q2_variants.py 30016 30016 10 14 15 > $O/q2_times_allvec.log 2>&1
grep -h -A 7 "^\[sbr_q2" $O/q2_prof.log | awk 'NR % 16 < 8'
echo ---- times without the profile; grep -h "variant" $O/q2_times.log; echo ---- window traffic behind the DMA; grep -h "variant" $O/q2_times_earlywin.log; echo ---- all vectors; grep -h "variant" $O/q2_times_allvec.log
for v in 10 15; do
  echo "eig variant $v: $(SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee $O/eig_q2.log
cat $O/summary.txt
