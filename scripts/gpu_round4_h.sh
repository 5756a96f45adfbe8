#!/bin/bash
# Q2 image kernel with counted waits (window traffic behind the DMA): tests, times, floors
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back_transformation or eigh or same_bits or two_stage" > $O/pytest_q2.log 2>&1; echo "q2 tests rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_q2.log
timeout 900 python scripts/q2_variants.py 30016 15008 7 9 10 11 > $O/q2_variants_15008.log 2>&1
tail -n 10 $O/q2_variants_15008.log
for dbg in 1 2 3; do
  SCLENS_HIP_Q2_DBG=$dbg timeout 600 python scripts/q2_variants.py 30016 15008 10 2>&1 | grep "variant 10:" > $O/q2_v10_dbg$dbg.log; echo "v10 dbg $dbg: $(cat $O/q2_v10_dbg$dbg.log)"
done
SCLENS_HIP_Q2_DBG=1 timeout 600 python scripts/q2_variants.py 30016 15008 11 2>&1 | grep "variant 11:" > $O/q2_v11_dbg1.log; echo "v11 dbg 1: $(cat $O/q2_v11_dbg1.log)"
export LOW_HALF=1 TWO_STAGE=1
for v in 10 11; do
  SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_v$v.log; echo "all vectors v$v: $(cat $O/eig_all_v$v.log)"
done
cat $O/summary.txt
