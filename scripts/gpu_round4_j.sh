#!/bin/bash
# loader-wave variants (12 / 13) of the image-fed second back-transformation: tests, times
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back_transformation" > $O/pytest_q2.log 2>&1; echo "q2 tests rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_q2.log
timeout 900 python scripts/q2_variants.py 30016 15008 10 12 13 > $O/q2_variants_15008.log 2>&1
tail -n 8 $O/q2_variants_15008.log
for dbg in 1 2; do
  SCLENS_HIP_Q2_DBG=$dbg timeout 600 python scripts/q2_variants.py 30016 15008 12 2>&1 | grep "variant 12:" > $O/q2_v12_dbg$dbg.log; echo "v12 dbg $dbg: $(cat $O/q2_v12_dbg$dbg.log)"
done
export LOW_HALF=1 TWO_STAGE=1
for v in 12 13; do
  SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_v$v.log; echo "v$v: $(cat $O/eig_v$v.log)"
done
cat $O/summary.txt
