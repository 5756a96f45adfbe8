#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3s
mkdir -p $O
ulimit -c 0
SCLENS_HIP_SY2SB_DELAY_MIN=300 timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; echo "sbr (delay from 300) rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_sbr.log
for d in nodelay 300 12288 16384 18432 20480 24576; do
  if [ $d = nodelay ]; then export SCLENS_HIP_SY2SB_NO_DELAY=1; else unset SCLENS_HIP_SY2SB_NO_DELAY; export SCLENS_HIP_SY2SB_DELAY_MIN=$d; fi
  LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_delay_$d.log; echo "delay_min=$d $(cat $O/eig_delay_$d.log)"
done
cat $O/summary.txt
