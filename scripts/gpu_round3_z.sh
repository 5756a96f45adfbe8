#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3z
mkdir -p $O
for pf in 4 16 32; do
  SCLENS_HIP_STEIN_PF=$pf LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_pf$pf.log; echo "stein_pf=$pf $(cat $O/eig_pf$pf.log)"
done
SCLENS_HIP_STEIN_PF=32 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "eigh" 2>&1 | tail -n 2
