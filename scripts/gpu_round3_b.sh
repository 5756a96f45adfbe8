#!/bin/bash
# round-3 GPU check B: Q2 variants and the GEMM phase stagger at n = 30 016
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> $O/summary.txt
for q in 0 1 3; do
  SCLENS_HIP_Q2_VARIANT=$q SCLENS_HIP_GEMM_STAGGER_PCT=0 LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/q2v${q}_stag0.log
done
for p in 50 100 150; do
  SCLENS_HIP_Q2_VARIANT=3 SCLENS_HIP_GEMM_STAGGER_PCT=$p LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/q2v3_stag${p}.log
done
SCLENS_HIP_Q2_VARIANT=0 timeout 900 python -m pytest tests/test_gpu_sbr.py -x -q > $O/pytest_sbr_q2v0.log 2>&1; echo "sbr q2v0 rc=$?" >> $O/summary.txt
for f in $O/q2*.log; do echo $f; cat $f; done
cat $O/summary.txt
