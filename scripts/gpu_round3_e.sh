#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
for d in 0 1 2 3 4 7; do
  SCLENS_HIP_GEMM_DBG=$d python scripts/perf_update.py 28160 128 4 2>&1 | tail -1 >> $O/update_k128.log
done
for d in 0 1 3 7; do
  SCLENS_HIP_GEMM_DBG=$d python scripts/perf_update.py 28160 256 4 2>&1 | tail -1 >> $O/update_k256.log
  SCLENS_HIP_GEMM_DBG=$d python scripts/perf_update.py 28160 512 4 2>&1 | tail -1 >> $O/update_k512.log
done
SCLENS_HIP_NO_ACC_INIT=1 python scripts/perf_update.py 28160 128 4 2>&1 | tail -1 >> $O/update_k128.log
LOWER=0 python scripts/perf_update.py 15104 256 4 2>&1 | tail -1 >> $O/update_full.log
LOWER=0 SCLENS_HIP_GEMM_DBG=2 python scripts/perf_update.py 15104 256 4 2>&1 | tail -1 >> $O/update_full.log
LOWER=0 SCLENS_HIP_GEMM_DBG=6 python scripts/perf_update.py 15104 256 4 2>&1 | tail -1 >> $O/update_full.log
cat $O/*.log
