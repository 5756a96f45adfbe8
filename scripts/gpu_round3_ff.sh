#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3ff
mkdir -p $O
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gram_on_split" 2>&1 | tail -n 4
SCLENS_HIP_GRAM_SPLIT=64 SCLENS_TEST_SKIP_FULL=1 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_size.py > $O/pytest_gramsplit_forced.log 2>&1; echo "forced suite rc=$?" >> $O/summary.txt
tail -n 12 $O/pytest_gramsplit_forced.log
