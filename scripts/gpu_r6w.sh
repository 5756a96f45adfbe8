#!/bin/bash
# round 6, call w: the three float64-arbitrated search evaluations replayed on the final build
O=gpurun_out/r6w; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -q -s -k "arbiter" > $O/pytest_arbiter.log 2>&1; echo "pytest rc $?" >> $O/pytest_arbiter.log; grep "search statistic\|passed\|failed" $O/pytest_arbiter.log | cut -c1-420
