#!/bin/bash
# round 6, call 7c: the cfg3 (50 000 x 30 000) float64 comparison with its printed distances, and the skip reasons of the full suite
O=gpurun_out/r7c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -s -rs -k "cfg3_spectrum" > $O/pytest_cfg3.log 2>&1; echo "pytest rc $?" >> $O/pytest_cfg3.log; grep -v "^$" $O/pytest_cfg3.log | cut -c1-400 | tail -12
