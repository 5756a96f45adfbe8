#!/bin/bash
# round 6, call 7h (last): after the positivity-rule change -- the float64 arbiter test of the search statistic at 100 000 x 30 000 (the
# one test that reads r at that size), then the accelerated-against-plain test at order 30 000, as far as the budget reaches
O=gpurun_out/r7h; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_bench_size.py -x -q -s -k "arbiter" > $O/pytest_arbiter.log 2>&1; echo "pytest rc $?" >> $O/pytest_arbiter.log; grep -v "^$" $O/pytest_arbiter.log | cut -c1-300 | tail -8
timeout 170 python -m pytest tests/test_gpu_bench_size.py -x -q -k "accelerated" > $O/pytest_accel.log 2>&1; echo "pytest rc $?" >> $O/pytest_accel.log; tail -2 $O/pytest_accel.log
