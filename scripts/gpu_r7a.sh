#!/bin/bash
# round 6, call 7a: the z_data_3869 golden fixture (BASELINE config 1's stand-in for the missing Z8eq) on the device path
O=gpurun_out/r7a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_golden.py -q -s -k "3869" > $O/pytest_3869.log 2>&1; echo "pytest rc $?" >> $O/pytest_3869.log; grep -v "^$" $O/pytest_3869.log | cut -c1-400 | tail -12
