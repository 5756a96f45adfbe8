#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
SCLENS_HIP_CHEFSI_TAIL_GAP=1e9 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size_nogap.log 2>&1; echo "bench-size (no tail gap) rc=$?" >> $O/summary.txt
SCLENS_HIP_CHEFSI_TAIL_GAP=0.1 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size_gap01.log 2>&1; echo "bench-size (gap 0.1) rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_atlas.py tests/test_gpu_pattern.py -x -q -m gpu --durations=6 > $O/pytest_atlas.log 2>&1; echo "atlas tests rc=$?" >> $O/summary.txt
grep "bench-size parity" $O/*.log; tail -n 12 $O/pytest_atlas.log; cat $O/summary.txt
