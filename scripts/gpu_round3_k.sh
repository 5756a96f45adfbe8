#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
SCLENS_HIP_CHEFSI_TAIL_GAP=1e9 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size_nogap.log 2>&1; echo "bench-size (no tail gap) rc=$?" >> $O/summary.txt
SCLENS_HIP_CHEFSI_TAIL_GAP=0.1 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size_gap01.log 2>&1; echo "bench-size (gap 0.1) rc=$?" >> $O/summary.txt
SCLENS_HIP_CHEFSI_TAIL_GAP=0.05 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size_gap005.log 2>&1; echo "bench-size (gap 0.05) rc=$?" >> $O/summary.txt
grep "bench-size parity" $O/*.log
timeout 1500 python scripts/atlas_dry_run.py 1000000 8 $O/atlas_slab_dry_run.json > $O/atlas_stdout.log 2> $O/atlas_stderr.log; echo "atlas dry run rc=$?" >> $O/summary.txt
tail -n 25 $O/atlas_stderr.log
cat $O/summary.txt
