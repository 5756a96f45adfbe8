#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
ulimit -c 0
# reproduce the atlas dry-run fault at a smaller size with serialised kernels (the failing call then returns an error code)
ATLAS_M=8192 AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 900 python scripts/atlas_dry_run.py 160000 8 $O/atlas_small.json > $O/atlas_small_stdout.log 2> $O/atlas_small_stderr.log; echo "atlas small rc=$?" >> $O/summary.txt
tail -n 20 $O/atlas_small_stderr.log
timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -s -m gpu > $O/pytest_bench_size.log 2>&1; echo "bench-size rc=$?" >> $O/summary.txt
grep "bench-size parity\|passed\|failed" $O/pytest_bench_size.log
timeout 2400 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_bench_size.py > $O/pytest_gpu_full.log 2>&1; echo "full suite (val_csr build) rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_gpu_full.log
cat $O/summary.txt
