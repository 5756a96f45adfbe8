#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" >> $O/summary.txt
tail -c 3000 $O/bench_cfg4.json
timeout 1500 python scripts/atlas_dry_run.py 1000000 8 $O/atlas_slab_dry_run.json > $O/atlas_stdout.log 2> $O/atlas_stderr.log; echo "atlas dry run rc=$?" >> $O/summary.txt
tail -n 8 $O/atlas_stderr.log
cat $O/summary.txt
