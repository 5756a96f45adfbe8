#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu_full.log 2>&1; echo "full gpu suite rc=$?" >> $O/summary.txt
tail -25 $O/pytest_gpu_full.log
timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -c 3000 $O/bench_cfg4.json
cat $O/summary.txt
