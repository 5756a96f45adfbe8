#!/bin/bash
# last call of the round on the final commit: smoke, the whole GPU suite, the bench line
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4q
mkdir -p $O
ulimit -c 0
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 12 $O/pytest_gpu_full.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_cfg4_final.json 2> $O/bench_cfg4_final.err; echo "bench cfg4 rc=$?" >> $O/summary.txt
tail -c 600 $O/bench_cfg4_final.json; echo
cat $O/summary.txt
