#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
ulimit -c 0
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" >> $O/summary.txt
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3v/bench_cfg4.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"])
PY
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_gpu.log
cat $O/summary.txt
