#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3u
mkdir -p $O
ulimit -c 0
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" >> $O/summary.txt
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3u/bench_cfg4.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"])
PY
timeout 900 python scripts/first_phase_cfg4.py > $O/first_phase.log 2>&1; echo "first phase rc=$?" >> $O/summary.txt
cat $O/first_phase.log | tail -n 40
cat $O/summary.txt
