#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3dd
mkdir -p $O
ulimit -c 0
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" >> $O/summary.txt
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3dd/bench_cfg4.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["signals"], d["observed"]["search_iters"], d["observed"]["p_"], d["roofline"]["stage_ms"])
PY
cat $O/summary.txt
