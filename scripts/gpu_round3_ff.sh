#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3ff
mkdir -p $O
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gram_on_split" 2>&1 | tail -n 4
SCLENS_HIP_GRAM_SPLIT=16000 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench_gs.json 2> $O/bench_gs.err; echo "bench rc=$?" >> $O/summary.txt
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3ff/bench_gs.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["signals"], d["observed"]["search_iters"], d["observed"]["p_"], d["roofline"]["stages"]["gram"])
PY
tail -n 3 $O/bench_gs.err
