#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_preprocess.py -x -q -m gpu > $O/pytest_pre.log 2>&1; echo "preprocess tests rc=$?" >> $O/summary.txt
SCLENS_HIP_CHEFSI_TAIL_GAP=1e9 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -m gpu > $O/pytest_bench_size_nogap.log 2>&1; echo "bench-size (no tail gap) rc=$?" >> $O/summary.txt
SCLENS_HIP_CHEFSI_TAIL_GAP=0.05 timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -m gpu > $O/pytest_bench_size_gap005.log 2>&1; echo "bench-size (gap 0.05) rc=$?" >> $O/summary.txt
for g in 0.2 0.1; do
SCLENS_HIP_CHEFSI_TAIL_GAP=$g timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_gap$g.json 2> /dev/null
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3i/bench_gap$g.json').read().strip().splitlines()[-1])
print("gap $g", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"])
PY
done
tail -5 $O/pytest*.log; cat $O/summary.txt
