#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_bench_size.py tests/test_gpu_z8eq.py tests/test_gpu_sclens.py -x -q -m gpu --durations=8 > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/summary.txt
tail -20 $O/pytest_new.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4_tailgap.json 2> $O/bench_cfg4.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r3h/bench_cfg4_tailgap.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"])
PY
cat $O/summary.txt
