#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3aa
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sclens.py tests/test_gpu_sbr.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "sclens+sbr rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_a.log
for f in 1 0; do
  SCLENS_HIP_DENSE_FUSED=$f timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench_fused$f.json 2> $O/bench_fused$f.err; echo "bench fused=$f rc=$?" >> $O/summary.txt
  python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3aa/bench_fused$f.json').read().strip().splitlines()[-1])
print("fused=$f", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["roofline"]["stages"]["normalise"], d["roofline"]["stages"].get("normalise_search_step"))
PY
done
cat $O/summary.txt
