#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -m gpu -x -q > $O/pytest_k.log 2>&1; echo "kernels+sbr rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_k.log
for f in 0 1; do
  SCLENS_HIP_BISECT_DIV=$f LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_div$f.log; echo "bisect_div=$f $(cat $O/eig_div$f.log)"
done
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench rc=$?" >> $O/summary.txt
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3y/bench_cfg4.json').read().strip().splitlines()[-1])
print(d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["signals"], d["observed"]["search_iters"], d["observed"]["p_"])
PY
cat $O/summary.txt
