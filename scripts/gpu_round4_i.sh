#!/bin/bash
# first-phase schedules (default: data | null, then signal vectors | binarised; chain: worker null -> binarised; chain2: worker
# binarised -> null) with a warm-up step each, then the whole GPU suite on this build
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
ulimit -c 0
for fp in default chain2 chain default chain2; do
  SCLENS_FIRST_PHASE=$fp timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_fp.json 2> $O/bench_fp.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_fp.json").read().strip().splitlines()[-1])
    print("first phase $fp:", d["sclens_wall_s"], [q["wall_s"] for q in d["observed"]["decisions_per_step"]], d["observed"]["phase_s_rank0_last_step"], d["observed"]["search_iters"], d["observed"]["p_"])
except Exception as e:
    print("first phase $fp: no result", e)
PY
done
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 22 $O/pytest_gpu_full.log
cat $O/summary.txt
