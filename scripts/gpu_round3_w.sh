#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3w
mkdir -p $O
ulimit -c 0
for st in 2 3; do
  timeout 900 python bench.py --steps 1 --warmup 1 --streams $st --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_streams$st.json 2> $O/bench_streams$st.err; echo "bench streams=$st rc=$?" >> $O/summary.txt
  python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3w/bench_streams$st.json').read().strip().splitlines()[-1])
print("streams=$st", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"])
PY
done
cat $O/summary.txt
