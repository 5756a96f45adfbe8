#!/bin/bash
# per-evaluation times of the sparsity search per worker after the default and the `three` first-phase schedules; smoke
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4p
mkdir -p $O
ulimit -c 0
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 2 $O/smoke.log
for fp in default three; do
  SCLENS_FIRST_PHASE=$fp timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$fp.json 2> $O/bench_$fp.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$fp.json").read().strip().splitlines()[-1])
    print("first phase $fp:", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], "HBM in use", d["observed"]["hbm_in_use_GB_after_timed_steps"])
    for w in (0, 1):
        print("   worker", w, [q[2] for q in d["observed"]["search_job_s_last_step"] if q[0] == w])
except Exception as e:
    print("first phase $fp: no result", e)
PY
done
cat $O/summary.txt
