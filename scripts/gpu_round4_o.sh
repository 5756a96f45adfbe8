#!/bin/bash
# first phase with three decompositions at once (SCLENS_FIRST_PHASE=three): identity test, A/B; kernel stats of one bench step (rocpd)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4o
mkdir -p $O
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_sclens.py -m gpu -x -q -k "chained_first_phase" > $O/pytest_fp.log 2>&1; echo "first-phase identity rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_fp.log
for fp in default three default three; do
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
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg4 -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --strict-fp32 off > /root/repo/$O/bench_cfg4_under_rocprof.json 2> /root/repo/$O/bench_cfg4_under_rocprof.err
echo "rocprof rc=$?" >> /root/repo/$O/summary.txt
cd /root/repo
DB=$(find /tmp/prof_cfg4 -name "*.db" | head -1)
CSV=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1)
if [ -n "$CSV" ]; then cp $CSV $O/cfg4_kernel_stats.csv; elif [ -n "$DB" ]; then python3 scripts/rocpd_stats.py $DB $O/cfg4_kernel_stats.csv > /dev/null; fi
head -n 14 $O/cfg4_kernel_stats.csv | cut -c1-170
cat $O/summary.txt
