#!/bin/bash
# end of round 4: smoke, the whole GPU suite, the bench line as the driver runs it (+ strict step, CPU baseline), kernel trace of the
# eigensolve and kernel stats of one bench step
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p $O
ulimit -c 0
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?" >> $O/summary.txt; tail -n 4 $O/pytest_gpu_full.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_cfg4_final.json 2> $O/bench_cfg4_final.err; echo "bench cfg4 rc=$?" >> $O/summary.txt
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4z/bench_cfg4_final.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], "strict", d.get("value_strict_fp32"), "differ", d.get("decisions_differ"))
    for x in d["observed"]["decisions_per_step"]:
        print("  step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
        print("     ", x["first_phase_jobs_s"])
    r = d["roofline"]; print("roofline", r["launch_ms"], r["frac"], r["stage_ms"])
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "wall_s_lower_bound")})
except Exception as e:
    print("bench: no result", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_eig -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/eig_under_rocprof.log 2>&1 ); echo "rocprof eig rc=$?" >> $O/summary.txt
DB=$(find /tmp/prof_eig -name "*.db" | head -1); [ -n "$DB" ] && python3 scripts/rocpd_stats.py $DB $O/eig_30016_kernel_stats_final.csv > /dev/null
head -n 16 $O/eig_30016_kernel_stats_final.csv | cut -c1-150
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg4 -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --strict-fp32 off > /root/repo/$O/bench_cfg4_under_rocprof.json 2> /root/repo/$O/bench_cfg4_under_rocprof.err ); echo "rocprof bench rc=$?" >> $O/summary.txt
DB=$(find /tmp/prof_cfg4 -name "*.db" | head -1); [ -n "$DB" ] && python3 scripts/rocpd_stats.py $DB $O/cfg4_kernel_stats_final.csv > /dev/null
head -n 24 $O/cfg4_kernel_stats_final.csv | cut -c1-150
cat $O/summary.txt
