#!/bin/bash
# defaults at the end of round 4: Q2 variant 15 (window traffic behind the DMA, relaxed wait in the second group), null matrix in
# page-locked memory; tests of both, times, smoke, two bench steps
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4x
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_pattern.py tests/test_gpu_sbr.py -m gpu -x -q -k "page_locked or second_back or two_stage_eigenvectors or two_stage_solver or switches" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/summary.txt; tail -n 8 $O/pytest.log
timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 > $O/q2_times.log 2>&1
timeout 600 python scripts/q2_variants.py 30016 30016 10 14 15 > $O/q2_times_allvec.log 2>&1
echo ---- m = n / 2; grep -h "variant" $O/q2_times.log; echo ---- all vectors; grep -h "variant" $O/q2_times_allvec.log
echo "eig: $(timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')" | tee $O/eig.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.log
for pd in 1 0; do
SCLENS_PINNED_DRAWS=$pd timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_pinned$pd.json 2> $O/bench_pinned$pd.err
python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_pinned$pd.json").read().strip().splitlines()[-1])
    for x in d["observed"]["decisions_per_step"]:
        print("pinned=$pd step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
    print("   first phase jobs:", d["observed"]["first_phase_jobs_s_last_step"])
except Exception as e:
    print("bench: no result", e)
PY
done
cat $O/summary.txt
