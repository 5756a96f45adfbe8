#!/bin/bash
# second back-transformation: bare barrier per group (the counted waits become effective), hand-scheduled group of the 16 KB image
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4v
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back or two_stage_eigenvectors or two_stage_solver or switches" > $O/pytest_q2.log 2>&1; rc=$?; echo "pytest q2 rc=$rc" >> $O/summary.txt; tail -n 8 $O/pytest_q2.log
SCLENS_HIP_Q2_PROF=1 timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 9 > $O/q2_prof.log 2>&1; echo "prof rc=$?" >> $O/summary.txt
SCLENS_HIP_Q2_PROF=1 SCLENS_HIP_Q2_DBG=4 timeout 600 python scripts/q2_variants.py 30016 15008 10 15 > $O/q2_prof_syncthreads.log 2>&1; echo "prof (syncthreads) rc=$?" >> $O/summary.txt
timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 > $O/q2_times.log 2>&1
timeout 600 python scripts/q2_variants.py 30016 30016 10 11 14 15 > $O/q2_times_allvec.log 2>&1
grep -h -A 7 "^\[sbr_q2" $O/q2_prof.log | awk 'NR % 16 < 8' ; grep -h "variant" $O/q2_prof.log
echo ---- per-group syncthreads; grep -h -A 7 "^\[sbr_q2" $O/q2_prof_syncthreads.log | awk 'NR % 16 < 8'; grep -h "variant" $O/q2_prof_syncthreads.log
echo ---- times without the profile; grep -h "variant" $O/q2_times.log; echo ---- all vectors; grep -h "variant" $O/q2_times_allvec.log
V=10; [ $rc -eq 0 ] && V=15
SCLENS_HIP_Q2_VARIANT=$V timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4v/bench.json").read().strip().splitlines()[-1])
    for x in d["observed"]["decisions_per_step"]:
        print("step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
    print("first phase jobs:", d["observed"]["first_phase_jobs_s_last_step"])
except Exception as e:
    print("bench: no result", e)
PY
cat $O/summary.txt
