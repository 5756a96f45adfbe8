#!/bin/bash
# second back-transformation from 16 KB images (variants 14 / 15: one copy of the reflectors + T, transposing LDS reads): tests,
# timing against the default, then a bench step with the first phase's job times
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back" > $O/pytest_q2.log 2>&1; rc=$?; echo "pytest q2 rc=$rc" >> $O/summary.txt; tail -n 15 $O/pytest_q2.log
for v in 10 14 15; do
  echo "variant $v: $(SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee $O/q2_variants.log
for v in 10 14; do
  echo "all vectors, variant $v: $(SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep 'rep=1')"
  echo "no products, variant $v: $(SCLENS_HIP_Q2_DBG=1 SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee -a $O/q2_variants.log
V=10; [ $rc -eq 0 ] && V=14
SCLENS_HIP_Q2_VARIANT=$V timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4t/bench.json").read().strip().splitlines()[-1])
    for x in d["observed"]["decisions_per_step"]:
        print("step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
    print("first phase jobs:", d["observed"]["first_phase_jobs_s_last_step"])
except Exception as e:
    print("bench: no result", e)
PY
cat $O/summary.txt
