#!/bin/bash
# Q1 group data prepared ahead (sbr_q1_prepare): identity test, stage times, A/B of whole calls; then the filtered PMC passes and the
# band-reduction trace of the same build (scripts/gpu_round4_c.sh)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "prepared_ahead or same_bits or two_stage" > $O/pytest_q1prep.log 2>&1; echo "q1 prep tests rc=$?" >> $O/summary.txt
tail -n 4 $O/pytest_q1prep.log
export LOW_HALF=1 TWO_STAGE=1
for prep in 1 0; do
  SCLENS_HIP_Q1_PREP=$prep timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_prep$prep.log; echo "prep $prep: $(cat $O/eig_prep$prep.log)"
done
SCLENS_HIP_Q1_PREP=1 timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_prep1.log; echo "all vectors prep 1: $(cat $O/eig_all_prep1.log)"
unset LOW_HALF TWO_STAGE
for prep in 0 1 0 1; do
  SCLENS_HIP_Q1_PREP=$prep timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_prep$prep.json 2> $O/bench_prep$prep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_prep$prep.json").read().strip().splitlines()[-1])
    print("prep $prep:", d["sclens_wall_s"], [q["wall_s"] for q in d["observed"]["decisions_per_step"]], d["observed"]["phase_s_rank0_last_step"])
except Exception as e:
    print("prep $prep: no result", e)
PY
done
cat $O/summary.txt
bash scripts/gpu_round4_c.sh
