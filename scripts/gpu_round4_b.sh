#!/bin/bash
# Second GPU call of round 4: the second back-transformation from pre-built images (variants 8 / 9) -- correctness against the
# unblocked reference, time against variant 7 at the bench's order -- then the child-process launch tests and the stream count.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "second_back_transformation" > $O/pytest_q2.log 2>&1; echo "q2 tests rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_q2.log
timeout 900 python scripts/q2_variants.py 30016 15008 7 8 9 > $O/q2_variants_15008.log 2>&1; echo "q2 variants rc=$?" >> $O/summary.txt
cat $O/q2_variants_15008.log | tail -n 8
export LOW_HALF=1 TWO_STAGE=1
for v in 8 9 7; do
  SCLENS_HIP_Q2_VARIANT=$v timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_v$v.log; echo "variant $v: $(cat $O/eig_v$v.log)"
done
SCLENS_HIP_Q2_VARIANT=8 timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_v8.log; echo "all vectors v8: $(cat $O/eig_all_v8.log)"
SCLENS_HIP_Q2_VARIANT=7 timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_v7.log; echo "all vectors v7: $(cat $O/eig_all_v7.log)"
unset LOW_HALF TWO_STAGE
timeout 1200 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest_multirank.log 2>&1; echo "multirank rc=$?" >> $O/summary.txt
tail -n 15 $O/pytest_multirank.log
for st in 2 3; do
  timeout 900 python bench.py --steps 1 --warmup 1 --streams $st --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_streams$st.json 2> $O/bench_streams$st.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_streams$st.json").read().strip().splitlines()[-1])
    print("streams $st:", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["search_iters"], d["observed"]["p_"])
except Exception as e:
    print("streams $st: no result", e)
PY
done
cat $O/summary.txt
