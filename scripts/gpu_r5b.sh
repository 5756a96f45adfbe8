#!/bin/bash
# round 5, second GPU call: (1) the split implicit operator of the partial eigensolver + the refactor's tests that matter, (2) the partial-chip
# stages on a CU-masked stream: chase alone under masks of 64 / 96 / 128 CUs, the eigensolve alone, a Gram product beside an eigensolve,
# (3) whole calls: lock-step rounds and pipelined search with and without the masks
set -x
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_sclens.py tests/test_gpu_sbr.py tests/test_gpu_kernels.py tests/test_gpu_gram_bits.py -m gpu -x -q > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
for pipe in 1 0; do
  SCLENS_HIP_OPTIONS="split_pipe=$pipe" timeout 600 python scripts/perf_split_products.py 30016 100000 corr,gram,bits >> $O/split_pipe.log 2>&1
done
grep -v "^$" $O/split_pipe.log | tail -40
for cus in 0 64 96 128; do
  echo "== perf_eig pstage_cus=$cus" >> $O/mask_alone.log
  SCLENS_HIP_OPTIONS="pstage_cus=$cus" LOW_HALF=1 timeout 300 python scripts/perf_eig.py 30016 2048 15008 >> $O/mask_alone.log 2>&1
done
for cus in 0 64 96; do
  echo "== perf_overlap pstage_cus=$cus" >> $O/mask_overlap.log
  SCLENS_HIP_OPTIONS="pstage_cus=$cus" timeout 300 python scripts/perf_overlap.py 30016 100000 3 >> $O/mask_overlap.log 2>&1
done
cat $O/mask_alone.log $O/mask_overlap.log | grep -v "^$" | tail -40
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
run() { name=$1; shift; env "$@" SCLENS_BENCH_DETAIL=$O/detail_$name.json timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d=json.load(open("$O/detail_$name.json")); print("$name", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], [q["wall_s"] for q in d["observed"]["decisions_per_step"]], d["observed"]["search_iters"], d["observed"]["p_"], d["observed"]["ensemble_partial_eig"])
except Exception as e: print("$name failed", e)
PY
}
run default A=1
run nosplitche SCLENS_HIP_OPTIONS="chefsi_split=0"
run nopipe SCLENS_HIP_OPTIONS="split_pipe=0"
run mask64 SCLENS_HIP_OPTIONS="pstage_cus=64"
run pipe SCLENS_SEARCH_PIPELINE=1
run pipe_mask64 SCLENS_SEARCH_PIPELINE=1 SCLENS_HIP_OPTIONS="pstage_cus=64"
run pipe_mask96 SCLENS_SEARCH_PIPELINE=1 SCLENS_HIP_OPTIONS="pstage_cus=96"
run pipe_mask64_s1 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=1.0 SCLENS_HIP_OPTIONS="pstage_cus=64"
