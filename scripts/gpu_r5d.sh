#!/bin/bash
# round 5, fourth GPU call: the -m gpu suite again (r5c's log was lost: its 85 MB kernel trace pushed gpurun_out over the copy-back limit),
# the ensemble's implicit operator on the 64 x 256 split kernel, block offsets staggered inside the pool (placement experiment)
set -x
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
run() { name=$1; shift; env "$@" SCLENS_BENCH_DETAIL=$O/detail_$name.json timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d=json.load(open("$O/detail_$name.json")); o=d["observed"]; print("$name", d["sclens_wall_s"], o["phase_s_rank0_last_step"], [q["wall_s"] for q in o["decisions_per_step"]], o["search_iters"], o["hbm_in_use_GB_after_timed_steps"], o["ensemble_partial_eig"])
except Exception as e: print("$name failed", e)
PY
}
run default A=1
run chefsi_fp32 SCLENS_HIP_OPTIONS="chefsi_split=0"
run stagger SCLENS_HIP_POOL_STAGGER=1
run stagger_three SCLENS_HIP_POOL_STAGGER=1 SCLENS_FIRST_PHASE=three
run three SCLENS_FIRST_PHASE=three
run stagger2 SCLENS_HIP_POOL_STAGGER=1
