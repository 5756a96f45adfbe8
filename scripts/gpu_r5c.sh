#!/bin/bash
# round 5, third GPU call: the whole -m gpu suite on the cleaned-up tree (pipelined split loops default, masks / chefsi split removed,
# scratch released at phase boundaries), then first-phase schedules and the scratch release A/B, then a kernel trace of one step
set -x
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
run() { name=$1; shift; env "$@" SCLENS_BENCH_DETAIL=$O/detail_$name.json timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d=json.load(open("$O/detail_$name.json")); o=d["observed"]; print("$name", d["sclens_wall_s"], o["phase_s_rank0_last_step"], [q["wall_s"] for q in o["decisions_per_step"]], o["search_iters"], o["hbm_in_use_GB_after_timed_steps"])
except Exception as e: print("$name failed", e)
PY
}
run default A=1
run norelease SCLENS_NO_RELEASE=1
run three SCLENS_FIRST_PHASE=three
run chain2 SCLENS_FIRST_PHASE=chain2
run three_norelease SCLENS_FIRST_PHASE=three SCLENS_NO_RELEASE=1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err
echo "rocprof rc $?"
cd $GRAFT_REPO_ROOT
ls -la $O/prof_step | head; find $O/prof_step -name "*stats*" | head
