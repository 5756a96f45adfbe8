#!/bin/bash
# round 5, seventh GPU call: Q2 with passes of 8 / 6 blocks (variants 16 / 17) against 15; the parallel split_image_rows (partial
# eigensolver tests + ensemble time); per-phase pool peaks; kernel statistics of one call on ONE stream (every kernel alone)
set -x
O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/q2_variants.py 2048 512 ref 15 16 17 > $O/q2_small.log 2>&1; tail -8 $O/q2_small.log
timeout 900 python scripts/q2_variants.py 30016 15008 15 16 17 > $O/q2_half.log 2>&1; tail -8 $O/q2_half.log
timeout 900 python scripts/q2_variants.py 30016 30016 15 16 > $O/q2_all.log 2>&1; tail -6 $O/q2_all.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -x > $O/pytest_part.log 2>&1; tail -4 $O/pytest_part.log
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
run() { name=$1; shift; env "$@" SCLENS_BENCH_DETAIL=$O/detail_$name.json timeout 700 $B > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d=json.load(open("$O/detail_$name.json")); o=d["observed"]; print("$name", d["sclens_wall_s"], o["phase_s_rank0_last_step"], [q["wall_s"] for q in o["decisions_per_step"]], o["search_iters"], o["hbm_in_use_GB_after_timed_steps"], o["hbm_peak_live_GB"], o.get("hbm_peak_live_GB_by_phase"))
except Exception as e: print("$name failed", e)
PY
}
run default A=1
run q2v16 SCLENS_HIP_OPTIONS="q2_variant=16"
cd /tmp
SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_one_stream.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_one_stream -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_one_stream.json 2> $GRAFT_REPO_ROOT/$O/bench_one_stream.err
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof_one_stream -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof_one_stream -name "*.db" -delete
cd $GRAFT_REPO_ROOT
du -sh $O; find $O -name "*kernel_stats.csv"
