#!/bin/bash
# round 5, fifth GPU call: the -m gpu suite on the final tree, the driver's exact bench command, kernel statistics of one step under the
# default first-phase schedule and under `three` (which kernels of the search slow down?), the eigensolve alone
set -x
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; wc -c $O/bench_line.json; cp bench_detail.json $O/ 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_line.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','value_strict_fp32','strict_steps','decisions_differ')});print(d['roofline']['frac'],d['roofline']['stage_frac'],d['observed'])"
LOW_HALF=1 timeout 300 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig.log 2>&1; tail -3 $O/perf_eig.log
cd /tmp
for mode in default three; do
  FP=""; [ $mode = three ] && FP=three
  SCLENS_FIRST_PHASE=$FP SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_rocprof_$mode.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$mode -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_rocprof_$mode.json 2> $GRAFT_REPO_ROOT/$O/bench_rocprof_$mode.err
  echo "rocprof $mode rc $?"
  find $GRAFT_REPO_ROOT/$O/prof_$mode -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof_$mode -name "*.db" -delete
done
cd $GRAFT_REPO_ROOT
du -sh $O; find $O -name "*kernel_stats.csv"
