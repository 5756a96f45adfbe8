#!/bin/bash
# round 5, final GPU call (after the certified ensemble tail): the -m gpu suite on the final tree, the driver's exact bench command, the eigensolve alone, kernel statistics of
# one step (rocprofv3 --kernel-trace --stats, two streams: the summary the bench line's stage durations are checked against)
set -x
O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; wc -c $O/bench_line.json; cp bench_detail.json $O/ 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_line.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','value_strict_fp32','strict_steps','decisions_differ')});print(d['roofline']['frac'],d['roofline'].get('stage_frac'),d['observed'])"
LOW_HALF=1 timeout 300 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig.log 2>&1; tail -3 $O/perf_eig.log
cd /tmp
SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_rocprof.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_rocprof.err
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof -name "*.db" -delete
cd $GRAFT_REPO_ROOT
du -sh $O; find $O -name "*kernel_stats.csv"
