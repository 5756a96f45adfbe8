#!/bin/bash
# round 6, call s: kernel statistics of ONE fp32 call on ONE stream (every kernel alone), final build; chunked tests after the cache-budget change
O=gpurun_out/r6s; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_rocprof.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_rocprof.err
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof -name "*.db" -delete
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chunked.py "tests/test_gpu_sclens.py::test_default_call_hands_its_device_memory_back" -q -k "not one_million" > $O/pytest_chunked.log 2>&1; tail -3 $O/pytest_chunked.log
