#!/bin/bash
# round 6, call e: the sparse-structured Gram product: its tests, then the A/B at cfg4 against the dense products; the fp32
# bench step on one stream; kernel statistics (rocprofv3 --kernel-trace --stats) of one fp32 step
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gram_sparse.py -x -q -s > $O/pytest_gram_sparse.log 2>&1; echo "pytest rc $?" >> $O/pytest_gram_sparse.log; grep -v "^$" $O/pytest_gram_sparse.log | tail -15
timeout 900 python scripts/perf_gram_sparse.py cfg4 2 > $O/gram_sparse_ab_cfg4.log 2>&1; cat $O/gram_sparse_ab_cfg4.log | tail -20
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off --streams 1 > $O/bench_fp32_one_stream.json 2> $O/bench_fp32_one_stream.err; echo "bench rc $?"; python3 -c "
import json;d=json.load(open('$O/bench_fp32_one_stream.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','dtype')}, d['observed'])"
timeout 2400 python -m pytest tests/test_gpu_multirank.py "tests/test_gpu_sclens.py::test_eight_rank_rehearsal_of_the_whole_call" tests/test_gpu_bench_size.py -x -q -s -k "not accelerated and not two_ranks" > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log; grep -v "^$" $O/pytest_new.log | cut -c1-600 | tail -30
cd /tmp
SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_rocprof.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_rocprof.err
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof -name "*.db" -delete
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_stats.csv" | head -2
