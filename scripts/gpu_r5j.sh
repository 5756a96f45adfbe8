#!/bin/bash
# round 5, tenth GPU call: the normalisation kernels with four loads per lane in flight and the 0/1 image written once (k_mask_fused):
# parity tests that cover them, kernel statistics of one call on ONE stream (every kernel alone), two timed steps
set -x
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gram_bits.py tests/test_gpu_sclens.py tests/test_gpu_golden.py tests/test_gpu_pattern.py tests/test_gpu_bench_size.py -m gpu -q -x > $O/pytest_part.log 2>&1; tail -4 $O/pytest_part.log
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
SCLENS_BENCH_DETAIL=$O/detail_default.json timeout 700 $B > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json;d=json.load(open('$O/detail_default.json'));o=d['observed'];print('default', d['sclens_wall_s'], o['phase_s_rank0_last_step'], [q['wall_s'] for q in o['decisions_per_step']], o['search_iters'])"
cd /tmp
SCLENS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_one_stream.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_one_stream -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $GRAFT_REPO_ROOT/$O/bench_one_stream.json 2> $GRAFT_REPO_ROOT/$O/bench_one_stream.err
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof_one_stream -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof_one_stream -name "*.db" -delete
cd $GRAFT_REPO_ROOT
grep -E "k_col_stats|k_dense_fused|k_val_set|k_mask|k_col_cent|k_gene_vecs|k_row_norms|k_val_init|k_row_sums|fillBuffer" $O/prof_one_stream/step_kernel_stats.csv | cut -c1-200
