#!/bin/bash
# round 5, first GPU call: the suite on the refactored library (options through the ABI, pruned Q2), the driver's exact bench command,
# kernel stats + L2 hit rates of the three full-chip split-fp16 products
set -x
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; wc -c $O/bench_line.json; cp bench_detail.json $O/ 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_line.json'));print({k:d[k] for k in ('value','ms_per_step','steps','value_strict_fp32','strict_steps','decisions_differ')});print(d['roofline']['frac'],d['roofline']['stage_frac'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_split -o split -- python3 $GRAFT_REPO_ROOT/scripts/perf_split_products.py 30016 100000 corr,gram,bits > $GRAFT_REPO_ROOT/$O/split_stats.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "corr_split|gemm_split|gram_bits" -d $GRAFT_REPO_ROOT/$O/pmc_split -o split -- python3 $GRAFT_REPO_ROOT/scripts/perf_split_products.py 30016 100000 corr,gram,bits > $GRAFT_REPO_ROOT/$O/split_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
ls $O/prof_split $O/pmc_split | head -20
python3 scripts/pmc_summary.py $O/pmc_split 2>&1 | tail -20
