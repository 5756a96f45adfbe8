#!/bin/bash
# round 6, call v: dense fp32 Gram products in slices of 32 768 cells (context option gram_ksplit): accuracy against the exact sparse form,
# the float64 spectra tests at cfg4 and cfg5, the time of the call
O=gpurun_out/r6v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/perf_gram_sparse.py cfg4 1 > $O/gram_ab_ksplit.log 2>&1; grep "dense, fp32\|dense fp32 vs\|max |sparse" $O/gram_ab_ksplit.log | cut -c1-400
timeout 2400 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_chunked.py tests/test_gpu_kernels.py -q -s -k "spectrum or one_million or gram or wishart" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; grep "cfg4 spectrum\|cfg5 precision\|passed\|failed" $O/pytest.log | cut -c1-500
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_fp32.json 2> $O/bench_fp32.err; python3 -c "
import json;d=json.load(open('$O/bench_fp32.json'));print(d['ms_per_step'], d['observed']['wall_s_per_step'])"
