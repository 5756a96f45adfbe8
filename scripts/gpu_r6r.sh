#!/bin/bash
# round 6, call r: the atlas configuration against the float64 fixture -- the -m gpu test of both spectra (both precisions), then the whole
# call on one MI355X at precision = 1 and precision = 0 (final build: automatic pattern cache, co-occurrence product per chunk)
O=gpurun_out/r6r; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas timeout 2400 python -m pytest tests/test_gpu_chunked.py -q -s -k "one_million" > $O/pytest_cfg5.log 2>&1; echo "pytest rc $?" >> $O/pytest_cfg5.log; grep -v "^$" $O/pytest_cfg5.log | cut -c1-900 | tail -8
timeout 2400 python scripts/atlas_chunked_run.py --precision 1 --out $O/cfg5_whole_call_p1.json > $O/cfg5_whole_call_p1.log 2>&1; echo "whole call p1 rc $?"; grep "wall_s\|pool_peak\|\"k\"\|\"p_\"\|n_search\|max_abs_err\|tolerance\|lambda_c" $O/cfg5_whole_call_p1.json
timeout 2400 python scripts/atlas_chunked_run.py --precision 0 --out $O/cfg5_whole_call_p0.json > $O/cfg5_whole_call_p0.log 2>&1; echo "whole call p0 rc $?"; grep "wall_s\|pool_peak\|\"k\"\|\"p_\"\|n_search\|max_abs_err\|tolerance\|lambda_c" $O/cfg5_whole_call_p0.json
