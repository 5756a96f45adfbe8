#!/bin/bash
# round 6, call n: chunked-session tests on the current build (co-occurrence product per chunk, dispatch from api.sclens), then the whole
# cfg5 call with precision = 0 and the automatic pattern-cache budget
O=gpurun_out/r6n; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_gram_bits.py -x -q > $O/pytest_chunked.log 2>&1; echo "pytest rc $?" >> $O/pytest_chunked.log; tail -6 $O/pytest_chunked.log
grep -q "failed\|error" $O/pytest_chunked.log && exit 1
timeout 3000 python scripts/atlas_chunked_run.py --precision 0 --out $O/cfg5_whole_call_p0.json > $O/cfg5_whole_call_p0.log 2>&1; echo "whole call rc $?"; grep "phase_s\|wall_s\|chunk_\|pool_peak\|\"k\"\|p_\|n_search" $O/cfg5_whole_call_p0.json | head -20
