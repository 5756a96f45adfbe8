#!/bin/bash
# round 6, call d: the WHOLE sclens() call at 1 000 000 x 30 000 on one MI355X through the chunked session (fp32 arithmetic, precision = 0),
# then the new / changed GPU tests (chunked session, float64 arbiter of the search statistic, 8-rank rehearsals)
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python scripts/atlas_chunked_run.py --precision 0 --out $O/cfg5_whole_call_p0.json > $O/cfg5_whole_call_p0.log 2>&1; echo "whole call rc $?"; tail -45 $O/cfg5_whole_call_p0.log
timeout 2400 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_bench_size.py tests/test_gpu_multirank.py "tests/test_gpu_sclens.py::test_eight_rank_rehearsal_of_the_whole_call" -x -q -s -k "not accelerated" > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log; grep -v "^$" $O/pytest_new.log | tail -30
