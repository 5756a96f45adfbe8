#!/bin/bash
# round 6, call g: the changed / new GPU tests on the current build (row-sharded first phase on one rank each, chunked session incl. the
# sparse-structured Gram per chunk, gram_sparse auto choice, float64 arbiter of the search statistic, 8-rank rehearsals)
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_atlas.py tests/test_gpu_chunked.py tests/test_gpu_gram_sparse.py tests/test_gpu_multirank.py "tests/test_gpu_sclens.py::test_eight_rank_rehearsal_of_the_whole_call" tests/test_gpu_bench_size.py -q -s -k "not accelerated and not two_ranks and not slab_of_one_rank" > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log; grep -v "^$" $O/pytest_new.log | cut -c1-700 | tail -40
