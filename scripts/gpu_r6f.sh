#!/bin/bash
# round 6, call f: where the dense fp32 Gram product and the sparse-structured one differ at cfg4 (8e-4 of the largest entry in call e)
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python scripts/perf_gram_sparse.py cfg4 1 > $O/gram_sparse_ab_cfg4.log 2>&1; cat $O/gram_sparse_ab_cfg4.log | tail -24
