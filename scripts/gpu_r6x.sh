#!/bin/bash
# round 6, call x: row norms of the eigenvector block of the order-30 016 eigensolve under both precisions (how noisy are the vectors?)
O=gpurun_out/r6x; mkdir -p $O
export TMPDIR=/tmp
for p in 0 1; do
  SCLENS_HIP_OPTIONS=precision=$p LOW_HALF=1 PRINT_HASH=1 REPS=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "crc32\|rep=0" | cut -c1-260
done > $O/eig_vector_norms.log 2>&1
cat $O/eig_vector_norms.log
