#!/bin/bash
# round 5: case 74 of the fuzz sweep (seed 21) replayed with the tail converged and certified: is the mismatch the search's?
O=gpurun_out/r5q; mkdir -p $O
for tail in converged certified; do
  FUZZ_ONLY=74 timeout 600 python scripts/fuzz_parity.py 150 21 $tail > $O/case74_$tail.log 2>&1; tail -4 $O/case74_$tail.log
done
