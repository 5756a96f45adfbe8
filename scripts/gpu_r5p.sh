#!/bin/bash
# round 5: randomised end-to-end sweep against the oracle with the certified ensemble tail forced on at small sizes
O=gpurun_out/r5p; mkdir -p $O
timeout 1500 python scripts/fuzz_parity.py 150 21 certified > $O/fuzz_certified.log 2>&1; tail -4 $O/fuzz_certified.log; grep -c MISMATCH $O/fuzz_certified.log
