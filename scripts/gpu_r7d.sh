#!/bin/bash
# round 6, call 7d: two further random sweeps against the oracle on the final build -- one with the shared inverse-iteration block
O=gpurun_out/r7d; mkdir -p $O
export TMPDIR=/tmp
SCLENS_HIP_OPTIONS=stein_shared=1 timeout 900 python scripts/fuzz_parity.py 150 41 certified > $O/fuzz_150_seed41_stein_shared.log 2>&1; tail -2 $O/fuzz_150_seed41_stein_shared.log | cut -c1-300
timeout 900 python scripts/fuzz_parity.py 150 51 certified > $O/fuzz_150_seed51.log 2>&1; tail -2 $O/fuzz_150_seed51.log | cut -c1-300
