#!/bin/bash
# round 6, call 7e: the seed-41 sweep again with the default (per-context) inverse-iteration workspaces: are its three soft cases the option's?
O=gpurun_out/r7e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/fuzz_parity.py 150 41 certified > $O/fuzz_150_seed41.log 2>&1; tail -2 $O/fuzz_150_seed41.log | cut -c1-300
cmp <(grep -v " s$" $O/fuzz_150_seed41.log | grep -v amdgpu) <(grep -v " s$" gpurun_out/r7d/fuzz_150_seed41_stein_shared.log | grep -v amdgpu) && echo "identical logs (but for the time)"
