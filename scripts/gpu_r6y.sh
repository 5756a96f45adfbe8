#!/bin/bash
# round 6, call y: what the two contexts hold as named scratch at the phase boundaries of an fp32 call (context option debug = 2)
O=gpurun_out/r6y; mkdir -p $O
export TMPDIR=/tmp
SCLENS_HIP_OPTIONS=debug=2 timeout 900 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench.json 2> $O/bench.err
grep "workspaces" $O/bench.err | tail -60 > $O/workspaces_fp32.log; tail -45 $O/workspaces_fp32.log
