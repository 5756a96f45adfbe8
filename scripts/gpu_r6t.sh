#!/bin/bash
# round 6, call t: the whole cfg5 call at precision = 1 with the final pattern-cache budget (device less 200 GiB): peak footprint, time
O=gpurun_out/r6t; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python scripts/atlas_chunked_run.py --precision 1 --out $O/cfg5_whole_call_p1.json > $O/cfg5_whole_call_p1.log 2>&1; echo "whole call p1 rc $?"; grep "wall_s\|pool_peak\|\"k\"\|\"p_\"\|n_search\|chunk_builds\|chunk_visits" $O/cfg5_whole_call_p1.json
