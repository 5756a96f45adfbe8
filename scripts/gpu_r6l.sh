#!/bin/bash
# round 6, call l: three concurrent decompositions at precision = 0 (two: 41.1 s, one: 46.7 s)
O=gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off --streams 3 > $O/bench_fp32_three_streams.json 2> $O/bench_fp32_three_streams.err; echo "bench rc $?"; python3 -c "
import json;d=json.load(open('$O/bench_fp32_three_streams.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','dtype')}, d['observed'])"
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_fp32_two_streams.json 2> $O/bench_fp32_two_streams.err; echo "bench rc $?"; python3 -c "
import json;d=json.load(open('$O/bench_fp32_two_streams.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','dtype')}, d['observed'])"
