#!/bin/bash
# round 6, call p: the other single-GPU configurations of BASELINE.json on the final build; a random end-to-end sweep against the oracle
O=gpurun_out/r6p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc $?"
timeout 1200 python3 bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc $?"
python3 -c "
import json
for c in ('cfg2','cfg3'):
    d=json.load(open('$O/bench_%s.json'%c)); print(c, {k:d.get(k) for k in ('value','ms_per_step','steps','dtype','value_split_f16','split_ms_per_step','decisions_differ')}, d['observed'])"
timeout 1500 python scripts/fuzz_parity.py 100 31 certified > $O/fuzz_parity_100.log 2>&1; tail -6 $O/fuzz_parity_100.log
