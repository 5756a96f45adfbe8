#!/bin/bash
# round 5: the 0/1-image kernel after a cosmetic change (tests), the other two single-GPU configurations of BASELINE.json
set -x
O=gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gram_bits.py tests/test_gpu_golden.py -m gpu -q > $O/pytest_part.log 2>&1; tail -3 $O/pytest_part.log
SCLENS_BENCH_DETAIL=$O/detail_cfg2.json timeout 600 python3 bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 600 $O/bench_cfg2.json
SCLENS_BENCH_DETAIL=$O/detail_cfg3.json timeout 900 python3 bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -c 600 $O/bench_cfg3.json
python3 -c "
import json
for c in ('cfg2','cfg3'):
    d=json.load(open('$O/detail_'+c+'.json')); print(c, d['sclens_wall_s'], d['value'], d.get('value_strict_fp32'), d['observed']['phase_s_rank0_last_step'], d['observed']['signals'], d['observed']['search_iters'], d['config'].get('ensemble_tail'))
"
