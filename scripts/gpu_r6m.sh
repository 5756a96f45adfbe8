#!/bin/bash
# round 6, call m: the fp32 Q2 pass length inside the whole call (two streams), same box: 4 / 8 / 16 blocks
O=gpurun_out/r6m; mkdir -p $O
export TMPDIR=/tmp
for b in 4 16 8 4 16; do
  SCLENS_HIP_OPTIONS=q2_fp32_blocks=$b timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_fp32_q2b$b.json 2> $O/bench_fp32_q2b$b.err
  python3 -c "
import json;d=json.load(open('$O/bench_fp32_q2b$b.json'));print('q2_fp32_blocks=$b', d['ms_per_step'], d['observed']['wall_s_per_step'])"
done
