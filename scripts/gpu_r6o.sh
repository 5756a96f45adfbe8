#!/bin/bash
# round 6, call o: the driver's exact bench command on the (near-)final build
O=gpurun_out/r6o; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; wc -c $O/bench_line.json; cp bench_detail.json $O/ 2>/dev/null
python3 -c "import json;d=json.load(open('$O/bench_line.json'));print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','dtype','value_split_f16','split_steps','split_ms_per_step','decisions_differ','bench_wall_s')});print(d['roofline']['frac'],d['roofline'].get('stage_frac'),d['observed'])"
