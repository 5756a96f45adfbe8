#!/bin/bash
# round 5, ninth GPU call: the -m gpu suite with Q2 variant 16 as the default; workspace table of both contexts at the end of the search
set -x
O=gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
SCLENS_HIP_OPTIONS="debug=2" SCLENS_BENCH_DETAIL=$O/detail_debug2.json timeout 600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_debug2.json 2> $O/bench_debug2.err
grep "workspaces" $O/bench_debug2.err > $O/workspaces.log; tail -60 $O/workspaces.log
python3 -c "import json;d=json.load(open('$O/detail_debug2.json'));print(d['sclens_wall_s'], d['observed']['phase_s_rank0_last_step'])"
du -sh $O
