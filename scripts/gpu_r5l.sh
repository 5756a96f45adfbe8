#!/bin/bash
# round 5: ensemble with the certified tail (ensemble_tail = "certified": tail pairs of a member not converged, matching accepted only
# with the proof that it does not depend on them): its parity tests, the bench-size tests, per-sweep trace, two timed steps
set -x
O=gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sclens.py tests/test_gpu_bench_size.py tests/test_gpu_golden.py tests/test_gpu_multirank.py -m gpu -q -x -s > $O/pytest_part.log 2>&1; tail -5 $O/pytest_part.log; grep "certified tail" $O/pytest_part.log
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
SCLENS_BENCH_DETAIL=$O/detail_default.json timeout 700 $B > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json;d=json.load(open('$O/detail_default.json'));o=d['observed'];print('default', d['sclens_wall_s'], o['phase_s_rank0_last_step'], [q['wall_s'] for q in o['decisions_per_step']], o['search_iters'], [(q['ensemble_tail'], q['members_solved_again'], q['robust_signals']) for q in o['decisions_per_step']])"
SCLENS_HIP_OPTIONS="debug=1" SCLENS_BENCH_DETAIL=$O/detail_debug.json timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_debug.json 2> $O/bench_debug.err
grep "chefsi" $O/bench_debug.err | cut -c1-300 > $O/chefsi_trace.log; wc -l $O/chefsi_trace.log; head -6 $O/chefsi_trace.log
