#!/bin/bash
# round 6, call 7b: the inverse iteration's workspaces shared per device (stein_shared = 1) -- eigensolver / session / multi-context
# tests, then the bench's call with the peak footprint by phase, and the same with stein_shared = 0 through the environment
O=gpurun_out/r7b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py tests/test_gpu_sclens.py tests/test_gpu_multirank.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --other-variant off --no-cpu-baseline > $O/bench_shared.json 2> $O/bench_shared.err; tail -c 1500 $O/bench_shared.json
cp bench_detail.json $O/bench_shared_detail.json 2>/dev/null
SCLENS_HIP_OPTIONS=stein_shared=0 timeout 900 python bench.py --steps 2 --warmup 1 --other-variant off --no-cpu-baseline --no-roofline > $O/bench_per_context.json 2> $O/bench_per_context.err; tail -c 1500 $O/bench_per_context.json
cp bench_detail.json $O/bench_per_context_detail.json 2>/dev/null
python - <<'P'
import json,glob
for f in glob.glob('gpurun_out/r7b/*detail*.json')+glob.glob('*detail*.json'):
    d=json.load(open(f))
    def find(o,k):
        if isinstance(o,dict):
            for a,b in o.items():
                if a==k: yield b
                else: yield from find(b,k)
    print(f, list(find(d,'hbm_peak_live_GB_by_phase'))[:1])
P
