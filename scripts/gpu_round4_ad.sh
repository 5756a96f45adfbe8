#!/bin/bash
# the pool cap's new default (device memory less an eighth): smoke, the sclens / pattern / multirank tests, the bench line
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4ad
mkdir -p $O
ulimit -c 0
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_sclens.py tests/test_gpu_pattern.py tests/test_gpu_multirank.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt; tail -n 3 $O/pytest.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_cfg4_final.json 2> $O/bench_cfg4_final.err; echo "bench cfg4 rc=$?" >> $O/summary.txt
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4ad/bench_cfg4_final.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], "strict", d.get("value_strict_fp32"), "differ", d.get("decisions_differ"), "HBM", d["observed"]["hbm_in_use_GB_after_timed_steps"])
    for x in d["observed"]["decisions_per_step"]:
        print("  step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
    r = d["roofline"]; print("roofline", r["launch_ms"], r["frac"], r["stage_ms"])
except Exception as e:
    print("bench: no result", e)
PY
cat $O/summary.txt
