#!/bin/bash
# search statistic (corr_split_kernel) with its operand stages through registers instead of LDS-DMA (SCLENS_HIP_CORR_RS=1): test, A/B
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4ab
mkdir -p $O
ulimit -c 0
SCLENS_HIP_CORR_RS=1 timeout 600 python -m pytest tests/test_gpu_gram_bits.py tests/test_gpu_sclens.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest (RS=1) rc=$rc" >> $O/summary.txt; tail -n 4 $O/pytest.log
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    for x in d["observed"]["decisions_per_step"]:
        print("$name step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
    for w in (0, 1):
        print("   worker", w, [q[2] for q in d["observed"]["search_job_s_last_step"] if q[0] == w])
except Exception as e:
    print("$name: no result", e)
PY
}
run rs0 SCLENS_HIP_CORR_RS=0
run rs1 SCLENS_HIP_CORR_RS=1
run rs0b SCLENS_HIP_CORR_RS=0
run rs1b SCLENS_HIP_CORR_RS=1
cat $O/summary.txt
