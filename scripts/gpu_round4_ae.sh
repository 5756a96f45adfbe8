#!/bin/bash
# first-phase schedules again, now that no call starts with hipMalloc: default, chain2 (binarised first), three (all at once)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4ae
mkdir -p $O
ulimit -c 0
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["ms_per_step"])
    for x in d["observed"]["decisions_per_step"]:
        print("   step", x["seed"], x["wall_s"], "first", x["phase_s"]["spectra_signal_vectors_vr2"], "search", x["phase_s"]["sparsity_search"], "ens", x["phase_s"]["perturbation_ensemble"], "S", x["search_iters"], x["p_"], x["signals"], x["robust_signals"])
except Exception as e:
    print("$name: no result", e)
PY
}
run default A=1
run chain2 SCLENS_FIRST_PHASE=chain2
run three SCLENS_FIRST_PHASE=three
