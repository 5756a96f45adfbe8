#!/bin/bash
# pool cap: half of the device memory (default) against 250 GB -- does the first phase stop paying for hipMalloc after a call whose
# freed blocks exceeded the cap?
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4ac
mkdir -p $O
ulimit -c 0
run() {
  local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["ms_per_step"], "HBM in use", d["observed"]["hbm_in_use_GB_after_timed_steps"])
    for x in d["observed"]["decisions_per_step"]:
        print("   step", x["seed"], x["wall_s"], "first", x["phase_s"]["spectra_signal_vectors_vr2"], "search", x["phase_s"]["sparsity_search"], "ens", x["phase_s"]["perturbation_ensemble"], [ (q[0], q[1]) for q in x["first_phase_jobs_s"] if q[0] in ("data_spectrum", "null_pattern_build")])
except Exception as e:
    print("$name: no result", e)
PY
}
run cap_default A=1
run cap_250 SCLENS_HIP_POOL_MAX_GB=250
run cap_default_b A=1
run cap_250_b SCLENS_HIP_POOL_MAX_GB=250
