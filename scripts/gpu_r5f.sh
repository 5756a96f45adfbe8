#!/bin/bash
# round 5, sixth GPU call: is the search slower when the device is nearly full (ballast, a lower pool cap)? three streams? PMC traffic of
# the eigensolve on this build
set -x
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off"
run() { name=$1; shift; extra=""; while [ "${1#--}" != "$1" ]; do extra="$extra $1 $2"; shift 2; done; env "$@" SCLENS_BENCH_DETAIL=$O/detail_$name.json timeout 700 $B $extra > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d=json.load(open("$O/detail_$name.json")); o=d["observed"]; print("$name", d["sclens_wall_s"], o["phase_s_rank0_last_step"], [q["wall_s"] for q in o["decisions_per_step"]], o["search_iters"], o["hbm_in_use_GB_after_timed_steps"], o["hbm_peak_live_GB"])
except Exception as e: print("$name failed", e)
PY
}
run default A=1
run ballast6 --ballast-gb 6 A=1
run cap200 SCLENS_HIP_POOL_MAX_GB=200
run cap160 SCLENS_HIP_POOL_MAX_GB=160
run streams3 --streams 3 A=1
run ballast_minus SCLENS_HIP_POOL_MAX_GB=230
export LOW_HALF=1 TWO_STAGE=1 REPS=1
REGEX='sbr_q2_apply|gemm_split_kernel|gemm_nt_big|sbr_chase_mb|tri_stein|gemm_kernel|split_image|sbr_q2_build|tri_bisect|sbr_panel_small|sbr_gram64|sbr_vmul|sbr_rmul|k_absmax|sbr_q1|sbr_w_split'
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX" --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/scripts/perf_eig.py 30016 2048 15008 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
  F=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" $GRAFT_REPO_ROOT/$O/pmc_${c}_per_kernel.csv <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:80].replace(",", ";")
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,calls,total\n")
    for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write("%s,%d,%.6g\n" % (k, c, v))
PY
done
cd $GRAFT_REPO_ROOT
head -12 $O/pmc_FETCH_SIZE_per_kernel.csv $O/pmc_WRITE_SIZE_per_kernel.csv
du -sh $O
