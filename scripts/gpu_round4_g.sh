#!/bin/bash
# W1 of the first back-transformation from split images: tests, stage times, A/B of whole calls; chase on fewer CUs beside a second
# stream; PMC passes with the raw per-dispatch CSVs kept (per-stage split by dispatch order)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -s -k "prepared_ahead or same_bits or eigh or two_stage" > $O/pytest_q1.log 2>&1; echo "q1 tests rc=$?" >> $O/summary.txt
grep "W1 split\|passed\|failed" $O/pytest_q1.log
export LOW_HALF=1 TWO_STAGE=1
for w1 in 1 0; do
  SCLENS_HIP_Q1_W1_SPLIT=$w1 timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=1" > $O/eig_w1split$w1.log; echo "w1 split $w1: $(cat $O/eig_w1split$w1.log)"
done
SCLENS_HIP_Q1_W1_SPLIT=1 timeout 300 python scripts/perf_eig.py 30016 2048 30016 2>&1 | grep "rep=1" > $O/eig_all_w1split1.log; echo "all vectors w1 split 1: $(cat $O/eig_all_w1split1.log)"
export REPS=1
REGEX='sbr_q2_apply|gemm_split_kernel|gemm_nt_big|sbr_chase_mb|tri_stein|gemm_kernel|split_image|sbr_q2_build|tri_bisect|sbr_panel_small|sbr_gram64|sbr_vmul|sbr_rmul|k_absmax|sbr_q1'
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX" --output-format csv -d /tmp/pmc_$c -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> /root/repo/$O/summary.txt
  F=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" /root/repo/$O/pmc_${c}_dispatches.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as fh:  # dispatch order, kernel name (short), counter value: small enough to keep
    fh.write("dispatch,kernel,value\n")
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        fh.write("%s,%s,%s\n" % (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:70].replace(",", ";"), r["Counter_Value"]))
PY
done
cd /root/repo
unset LOW_HALF TWO_STAGE REPS
for cfg in "SCLENS_HIP_Q1_W1_SPLIT=0" "SCLENS_HIP_Q1_W1_SPLIT=1" "SCLENS_HIP_CHASE_WGS=128" "SCLENS_HIP_CHASE_WGS=176" "SCLENS_HIP_Q1_W1_SPLIT=0" "SCLENS_HIP_Q1_W1_SPLIT=1"; do
  env $cfg timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_ab.json 2> $O/bench_ab.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_ab.json").read().strip().splitlines()[-1])
    print("$cfg:", d["sclens_wall_s"], [q["wall_s"] for q in d["observed"]["decisions_per_step"]], d["observed"]["phase_s_rank0_last_step"], d["observed"]["search_iters"], d["observed"]["p_"])
except Exception as e:
    print("$cfg: no result", e)
PY
done
cat $O/summary.txt
