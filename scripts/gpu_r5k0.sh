#!/bin/bash
# round 5: HBM traffic of the eigensolve on the final build (Q2 with passes of eight blocks), FETCH_SIZE / WRITE_SIZE in separate passes;
# the new pattern-image test
set -x
O=gpurun_out/r5k0; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gram_bits.py -m gpu -q -x -k "written_once" > $O/pytest_mask.log 2>&1; tail -3 $O/pytest_mask.log
export LOW_HALF=1 TWO_STAGE=1 REPS=1
REGEX='sbr_q2_apply|gemm_split_kernel|gemm_nt_big|sbr_chase_mb|tri_stein|gemm_kernel|split_image|sbr_q2_build|tri_bisect|sbr_panel_small|sbr_gram64|sbr_vmul|sbr_rmul|k_absmax|sbr_q1|sbr_w_split'
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX" --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/scripts/perf_eig.py 30016 2048 15008 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
  F=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" $GRAFT_REPO_ROOT/$O/pmc_${c}_per_kernel.csv <<'PY'
import collections, csv, re, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    k = k.split("(")[0][:80].replace(",", ";")
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,calls,total\n")
    for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write("%s,%d,%.6g\n" % (k, c, v))
PY
done
cd $GRAFT_REPO_ROOT
head -8 $O/pmc_FETCH_SIZE_per_kernel.csv $O/pmc_WRITE_SIZE_per_kernel.csv
