#!/bin/bash
# round 6, call c: (1) HBM traffic of the eigensolve at precision = 0 and (2) of the normalisation kernels of one decomposition / one search
# evaluation (FETCH_SIZE / WRITE_SIZE in separate passes); (3) two fp32 bench steps with the binarised Gram as the
# exact co-occurrence product (gram_bits_strict), (4) the bench-size parity test on that build
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
summarise() {
python3 - "$1" "$2" <<'PY'
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
}
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --strict-steps 2 > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench rc $?"
cp bench_detail.json $O/bench_fp32_detail.json 2>/dev/null; tail -c 2500 $O/bench_fp32.json
REGEX_EIG='sbr_q2_apply|gemm_split_kernel|gemm_nt_big|sbr_chase_mb|tri_stein|gemm_kernel|split_image|sbr_q2_build|tri_bisect|sbr_panel_small|sbr_gram64|sbr_vmul|sbr_rmul|k_absmax|sbr_q1|sbr_w_split|sbr_top|sbr_sum|sbr_mirror|k_transpose|sbr_small'
REGEX_SC='k_row_sums|k_col_stats|k_row_norms|k_col_cent|k_dense_fused|k_dense_fill|k_dense_scatter|k_val_init|k_val_set_ones|k_mask_fused|k_mask_scatter|k_gene_vecs|k_row_scale|k_reduce|k_cell_weights|k_weight_scale|k_split_weights'
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  SCLENS_HIP_OPTIONS=precision=0 LOW_HALF=1 TWO_STAGE=1 REPS=1 timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX_EIG" --output-format csv -d /tmp/pmce_$c -- python3 $GRAFT_REPO_ROOT/scripts/perf_eig.py 30016 2048 15008 > $GRAFT_REPO_ROOT/$O/pmc_eig_$c.log 2>&1
  echo "pmc eig $c rc=$?"
  F=$(find /tmp/pmce_$c -name "*counter_collection.csv" | head -1); [ -n "$F" ] && summarise "$F" $GRAFT_REPO_ROOT/$O/pmc_eig_strict_${c}_per_kernel.csv
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "$REGEX_SC" --output-format csv -d /tmp/pmcs_$c -- python3 $GRAFT_REPO_ROOT/scripts/perf_scale.py cfg4 > $GRAFT_REPO_ROOT/$O/pmc_scale_$c.log 2>&1
  echo "pmc scale $c rc=$?"
  F=$(find /tmp/pmcs_$c -name "*counter_collection.csv" | head -1); [ -n "$F" ] && summarise "$F" $GRAFT_REPO_ROOT/$O/pmc_scale_${c}_per_kernel.csv
done
cd $GRAFT_REPO_ROOT
head -12 $O/pmc_eig_strict_*_per_kernel.csv $O/pmc_scale_*_per_kernel.csv; tail -2 $O/pmc_scale_FETCH_SIZE.log
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -x -q -k "accelerated" > $O/pytest_bench_size.log 2>&1; tail -5 $O/pytest_bench_size.log
