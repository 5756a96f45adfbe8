#!/bin/bash
# split-fp16 update started from C (buffer loads up front), diagonal block folded into the look-ahead strip, grid of the W kernel:
# tests, A/B of the band reduction and the whole eigensolve, then one bench step
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4s
mkdir -p $O
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_sbr.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; rc=$?; echo "pytest sbr+kernels rc=$rc" >> $O/summary.txt; tail -n 5 $O/pytest_sbr.log
sb() { echo "$1: $(env $2 $3 $4 timeout 300 python scripts/perf_sbr.py 30016 2>&1 | tail -n 1)"; }
{
sb default A=1
sb acc_init_off SCLENS_HIP_SPLIT_ACC_INIT=0
sb fold_diag_off SCLENS_HIP_SY2SB_FOLD_DIAG=0
sb ws_slots_256 SCLENS_HIP_SY2SB_WS_SLOTS=256
sb ws_slots_1024 SCLENS_HIP_SY2SB_WS_SLOTS=1024
sb ws_slots_1536 SCLENS_HIP_SY2SB_WS_SLOTS=1536
sb all_off SCLENS_HIP_SPLIT_ACC_INIT=0 SCLENS_HIP_SY2SB_FOLD_DIAG=0 SCLENS_HIP_SY2SB_WSPLIT=0
} 2>&1 | tee $O/sy2sb_ab.log
for m in default acc_init_off; do
  e=A=1; [ $m = acc_init_off ] && e=SCLENS_HIP_SPLIT_ACC_INIT=0
  echo "eig $m: $(env $e timeout 300 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee $O/eig_ab.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4s/bench.json").read().strip().splitlines()[-1])
    ph = d["observed"]["phase_s_rank0_last_step"]
    dec = d["observed"]["decisions_per_step"][-1]
    print("bench:", d["sclens_wall_s"], "search", ph["sparsity_search"], "first", ph["spectra_signal_vectors_vr2"], "ens", ph["perturbation_ensemble"],
          "S", dec["search_iters"], "p_", dec["p_"], "signals", dec["signals"], dec["robust_signals"])
    print("roofline", d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["stage_ms"])
except Exception as e:
    print("bench: no result", e)
PY
cat $O/summary.txt
