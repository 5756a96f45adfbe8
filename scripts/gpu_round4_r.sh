#!/bin/bash
# (1) W = A22 V of the band reduction from fp16 pieces (sbr_w_split): tests, stage time on/off, kernel trace
# (2) sparsity search without round barriers (SCLENS_SEARCH_PIPELINE=1) at several worker staggers against the rounds
# (3) draw tests with the threaded bucket pass of the null-matrix shuffle
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p $O
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "sy2sb or two_stage or switches" > $O/pytest_sbr.log 2>&1; rc=$?; echo "pytest sbr rc=$rc" >> $O/summary.txt; tail -n 5 $O/pytest_sbr.log
if [ $rc -ne 0 ]; then export SCLENS_HIP_SY2SB_WSPLIT=0; echo "W split OFF for the rest" >> $O/summary.txt; fi
for w in 0 4096; do
  echo "WSPLIT=$w: $(SCLENS_HIP_SY2SB_WSPLIT=$w timeout 300 python scripts/perf_sbr.py 30016 2>&1 | tail -n 2 | tr '\n' ' ')"
done | tee $O/wsplit_sy2sb.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/trace_sy2sb -- python3 /root/repo/scripts/perf_sbr.py 30016 > /root/repo/$O/trace_sy2sb.log 2>&1 )
python3 - <<'PY' | tee -a gpurun_out/r4r/wsplit_sy2sb.log
import glob, csv
for f in glob.glob("gpurun_out/r4r/trace_sy2sb/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"]) / 3e6:9.2f} ms/run avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
timeout 600 python -m pytest tests/test_gpu_pattern.py tests/test_gpu_sclens.py -m gpu -x -q > $O/pytest_draws.log 2>&1; echo "pytest draws rc=$?" >> $O/summary.txt; tail -n 3 $O/pytest_draws.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    ph = d["observed"]["phase_s_rank0_last_step"]
    dec = d["observed"]["decisions_per_step"][-1]
    print("$name:", d["sclens_wall_s"], "search", ph["sparsity_search"], "first", ph["spectra_signal_vectors_vr2"], "ens", ph["perturbation_ensemble"],
          "S", dec["search_iters"], "p_", dec["p_"], "signals", dec["signals"], dec["robust_signals"])
    for w in (0, 1):
        print("   worker", w, [q[2] for q in d["observed"]["search_job_s_last_step"] if q[0] == w])
except Exception as e:
    print("$name: no result", e)
PY
}
run rounds SCLENS_SEARCH_PIPELINE=0
run rounds_w0 SCLENS_SEARCH_PIPELINE=0 SCLENS_HIP_SY2SB_WSPLIT=0
run pipe_s0 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=0
run pipe_s040 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=0.4
run pipe_s065 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=0.65
run pipe_s090 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=0.9
run pipe_s120 SCLENS_SEARCH_PIPELINE=1 SCLENS_SEARCH_STAGGER_S=1.2
cat $O/summary.txt
