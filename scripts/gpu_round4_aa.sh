#!/bin/bash
# gemm_split_kernel with half stages of 16 of K, three in flight (SCLENS_HIP_SPLIT_DEEP, default 1): tests, A/B, bench
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4aa
mkdir -p $O
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_sbr.py tests/test_gpu_kernels.py tests/test_gpu_gram_bits.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/summary.txt; tail -n 6 $O/pytest.log
for d in 1 0 1 0; do
  echo "DEEP=$d: $(SCLENS_HIP_SPLIT_DEEP=$d timeout 300 python scripts/perf_eig.py 30016 60032 15008 2>&1 | grep 'rep=1')"
done 2>&1 | tee $O/eig_deep.log
if [ $rc -ne 0 ]; then export SCLENS_HIP_SPLIT_DEEP=0; echo "DEEP OFF for the bench" >> $O/summary.txt; fi
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4aa/bench.json").read().strip().splitlines()[-1])
    for x in d["observed"]["decisions_per_step"]:
        print("step", x["seed"], x["wall_s"], x["phase_s"], "S", x["search_iters"], "p_", x["p_"], "signals", x["signals"], x["robust_signals"])
except Exception as e:
    print("bench: no result", e)
PY
cat $O/summary.txt
