#!/bin/bash
# First GPU call of round 4: what round 3 wrote after its GPU budget was spent, then the evidence that round 3 could not finish.
# Every profiler command sits under its own `timeout`: the WRITE_SIZE pass of round 3 produced nothing for 42 minutes and took
# the rest of that round's budget with it (scripts/gpu_round3_x2.sh had no limit on it).
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
ulimit -c 0
# (1) code and tests not yet seen on hardware: the second scale of the split update, the all-fp32 reference point at order 30 000
SCLENS_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "separate_scales" > $O/pytest_scales.log 2>&1; echo "two-scale rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_scales.log
SCLENS_TEST_EXPERIMENTAL=1 timeout 1500 python -m pytest tests/test_gpu_bench_size.py -m gpu -x -q -s > $O/pytest_bench_size.log 2>&1; echo "bench-size (default + strict) rc=$?" >> $O/summary.txt
grep "bench-size parity\|passed\|failed" $O/pytest_bench_size.log
SCLENS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_sclens.py -m gpu -x -q -k "chained_first_phase" > $O/pytest_chain.log 2>&1; echo "chained first phase rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_chain.log
for fp in default chain; do
  SCLENS_FIRST_PHASE=$fp timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_first_phase_$fp.json 2> $O/bench_first_phase_$fp.err
  python3 - <<PY
import json
d = json.loads(open("$O/bench_first_phase_$fp.json").read().strip().splitlines()[-1])
print("first phase = $fp:", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["signals"], d["observed"]["search_iters"], d["observed"]["p_"])
PY
done
# (2) the whole suite as the driver runs it
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 4 $O/pytest_gpu.log
# (3) HBM traffic of one eigendecomposition on this build (round 3 has it for the mid-round build only)
export LOW_HALF=1 TWO_STAGE=1
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_eig_f -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" >> /root/repo/$O/summary.txt
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_eig_w -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_write.log 2>&1; echo "pmc write rc=$?" >> /root/repo/$O/summary.txt
cd /root/repo
unset LOW_HALF TWO_STAGE
timeout 300 python3 scripts/pmc_summary.py /tmp/pmc_eig_f /tmp/pmc_eig_w > $O/pmc_eig_summary.txt 2>&1
tail -n 20 $O/pmc_eig_summary.txt
cat $O/summary.txt
