#!/bin/bash
# First GPU call of round 4: (1) what round 3 wrote after its GPU budget was spent, (2) the decision that moved between the
# round-2 and round-3 builds at cfg4 seed 1019 (S 19 -> 18): accelerated and strict arithmetic on THAT seed with the whole search
# trace, then one switch at a time, (3) first phase A/B, (4) the PMC passes, each under its own time limit.
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
ulimit -c 0
SCLENS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q -k "separate_scales" > $O/pytest_scales.log 2>&1; echo "two-scale rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_scales.log
SCLENS_TEST_EXPERIMENTAL=1 timeout 1200 python -m pytest tests/test_gpu_bench_size.py -m gpu -x -q -s > $O/pytest_bench_size.log 2>&1; echo "bench-size (default + strict) rc=$?" >> $O/summary.txt
grep "bench-size parity\|passed\|failed" $O/pytest_bench_size.log
SCLENS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_sclens.py -m gpu -x -q -k "chained_first_phase" > $O/pytest_chain.log 2>&1; echo "chained first phase rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_chain.log

show() {  # name
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    for q in d["observed"]["decisions_per_step"]:
        print("$1", "acc", q["search_iters"], q["p_"], q["signals"], q["robust_signals"], q["min_abs_margin"], q["wall_s"], q["d5_second_smallest"])
    s = d.get("extra", {}).get("strict_fp32")
    if s:
        q = s["decisions"]
        print("$1", "strict", q["search_iters"], q["p_"], q["signals"], q["robust_signals"], q["min_abs_margin"], q["wall_s"], q["d5_second_smallest"])
        print("$1", "decisions_differ", d["decisions_differ"], "max diff", s["max_abs_diff_d5_second_smallest"], "p_th", q["p_th"])
    print("$1", "phases", d["observed"]["phase_s_rank0_last_step"])
except Exception as e:
    print("$1", "no result:", e)
PY
}
# (2) seed 1019 = the last timed step of the driver's 20-step run
timeout 900 python bench.py --steps 1 --warmup 0 --seed-base 1019 --no-cpu-baseline --no-roofline --strict-fp32 on > $O/seed1019.json 2> $O/seed1019.err
show seed1019
one() {  # name VAR=value ...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 1 --warmup 0 --seed-base 1019 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/$name.json 2> $O/$name.err
  show $name
}
one s1019_stein3 SCLENS_HIP_STEIN_ITS=3
one s1019_bisectdiv SCLENS_HIP_BISECT_DIV=1
one s1019_q2v3 SCLENS_HIP_Q2_VARIANT=3
one s1019_sy2sb0 SCLENS_HIP_SY2SB_SPLIT=0
one s1019_q1s0 SCLENS_HIP_Q1_SPLIT=0
one s1019_gram0 SCLENS_HIP_GRAM_SPLIT=0
one s1019_r2like SCLENS_HIP_STEIN_ITS=3 SCLENS_HIP_BISECT_DIV=1 SCLENS_HIP_Q2_VARIANT=3 SCLENS_HIP_SY2SB_SPLIT=0 SCLENS_HIP_Q1_SPLIT=0 SCLENS_HIP_GRAM_SPLIT=0
# (3) first phase: default against the chained schedule (seed 1000)
one fp_default SCLENS_FIRST_PHASE=default
one fp_chain SCLENS_FIRST_PHASE=chain
# (4) HBM traffic of one eigendecomposition on this build
export LOW_HALF=1 TWO_STAGE=1
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_eig_f -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" >> /root/repo/$O/summary.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_eig_w -- python3 /root/repo/scripts/perf_eig.py 30016 2048 15008 > /root/repo/$O/pmc_write.log 2>&1; echo "pmc write rc=$?" >> /root/repo/$O/summary.txt
cd /root/repo
unset LOW_HALF TWO_STAGE
timeout 300 python3 scripts/pmc_summary.py /tmp/pmc_eig_f /tmp/pmc_eig_w > $O/pmc_eig_summary.txt 2>&1
tail -n 30 $O/pmc_eig_summary.txt
cat $O/summary.txt
