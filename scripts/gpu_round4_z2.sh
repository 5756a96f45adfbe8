#!/bin/bash
# end of round 4, regression check at the other configurations: cfg2 (3 streams, order 10 000), cfg3 (50 000 x 30 000), the atlas slab test,
# and the first phase with the binarised decomposition first (chain2) on the final build
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4z2
mkdir -p $O
ulimit -c 0
line() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["config"]["workload"] if isinstance(d.get("config"), dict) else "", d["value"], d["ms_per_step"], [ (x["wall_s"], x["search_iters"], x["p_"], x["signals"], x["robust_signals"]) for x in d["observed"]["decisions_per_step"]])
    print("   phases", d["observed"]["phase_s_rank0_last_step"])
except Exception as e:
    print(sys.argv[1], "no result", e)
PY
}
timeout 600 python bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline --strict-fp32 off > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?" >> $O/summary.txt; line $O/bench_cfg2.json
timeout 900 python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?" >> $O/summary.txt; line $O/bench_cfg3.json
SCLENS_FIRST_PHASE=chain2 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4_chain2.json 2> $O/bench_cfg4_chain2.err; echo "cfg4 chain2 rc=$?" >> $O/summary.txt; line $O/bench_cfg4_chain2.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cfg4_default.json 2> $O/bench_cfg4_default.err; echo "cfg4 default rc=$?" >> $O/summary.txt; line $O/bench_cfg4_default.json
SCLENS_TEST_SLOW=1 timeout 900 python -m pytest tests/test_gpu_atlas.py -m gpu -x -q -k atlas_slab -s > $O/pytest_atlas_slab.log 2>&1; echo "atlas slab rc=$?" >> $O/summary.txt; tail -n 5 $O/pytest_atlas_slab.log
cat $O/summary.txt
