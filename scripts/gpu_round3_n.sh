#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
ulimit -c 0
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('/root/repo/gpurun_out/r3n/bench_$name.json').read().strip().splitlines()[-1])
    print("$name", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["ensemble_partial_eig"])
except Exception as e:
    print("$name FAILED", e); print(open('/root/repo/gpurun_out/r3n/bench_$name.err').read()[-1500:])
PY
}
run base SCLENS_HIP_CHASE_CUMASK=0
run novalcsr SCLENS_HIP_VAL_CSR=0
run cumask SCLENS_HIP_CHASE_CUMASK=1
run gapoff SCLENS_HIP_CHEFSI_TAIL_GAP=1e9
timeout 1500 python scripts/atlas_dry_run.py 1000000 8 $O/atlas_slab_dry_run.json > $O/atlas_stdout.log 2> $O/atlas_stderr.log; echo "atlas dry run rc=$?" >> $O/summary.txt
tail -n 14 $O/atlas_stderr.log
cat $O/summary.txt
