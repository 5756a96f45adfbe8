#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
ulimit -c 0
ATLAS_M=8192 AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 900 python scripts/atlas_dry_run.py 160000 8 $O/atlas_small.json > $O/atlas_small_stdout.log 2> $O/atlas_small_stderr.log; echo "atlas small rc=$?" >> $O/summary.txt
tail -n 6 $O/atlas_small_stderr.log
# two concurrent decompositions with and without the CU-masked chase stream (cfg4, one step each)
for m in 0 1; do
SCLENS_HIP_CHASE_CUMASK=$m timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --strict-fp32 off > $O/bench_cumask$m.json 2> /dev/null
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r3m/bench_cumask$m.json').read().strip().splitlines()[-1])
print("cumask $m", d["sclens_wall_s"], d["observed"]["phase_s_rank0_last_step"], d["observed"]["ensemble_partial_eig"])
PY
done
cat $O/summary.txt
