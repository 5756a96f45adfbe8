# cfg2 (10 000 x 20 000) through the alternative paths: two-stage solver and split-fp16 search statistic below their default order
run() { name=$1; shift; env "$@" timeout 300 python bench.py --config cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $EXTRA > gpurun_out/cfg2_$name.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/cfg2_$name.json').read().strip().splitlines()[-1]); o=d['observed']; print('$name', d['sclens_wall_s'], o['signals'], o['robust_signals'], o['search_iters'], o['p_'], o['phase_s_rank0_last_step'])"; }
EXTRA="" run default A=1
EXTRA="" run two_stage_s3 SCLENS_HIP_TWO_STAGE=1
EXTRA="--streams 1" run two_stage_s1 SCLENS_HIP_TWO_STAGE=1
EXTRA="" run f16corr SCLENS_HIP_GRAM_BITS=1
EXTRA="" run both_s3 SCLENS_HIP_TWO_STAGE=1 SCLENS_HIP_GRAM_BITS=1
EXTRA="--streams 1" run both_s1 SCLENS_HIP_TWO_STAGE=1 SCLENS_HIP_GRAM_BITS=1
