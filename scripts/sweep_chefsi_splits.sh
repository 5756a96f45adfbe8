for v in "9 3" "4 1" "16 6" "8 2"; do
  set -- $v
  SCLENS_HIP_CHEFSI_SPLITS=$1 SCLENS_HIP_CHEFSI_SPLITS1=$2 timeout 600 python bench.py --steps 1 --no-cpu-baseline --no-roofline --stage-timing > gpurun_out/che_$1_$2.json 2> gpurun_out/che_$1_$2.err
  echo "S=$1 S1=$2"; grep -o "'chefsi': ([0-9., ]*)" gpurun_out/che_$1_$2.err; grep -o "wall_s [0-9.]*" gpurun_out/che_$1_$2.err; python -c "
import json,sys; d=json.loads(open('gpurun_out/che_$1_$2.json').read().strip().splitlines()[-1]); print(d['sclens_wall_s'], d['observed']['signals'], d['observed']['robust_signals'], d['observed']['synth_s'], d['observed']['ensemble_partial_eig'])"
done
