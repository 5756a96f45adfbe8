#!/bin/bash
# delayed-update threshold of the band reduction after the cheaper panel algebra
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4m
mkdir -p $O
for d in 18432 16384 14336 12288 10240 20480; do
  echo "delay_min $d: $(SCLENS_HIP_SY2SB_DELAY_MIN=$d timeout 300 python scripts/perf_sbr.py 30016 2>&1 | tail -n 2 | tr '\n' ' ')"
done | tee $O/delay_sweep.log
