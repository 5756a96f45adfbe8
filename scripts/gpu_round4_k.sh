#!/bin/bash
# per-phase clocks of the 64 x 64 panel algebra (sbr_panel_small), the new failure-injection tests of the row-sharded rounds
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
ulimit -c 0
SCLENS_HIP_PANEL_PROF=1 timeout 300 python scripts/perf_sbr.py 30016 > $O/panel_prof.log 2>&1
tail -n 16 $O/panel_prof.log
timeout 900 python -m pytest tests/test_gpu_atlas.py -m gpu -x -q -k "failing_rank or row_sharded_blocks" > $O/pytest_atlas.log 2>&1; echo "atlas tests rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_atlas.log
cat $O/summary.txt
