#!/bin/bash
# blocked triangular inverses in sbr_panel_small: the band-reduction tests, per-phase clocks, stage time
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4l
mkdir -p $O
ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_sbr.py -m gpu -x -q > $O/pytest_sbr.log 2>&1; echo "sbr tests rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_sbr.log
SCLENS_HIP_PANEL_PROF=1 timeout 300 python scripts/perf_sbr.py 30016 > $O/panel_prof.log 2>&1
tail -n 13 $O/panel_prof.log
timeout 300 python scripts/perf_sbr.py 30016 2>&1 | tail -n 2
cat $O/summary.txt
