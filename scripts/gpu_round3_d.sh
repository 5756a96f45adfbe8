#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> $O/summary.txt
LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 2>&1 | grep "rep=" > $O/perf_eig.log
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_sbr -- python3 /root/repo/scripts/perf_sbr.py 30016 > /root/repo/$O/perf_sbr_under_rocprof.log 2>&1
cd /root/repo
F=$(find /tmp/trace_sbr -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_sy2sb.py $F $O/sy2sb_trace_summary.json > /dev/null 2>&1
cat $O/perf_eig.log $O/summary.txt; tail -3 $O/pytest_kernels.log
