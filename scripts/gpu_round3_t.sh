#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_sbr -- python3 /root/repo/scripts/perf_sbr.py 30016 > /root/repo/$O/perf_sbr_under_rocprof.log 2>&1
cd /root/repo
F=$(find /tmp/trace_sbr -name "*kernel_trace.csv" | head -1)
echo "trace file: $F" >> $O/summary.txt
python3 scripts/trace_sy2sb.py $F $O/sy2sb_trace_summary.json > $O/trace_summary_stdout.log 2>&1
tail -c 600 $O/trace_summary_stdout.log
