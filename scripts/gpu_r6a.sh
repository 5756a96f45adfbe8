#!/bin/bash
# round 6, call a: what the GPU box's host offers (cores, memory: the cfg5 test generates a 1M x 30k matrix there), the eigensolve and the Gram
# product alone at precision = 0 (stage table + rocprofv3 kernel statistics: VERDICT r5 item 2), two strict bench steps with stage timing
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
{ free -g; nproc; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max; df -h /tmp /dev/shm .; python3 -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"; } > $O/host.log 2>&1
cat $O/host.log
SCLENS_HIP_OPTIONS=precision=0 LOW_HALF=1 timeout 600 python scripts/perf_eig.py 30016 100000 15008 > $O/perf_eig_strict.log 2>&1; tail -4 $O/perf_eig_strict.log
LOW_HALF=1 timeout 600 python scripts/perf_eig.py 30016 100000 15008 > $O/perf_eig_split.log 2>&1; tail -4 $O/perf_eig_split.log
cd /tmp
SCLENS_HIP_OPTIONS=precision=0 LOW_HALF=1 REPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o eig_strict -- python3 $GRAFT_REPO_ROOT/scripts/perf_eig.py 30016 2048 15008 > $GRAFT_REPO_ROOT/$O/rocprof_eig_strict.log 2>&1
echo "rocprof rc $?"
find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_trace*" -delete; find $GRAFT_REPO_ROOT/$O/prof -name "*.db" -delete
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --steps 2 --warmup 1 --precision 0 --strict-fp32 off --no-cpu-baseline --stage-timing > $O/bench_strict.json 2> $O/bench_strict.err; echo "bench rc $?"
cp bench_detail.json $O/bench_strict_detail.json 2>/dev/null
tail -c 1500 $O/bench_strict.json
grep -i "stage\|ms" $O/bench_strict.err | tail -40
