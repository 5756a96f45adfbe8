#!/bin/bash
# evidence of the round-4 build: the bench line as the driver takes it (fewer steps), kernel stats of the same command, cfg2, the whole
# GPU suite, the atlas slab (slow test)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4n
mkdir -p $O
ulimit -c 0
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_cfg4_final.json 2> $O/bench_cfg4_final.err; echo "bench cfg4 rc=$?" >> $O/summary.txt
tail -c 900 $O/bench_cfg4_final.json; echo
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --strict-fp32 off > /root/repo/$O/bench_cfg4_under_rocprof.json 2> /root/repo/$O/bench_cfg4_under_rocprof.err
cd /root/repo
echo "rocprof rc=$?" >> $O/summary.txt
CSV=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1)
[ -n "$CSV" ] && cp $CSV $O/cfg4_kernel_stats.csv && head -n 14 $O/cfg4_kernel_stats.csv | cut -c1-160
timeout 600 python bench.py --config cfg2 --steps 3 --warmup 1 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?" >> $O/summary.txt
tail -c 400 $O/bench_cfg2.json; echo
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "suite rc=$?" >> $O/summary.txt
tail -n 16 $O/pytest_gpu_full.log
SCLENS_TEST_SLOW=1 SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab_dry_run.json timeout 1500 python -m pytest tests/test_gpu_atlas.py -m gpu -x -q -k "atlas_slab" > $O/pytest_atlas_slab.log 2>&1; echo "atlas slab rc=$?" >> $O/summary.txt
tail -n 4 $O/pytest_atlas_slab.log
cat $O/summary.txt
