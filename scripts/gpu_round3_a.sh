#!/bin/bash
# round-3 GPU check A: kernel tests of the changed pieces, the library RCCL communicator, stage timings at n = 30 016
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_atlas.py -x -q -k "rccl or communicator" > $O/pytest_rccl.log 2>&1; echo "rccl rc=$?" >> $O/summary.txt
LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig_new.log 2>&1
SCLENS_HIP_NO_ACC_INIT=1 LOW_HALF=1 TWO_STAGE=1 timeout 600 python scripts/perf_eig.py 30016 2048 15008 > $O/perf_eig_noaccinit.log 2>&1
timeout 300 python scripts/perf_sbr.py 30016 > $O/perf_sbr.log 2>&1
tail -3 $O/*.log
cat $O/summary.txt
