#!/bin/bash
# round 6, call 7i: __graft_entry__.smoke() and the kernel-level tests on the final commit
O=gpurun_out/r7i; mkdir -p $O
export TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 160 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sbr.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; grep -E "passed|failed|rc" $O/pytest.log | tail -2
