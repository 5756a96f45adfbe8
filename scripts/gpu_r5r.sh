#!/bin/bash
# round 5, last call: the driver's own commands on the final commit -- `python -m pytest tests/ -x -q -m gpu` and __graft_entry__.smoke()
O=gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
