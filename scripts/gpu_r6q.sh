#!/bin/bash
# round 6, call q: the driver's own commands on the final build -- `python -m pytest tests/ -x -q -m gpu` and __graft_entry__.smoke()
O=gpurun_out/r6q; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas timeout 3500 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
