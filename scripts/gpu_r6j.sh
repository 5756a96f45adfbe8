#!/bin/bash
# round 6, call j: the whole -m gpu suite on the current build (without the cfg5 float64 case: the fixture is still being computed)
O=gpurun_out/r6j; mkdir -p $O
export TMPDIR=/tmp
SCLENS_ATLAS_LOG=$PWD/$O/atlas_slab.json timeout 3300 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
