#!/bin/bash
# round 6, call z: the atlas cases of the suite once more after the resource guards (dry run of one rank's slab, cfg5 spectra against float64)
O=gpurun_out/r6z; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_atlas.py tests/test_gpu_chunked.py -q -k "slab_of_one_rank or one_million" -rs > $O/pytest_atlas.log 2>&1; echo "pytest rc $?" >> $O/pytest_atlas.log; tail -6 $O/pytest_atlas.log; df -h /tmp | tail -1
