#!/bin/bash
# round 6, call u: chunk products formed standalone and added once (instead of accumulated into the running sum): the chunked tests and the
# cfg5 spectra against float64 again
O=gpurun_out/r6u; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_chunked.py -q -s > $O/pytest_chunked.log 2>&1; echo "pytest rc $?" >> $O/pytest_chunked.log; grep -v "^$" $O/pytest_chunked.log | cut -c1-700 | tail -8
