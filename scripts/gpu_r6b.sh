#!/bin/bash
# round 6, call b: the chunked session -- its small-size parity tests, then the spectra of the 1M x 30k matrix (both precisions; no float64
# fixture yet: timings, footprint, pattern builds)
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chunked.py -x -q > $O/pytest_chunked.log 2>&1; echo "pytest rc $?" >> $O/pytest_chunked.log; tail -25 $O/pytest_chunked.log
grep -q "failed\|error" $O/pytest_chunked.log && exit 1
timeout 1500 python scripts/atlas_chunked_run.py --stop-after spectra --precision 1 --out $O/cfg5_spectra_p1.json > $O/cfg5_spectra_p1.log 2>&1; tail -30 $O/cfg5_spectra_p1.log
timeout 900 python scripts/atlas_chunked_run.py --stop-after spectra --precision 0 --out $O/cfg5_spectra_p0.json > $O/cfg5_spectra_p0.log 2>&1; tail -12 $O/cfg5_spectra_p0.log
