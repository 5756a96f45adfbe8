#!/bin/bash
# round 6, call 7g: sessions count positive eigenvalues with the floor at 1 sqrt(n) eps32 lambda_max and the structural zero by count:
# the two sweeps that had soft cases, then the session / golden / property / atlas / chunked tests
O=gpurun_out/r7g; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python scripts/fuzz_parity.py 150 41 certified > $O/fuzz_150_seed41.log 2>&1; tail -1 $O/fuzz_150_seed41.log | cut -c1-300
timeout 400 python scripts/fuzz_parity.py 150 51 certified > $O/fuzz_150_seed51.log 2>&1; tail -1 $O/fuzz_150_seed51.log | cut -c1-300
timeout 400 python scripts/fuzz_parity.py 150 21 certified > $O/fuzz_150_seed21.log 2>&1; tail -1 $O/fuzz_150_seed21.log | cut -c1-300
timeout 560 python -m pytest tests/test_gpu_sclens.py tests/test_gpu_golden.py tests/test_gpu_z8eq.py tests/test_gpu_large.py tests/test_gpu_properties.py tests/test_gpu_atlas.py tests/test_gpu_multirank.py tests/test_gpu_chunked.py -x -q -k "not one_million" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
