#!/bin/bash
# round 6, call 7f: the soft cases of the seed-41 sweep taken apart (which eigenvectors differ from float64, drop-in and session)
O=gpurun_out/r7f; mkdir -p $O
export TMPDIR=/tmp
for c in "100 0" "100 9" "121 8"; do timeout 300 python scripts/fuzz_case_device.py 150 41 $c >> $O/cases.log 2>&1; done
grep -v amdgpu $O/cases.log | cut -c1-700
