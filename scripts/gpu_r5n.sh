#!/bin/bash
# round 5: the matching-certificate test alone (its second session needs its own context)
O=gpurun_out/r5n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sclens.py -m gpu -q -s -k "certificate or certified" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
