#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 600 python scripts/q2_variants.py 30016 15008 3 7 8 2>&1 | tail -n 7
timeout 300 python scripts/q2_variants.py 4160 512 3 7 8 2>&1 | tail -n 7
