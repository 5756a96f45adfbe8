#!/bin/bash
# per-phase shader clocks of a group of the image-fed second back-transformation (SCLENS_HIP_Q2_PROF=1), variants 10 / 11 / 14 / 15,
# with and without the products
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4u
mkdir -p $O
ulimit -c 0
SCLENS_HIP_Q2_PROF=1 timeout 600 python scripts/q2_variants.py 30016 15008 10 11 14 15 > $O/q2_prof.log 2>&1; echo "prof rc=$?" >> $O/summary.txt
SCLENS_HIP_Q2_PROF=1 SCLENS_HIP_Q2_DBG=1 timeout 600 python scripts/q2_variants.py 30016 15008 10 14 > $O/q2_prof_noproducts.log 2>&1; echo "prof (no products) rc=$?" >> $O/summary.txt
SCLENS_HIP_Q2_PROF=1 SCLENS_HIP_Q2_DBG=2 timeout 600 python scripts/q2_variants.py 30016 15008 10 14 > $O/q2_prof_nodma.log 2>&1; echo "prof (no DMA) rc=$?" >> $O/summary.txt
SCLENS_HIP_Q2_PROF=1 timeout 600 python scripts/q2_variants.py 30016 30016 10 14 > $O/q2_prof_allvec.log 2>&1; echo "prof (all vectors) rc=$?" >> $O/summary.txt
grep -h -A 7 "^\[sbr_q2\|variant" $O/q2_prof.log | grep -v "^--" | head -120
echo ---- no products; grep -h -A 7 "^\[sbr_q2" $O/q2_prof_noproducts.log | head -40
echo ---- no DMA; grep -h -A 7 "^\[sbr_q2" $O/q2_prof_nodma.log | head -40
echo ---- all vectors; grep -h -A 7 "^\[sbr_q2\|variant" $O/q2_prof_allvec.log | head -60
cat $O/summary.txt
