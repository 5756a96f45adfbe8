#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3cc
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_q2 -- python3 /root/repo/scripts/q2_variants.py 30016 15008 3 7 > /root/repo/$O/pmc_run.log 2>&1
cd /root/repo
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pmc_q2/*/*counter_collection.csv") + glob.glob("/tmp/pmc_q2/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "q2_apply" in k:
        agg[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    wc = c["SQ_WAVE_CYCLES"]
    print(k, {n: round(v / wc, 3) for n, v in c.items() if n != "SQ_WAVE_CYCLES"}, "wave_qcycles", wc, "mfma_busy/(4*wave_qcycles)", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 3))
PY
