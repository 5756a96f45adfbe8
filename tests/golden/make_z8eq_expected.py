"""Generator of tests/golden/z8eq_expected.json: the only numbers the reference repository holds for the sclens() path are in
its committed example output `out/pca.csv` (`example.jl:17-24` run on `data/Z8eq.csv.gz`, which is NOT in the tree:
`.MISSING_LARGE_BLOBS:27`). Since `Xout1 = nV[:, sig] .* sqrt(nL[sig])'` with unit columns of nV (scLENS.jl:811), the squared
column norms of that file are the robust signals' eigenvalues. This script reads the reference's file (build container only) and
stores those derived numbers; it copies no data rows. Usage: python tests/golden/make_z8eq_expected.py"""
import csv
import json
import os

import numpy as np

SRC = "/root/reference/out/pca.csv"
rows = list(csv.reader(open(SRC)))
hdr, body = rows[0], rows[1:]
A = np.array([[float(x) for x in r[1:]] for r in body])
ev = (A * A).sum(axis=0)
out = {"source": "Mathbiomed/scLENS out/pca.csv (example.jl:17-24 on data/Z8eq.csv.gz, Float32 GPU path)",
       "cells_after_qc": int(A.shape[0]), "robust_signals": int(A.shape[1]),
       "robust_signal_eigenvalues": [round(float(v), 4) for v in ev],
       "note": "the reference's RNG is unseeded: eigenvalues of well separated signals reproduce to a few 1e-3 relative, the "
               "count of robust signals only statistically"}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "z8eq_expected.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(out)
