"""The normalisation kernels (csrc/scale.hip, + the value-array kernels around them) of ONE decomposition of each kind at a bench
configuration, for a `rocprofv3 --pmc` pass restricted to them: the data matrix on the counts-only pattern (scLENS.jl:676-696) and one
evaluation of the sparsity search on the union pattern (binarised values + sampled candidates, :735-738). Usage: perf_scale.py [cfg4]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from sclens_amd import api
from sclens_amd.synth import synth_counts

CFG = {"cfg2": (10000, 20000, 1), "cfg3": (50000, 30000, 2), "cfg4": (100000, 30000, 3), "rs20k": (20000, 6000, 0)}
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M, idx = CFG[cfg]
path = os.path.join(os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir()), f"sclens_bench_v2_{cfg}_{N}x{M}_{20240427 + idx}.npz")
if os.path.exists(path):
    z = np.load(path)
    X = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(N, M))
else:
    X = synth_counts(N, M, seed=20240427 + idx)
    np.savez(path, data=X.data, indices=X.indices, indptr=X.indptr)
X = api._csc_f32(X)
ctx = api.Context(0)
ctx.set_timing(True)
ses = api.Session(ctx, X)
try:
    ses.data_spectrum(False)
    t_data = ctx.timing("scale")
    _, r = ses.binary_basis()
    pat = api.Pattern.drawn(ctx, X, 12345)
    ses.set_pattern(pat)
    t0 = ctx.timing("scale")
    ses.search_step_seeded(777, int(round(0.01 * N * M)), int(round(r / 2)))
    t1 = ctx.timing("scale")
    print(f"{cfg}: nnz {X.nnz}; scale stage: data matrix {t_data[0]:.2f} ms, search evaluation {t1[0] - t0[0]:.2f} ms; "
          f"algorithmic bytes 8 nnz + 4 N M = {(8 * X.nnz + 4 * N * M) / 1e9:.2f} GB")
finally:
    ses.close()
