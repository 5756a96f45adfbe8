"""A/B of SURVEY 8f-1 at a bench configuration: the Gram matrix of the scaled DATA matrix (count-valued, scLENS.jl:676-696 + :332-361) and
of the scaled BINARISED matrix, from the sparse structure (csrc/gram_sparse.hip) against the dense products (fp32 MFMA with precision = 0,
split-fp16 with precision = 1; the binarised matrix also against the co-occurrence product of gram_bits.hip). Times are the library's own
HIP-event stage timers ("scale" = statistics (+ dense write), "gram" = the product). Usage: perf_gram_sparse.py [cfg4] [reps]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from sclens_amd import api
from sclens_amd.synth import synth_counts

CFG = {"cfg2t": (20000, 10000, 1), "cfg3": (50000, 30000, 2), "cfg4": (100000, 30000, 3), "rs20k": (20000, 6000, 0)}
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, M, idx = CFG[cfg]
path = os.path.join(os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir()), f"sclens_bench_v2_{cfg}_{N}x{M}_{20240427 + idx}.npz")
if os.path.exists(path):
    z = np.load(path)
    X = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(N, M))
else:
    X = synth_counts(N, M, seed=20240427 + idx)
    np.savez(path, data=X.data, indices=X.indices, indptr=X.indptr)
X = api._csc_f32(X)
print(f"{cfg}: {N} x {M}, nnz {X.nnz}, sum_i r_i^2 / 2 = {0.5 * float((np.bincount(X.indices, minlength=N).astype(np.float64) ** 2).sum()):.3e} multiply-adds "
      f"(dense lower half: {0.5 * M * (M + 1) * N:.3e})", flush=True)
ctx = api.Context(0)
ctx.set_timing(True)


def run(label, mode, precision, binary=False, bits=None):
    ctx.set_option("precision", precision)
    out = None
    for r in range(reps):
        ctx.reset_timing()
        t0 = time.perf_counter()
        if bits is not None:
            out = api._gram_binary(X, bool(bits), divisor=float(N), ctx=ctx)
        else:
            out = api._gram_counts(X, mode, f32path=not (mode >= 0 and not binary and False), binary=binary, divisor=float(N if binary else M), ctx=ctx)
        wall = time.perf_counter() - t0
        sc, gr = ctx.timing("scale"), ctx.timing("gram")
    print(f"  {label:58s} scale {sc[0]:8.2f} ms  gram {gr[0]:8.2f} ms  (call incl. upload / pattern / download {wall:.2f} s)", flush=True)
    return out.astype(np.float64)


print("data matrix (count-valued):")
d0 = run("dense, fp32 MFMA (precision 0)", 0, 0)
d1 = run("dense, split fp16 (precision 1)", 0, 1)
s0 = run("sparse structure (gram_sparse)", 1, 0)
sc = np.abs(d0).max()
print(f"  max |sparse - dense fp32| / largest entry {np.abs(s0 - d0).max() / sc:.2e}; |split - dense fp32| {np.abs(d1 - d0).max() / sc:.2e}; "
      f"|sparse - split| {np.abs(s0 - d1).max() / sc:.2e}; symmetric {np.array_equal(s0, s0.T)}")


def where(name, a, b):
    e = np.abs(a - b)
    j, k = np.unravel_index(int(np.argmax(e)), e.shape)
    print(f"  {name}: largest difference at ({j}, {k}): {a[j, k]:.9g} vs {b[j, k]:.9g}; entries off by more than 1e-5 of the largest entry: "
          f"{int((e > 1e-5 * sc).sum())} of {e.size}; diagonal only: {float(np.abs(np.diag(a) - np.diag(b)).max() / sc):.2e}; "
          f"largest entry {sc:.6g} at {np.unravel_index(int(np.argmax(np.abs(b))), b.shape)}")


where("dense fp32 vs sparse", d0, s0)
where("split vs sparse", d1, s0)
ctx.set_option("gemm_force", 2)
d2 = run("dense, fp32 MFMA, 128 x 128 kernel (gemm_force = 2)", 0, 0)
ctx.set_option("gemm_force", 0)
where("dense fp32 (128 x 128 kernel) vs sparse", d2, s0)
print("binarised matrix:")
b0 = run("dense, fp32 MFMA (precision 0)", 0, 0, binary=True)
b2 = run("co-occurrence product, 33-bit weights (gram_bits, precision 0)", 0, 0, bits=1)
b3 = run("co-occurrence product, 22-bit weights (gram_bits, precision 1)", 0, 1, bits=1)
bs = run("sparse structure (gram_sparse)", 1, 0, binary=True)
sc = np.abs(b0).max()
print(f"  max |sparse - dense fp32| / largest entry {np.abs(bs - b0).max() / sc:.2e}; |bits33 - dense| {np.abs(b2 - b0).max() / sc:.2e}; "
      f"|bits22 - dense| {np.abs(b3 - b0).max() / sc:.2e}")
