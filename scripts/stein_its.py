"""fp64 quality of the inverse-iteration vectors for the context option stein_its = 3 (dstein's 1 + EXTRA) against 2: residual
|T z - lambda z| and orthogonality of the tridiagonal eigenvectors BEFORE any fp32 back-transformation is not observable through
the C ABI (vectors leave the solver as fp32), so: eigenvectors of a symmetric matrix through the whole solver, residual and
orthogonality in float64 on the host. Usage: stein_its.py n"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = Context(0)
rng = np.random.default_rng(1)
K = 2 * n
B = rng.standard_normal((n, K)).astype(np.float32)
B -= B.mean(axis=0, keepdims=True)
# a few planted signals + a cluster of nearly equal eigenvalues
B[:, :5] *= np.array([9, 7, 5, 5.000001, 3], dtype=np.float32)
A = (B @ B.T / K).astype(np.float32)
A = ((A + A.T) / 2).astype(np.float32)
lda = rup(n, 32)
Ap = np.zeros((n, lda), np.float32); Ap[:, :n] = A
for its in ("3", "2", "1"):
    ctx.set_option("stein_its", int(its))
    dA = DevArray(ctx, Ap); dw = DevArray(ctx, nbytes=8 * n); dZ = DevArray(ctx, nbytes=4 * n * lda)
    ctx.set_timing(True); ctx.reset_timing()
    ctx.check(ctx.lib.sclens_hip_dev_eigh_f32(ctx.h, dA.p, n, lda, dw.p, 0, n, dZ.p, lda))
    ctx.sync()
    ms = ctx.timing("stein")[0]
    w = dw.get((n,), np.float64); Z = dZ.get((n, lda), np.float32)[:, :n].astype(np.float64)
    A64 = A.astype(np.float64)
    res = np.abs(Z @ A64 - w[:, None] * Z).max() / np.abs(w).max()
    orth = np.abs(Z @ Z.T - np.eye(n)).max()
    print(f"n={n} good_its={its}: stein {ms:.1f} ms, max residual / |lambda|max = {res:.3e}, max |Z Z' - I| = {orth:.3e}", flush=True)
    for x in (dA, dw, dZ): x.free()
