"""Second back-transformation variants on the reflectors of one bulge chase: error against the same product in float64 on the
host (small n) or against variant 3 (large n), and time. Usage: q2_variants.py n m [variants...]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n, m = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3:] or ["3", "5"]
ctx = Context(0)
rng = np.random.default_rng(3)
K = 2 * n if n <= 8192 else 2048
B = rng.standard_normal((n, K)).astype(np.float32)
B -= B.mean(axis=0, keepdims=True)
lda = rup(n, 32)
dB = DevArray(ctx, B if K == 2 * n else np.ascontiguousarray(B)); dA = DevArray(ctx, nbytes=4 * n * lda)
ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB.p, n, K, K, float(K), dA.p, lda))
SB = 64
dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
bd = C.c_int(-1)
ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
ctx.sync()
# orthonormal test vectors (rows): the transformation is orthogonal, so Z Z' = I must survive
Z0 = np.zeros((m, lda), dtype=np.float32)
big = m > 4096  # the host-side QR and Gram products of 15 008 x 30 016 cost minutes: rows of unit length instead, norms only
if big:
    R = rng.standard_normal((m, n)).astype(np.float32)
    Z0[:, :n] = R / np.linalg.norm(R, axis=1, keepdims=True)
else:
    Q, _ = np.linalg.qr(rng.standard_normal((n, min(m, n))))
    Z0[:, :n] = Q.T[:m].astype(np.float32)
outs = {}
ctx.set_timing(True)
for v in variants:
    ctx.set_option("q2_reference", 1 if v == "ref" else 0)
    if v != "ref":
        ctx.set_option("q2_variant", int(v))
    for rep in range(2):
        dZ = DevArray(ctx, Z0)
        ctx.reset_timing()
        ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q2_f32(ctx.h, n, dZ.p, m, lda))
        ctx.sync()
        ms = ctx.timing("sbr_q2")[0]
        out = dZ.get((m, lda), np.float32)[:, :n].astype(np.float64)
        dZ.free()
    outs[v] = out
    nrm = np.linalg.norm(out, axis=1) / np.linalg.norm(Z0[:, :n].astype(np.float64), axis=1)
    orth = "-" if big else f"{np.abs(out @ out.T - np.eye(m)).max():.3e}"
    print(f"n={n} m={m} variant {v}: {ms:.1f} ms, max |Z Z' - I| = {orth}, max | |z| - 1 | = {np.abs(nrm - 1).max():.3e}", flush=True)
base = outs.get("ref", outs[variants[0]])
for v in variants:
    print(f"  variant {v} against {'ref' if 'ref' in outs else variants[0]}: max abs diff {np.abs(outs[v] - base).max():.3e} (entries ~ {1 / np.sqrt(n):.1e})")
