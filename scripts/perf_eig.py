"""Stage timings of the device eigensolver + Gram at a given size (GPU box). Usage: perf_eig.py n [K] [mvec]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
mvec = int(sys.argv[3]) if len(sys.argv) > 3 else n
ctx = Context(0)
if os.environ.get("TWO_STAGE"):
    ctx.set_option("two_stage", 1)
lo_first = bool(os.environ.get("LOW_HALF"))
rng = np.random.default_rng(0)
ldb = rup(K, 32); lda = rup(n, 32)
Bp = np.zeros((n, ldb), np.float32)
for r0 in range(0, n, 2048):  # float32 draws in row chunks: 30 016 x 100 000 must not pass through a float64 temporary
    Bp[r0:r0 + 2048, :K] = rng.standard_normal((min(2048, n - r0), K), dtype=np.float32)
Bp[:, :K] -= Bp[:, :K].mean(axis=0, keepdims=True)
B = Bp[:, :K]
dB = DevArray(ctx, Bp); dA = DevArray(ctx, nbytes=4 * n * lda); dw = DevArray(ctx, nbytes=8 * n)
dZ = DevArray(ctx, nbytes=4 * max(mvec, 1) * lda)
ctx.set_timing(True)
for rep in range(int(os.environ.get("REPS", "2"))):  # REPS=1: counter passes (every profiled dispatch is serialised)
    ctx.reset_timing()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB.p, n, K, ldb, float(K), dA.p, lda))
    ctx.check(ctx.lib.sclens_hip_dev_eigh_f32(ctx.h, dA.p, n, lda, dw.p, 0 if lo_first else n - mvec, mvec if lo_first else n, dZ.p, lda))
    ctx.sync()
    wall = time.perf_counter() - t0
    out = {s: ctx.timing(s) for s in ("gram", "sytrd", "sy2sb", "sb2st", "stebz", "stein", "ormtr", "sbr_q2", "sbr_q1")}
    print(f"n={n} K={K} mvec={mvec} rep={rep} wall={wall:.3f}s", {k: round(v[0], 2) for k, v in out.items()})
w = dw.get((n,), np.float64)
if os.environ.get("PRINT_HASH"):  # compare variants of a back-transformation bit by bit
    import zlib
    Zh = dZ.get((mvec, lda), np.float32)
    print("eigenvector block: crc32", zlib.crc32(Zh.tobytes()), "sum |z|", float(np.abs(Zh[:, :n]).sum(dtype=np.float64)),
          "max |row norm - 1|", float(np.abs(np.sqrt((Zh[:, :n].astype(np.float64) ** 2).sum(axis=1)) - 1).max()))
print("gram TF/s (full 2n^2K):", 2 * n * n * K / (out["gram"][0] * 1e-3) / 1e12)
if out["sytrd"][0] > 0:
    print("sytrd algorithmic GB/s (4/3 n^3 * 4B / 2... full-matrix symv reads 4/3 n^3 B):", (4 / 3 * n**3) / (out["sytrd"][0] * 1e-3) / 1e9)
print("null eig / max eig:", w[0] / w[-1], " second:", w[1] / w[-1])
if n <= 4000:
    ref = np.linalg.eigvalsh((B.astype(np.float64) @ B.T.astype(np.float64)) / K)
    print("max eig err rel:", np.abs(w - ref).max() / ref.max())
    Z = dZ.get((mvec, lda), np.float32)[:, :n].astype(np.float64)
    print("orth:", np.abs(Z @ Z.T - np.eye(mvec)).max())
