"""Gram (lower + mirror) GEMM timing, large-tile kernel vs the 128x128 kernel, in one process (GPU box).
Usage: perf_gemm.py [n K]..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

args = [int(x) for x in sys.argv[1:]] or [10000, 20000, 30000, 20000]
ctx = Context(0)
rng = np.random.default_rng(0)
for n, K in zip(args[0::2], args[1::2]):
    ldb, lda = rup(K, 32), rup(n, 32)
    blk = rng.standard_normal((min(n, 2048), ldb)).astype(np.float32)
    blk[:, K:] = 0
    dB = DevArray(ctx, nbytes=4 * n * ldb)
    for r0 in range(0, n, blk.shape[0]):  # the same random block repeated (row-shifted): full-range random operands
        rows = min(blk.shape[0], n - r0)
        ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, dB.p + 4 * r0 * ldb, blk[:rows].ctypes.data, 4 * rows * ldb, 1))
    dA = DevArray(ctx, nbytes=4 * n * lda)
    res = {}
    for rnd in range(3):
        for small in ("1", ""):
            ctx.set_option("gemm_force", 2 if small else 0)
            ctx.sync()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB.p, n, K, ldb, float(K), dA.p, lda))
            ctx.sync()
            dt = time.perf_counter() - t0
            res.setdefault(small or "big", []).append(dt)
    for k, v in res.items():
        best = min(v)
        print(f"n={n} K={K} kernel={'128x128' if k == '1' else '256x256'}: best {best * 1e3:.2f} ms  "
              f"{n * (n + 1) * K / best / 1e12:.1f} TF/s on the lower half ({2 * n * n * K / best / 1e12:.1f} full-count)  all={['%.1f' % (x * 1e3) for x in v]}",
              flush=True)
    if n <= 4096:
        A = dA.get((n, lda), np.float32)[:, :n]
        Bh = np.vstack([blk] * ((n + blk.shape[0] - 1) // blk.shape[0]))[:n, :K].astype(np.float64)
        ref = Bh @ Bh.T / K
        print("max abs err vs fp64:", np.abs(A - ref).max(), "sym:", np.array_equal(A, A.T))
    dB.free()
    dA.free()
