"""The three full-chip split-fp16 products of a search step at the bench's order, each alone (GPU box): the search statistic
(corr_split_kernel, 30 016 x 15 008 x 30 016), the dense Gram product from split operands (gemm_split_kernel, n = 30 016, K given) and the
Gram matrix of a binarised matrix (gram_bits_kernel) -- wall time per launch over REPS launches, for `rocprofv3 --kernel-trace --stats`
and `--pmc TCC_HIT_sum TCC_MISS_sum` passes (is the operand stream served by the L2s?).
Usage: perf_split_products.py [n] [K_gram] [what: corr,gram,bits]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd import api
from sclens_amd._lib import Context
from devutil import DevArray, rup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30016
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
what = (sys.argv[3] if len(sys.argv) > 3 else "corr,gram,bits").split(",")
REPS = int(os.environ.get("REPS", "3"))
ctx = Context(0)  # SCLENS_HIP_OPTIONS (e.g. split_pipe=0) applies
ctx.set_timing(True)
print("split_pipe =", ctx.get_option("split_pipe"), flush=True)
rng = np.random.default_rng(0)


def timed(name, stage, fn, flop_issued, peak_tfs=2500.0):
    for rep in range(REPS + 1):
        ctx.reset_timing()
        fn()
        ctx.sync()
        ms = ctx.timing(stage)[0]
        if rep:
            print(f"{name}: {ms:.2f} ms  {flop_issued / ms / 1e9:.1f} TF/s issued = {flop_issued / ms / 1e9 / peak_tfs:.3f} of the fp16 peak", flush=True)


if "corr" in what:
    p, q = n, n // 2
    blk = rng.standard_normal((n, 1024)).astype(np.float32)
    blk /= np.linalg.norm(blk, axis=0)
    Xh = np.asfortranarray(np.tile(blk, (1, (p + 1023) // 1024))[:, :p])
    Yh = np.asfortranarray(np.roll(Xh[:, :q], 7, axis=0))
    out = np.empty(q, dtype=np.float32)
    import ctypes as C
    from sclens_amd._lib import ptr

    def run():
        ctx.check(ctx.lib.sclens_hip_corr_colmax_f32(ctx.h, ptr(Xh, C.c_float), n, p, ptr(Yh, C.c_float), q, 1, ptr(out, C.c_float)))

    timed(f"corr_split {p} x {q} x {n}", "corr", run, 3.0 * 2.0 * p * q * n)
    del Xh, Yh
if "gram" in what:
    ldb, lda = rup(K, 32), rup(n, 32)
    blk = rng.standard_normal((2048, ldb)).astype(np.float32)
    blk[:, K:] = 0
    dB = DevArray(ctx, nbytes=4 * n * ldb)
    for r0 in range(0, n, 2048):
        rows = min(2048, n - r0)
        ctx.h2d(dB.p + 4 * r0 * ldb, np.roll(blk[:rows], r0 // 2048, axis=1))
    dA = DevArray(ctx, nbytes=4 * n * lda)
    ctx.set_option("gram_bits", 1)  # the split product at any order
    timed(f"gram split n = {n}, K = {K}", "gram", lambda: ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB.p, n, K, ldb, float(K), dA.p, lda)),
          3.0 * n * (n + 1) * K)
    ctx.set_option("gram_bits", -1)
    dB.free()
    dA.free()
if "bits" in what:
    from sclens_amd.synth import synth_counts

    N, M = K, n - 16 if n % 64 == 0 else n
    X = api._csc_f32(synth_counts(N, M, seed=3))
    X.data[:] = 1.0
    for terms in (2, 3):
        ctx.set_option("gram_bits_terms", terms)
        t0 = time.perf_counter()
        for rep in range(REPS):
            ctx.reset_timing()
            api._gram_binary(X, use_bits=True, ctx=ctx)
            ms = ctx.timing("gram")[0]
            print(f"gram_bits terms = {terms}: gram stage {ms:.2f} ms = {terms * 1.0 * M * (M + 1) * N / ms / 1e9 / 2500.0:.3f} of the fp16 peak "
                  f"(whole call incl. upload {time.perf_counter() - t0:.1f} s)", flush=True)
            t0 = time.perf_counter()
ctx.close()
