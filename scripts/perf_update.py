"""Timing of one rank-K symmetric update C(lower + mirror) += P Q' through the C ABI (GPU box). Usage: perf_update.py n K [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n, K = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lower = int(os.environ.get("LOWER", "1"))
ctx = Context(0)
lda = rup(n, 32)
rng = np.random.default_rng(0)
P = rng.standard_normal((n, K)).astype(np.float32)
dP, dQ = DevArray(ctx, P), DevArray(ctx, P)
dC = DevArray(ctx, nbytes=4 * n * lda)
for rep in range(reps):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.sclens_hip_dev_gemm_f32(ctx.h, dP.p, dQ.p, dC.p, n, n, K, K, K, lda, 1.0, 1.0, 1, lower, None))
    ctx.sync()
    dt = time.perf_counter() - t0
    fl = n * (n + 1) * K if lower else 2.0 * n * n * K
    print(f"n={n} K={K} lower={lower} dbg={os.environ.get('SCLENS_HIP_GEMM_DBG','0')} {1e3*dt:.3f} ms  {fl/dt/1e12:.1f} TF/s (computed half)", flush=True)
