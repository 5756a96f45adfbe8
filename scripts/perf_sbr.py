"""Timing of the two-stage building blocks (GPU box). Usage: perf_sbr.py n"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n = int(sys.argv[1])
assert n % 64 == 0
ctx = Context(0)
lda = rup(n, 32)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 256)).astype(np.float32)
A0 = np.zeros((n, lda), np.float32)
A0[:, :n] = B @ B.T / 256 + np.eye(n, dtype=np.float32)
src = DevArray(ctx, A0)
dA = DevArray(ctx, A0)
dT = DevArray(ctx, nbytes=4 * (n // 64) * 64 * 64)
bd = C.c_int(0)
dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
ctx.set_timing(True)
for rep in range(3):
    ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, dA.p, src.p, 4 * n * lda, 3))
    ctx.sync()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    ctx.sync()
    t1 = time.perf_counter()
    ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
    ctx.sync()
    t2 = time.perf_counter()
    print(f"n={n} sy2sb {1e3 * (t1 - t0):.1f} ms  sb2st {1e3 * (t2 - t1):.1f} ms  breakdown={bd.value}", flush=True)
