"""Does a Gram product (MFMA-bound) overlap with the eigensolver of another stream? (GPU box)
Usage: perf_overlap.py n K reps"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n, K, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c1, c2 = Context(0), Context(0)
lda, ldb = rup(n, 32), rup(K, 32)
rng = np.random.default_rng(0)
blk = rng.standard_normal((2048, ldb)).astype(np.float32)


def fill(ctx):
    d = DevArray(ctx, nbytes=4 * n * ldb)
    for r0 in range(0, n, 2048):
        rows = min(2048, n - r0)
        ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, d.p + 4 * r0 * ldb, blk[:rows].ctypes.data, 4 * rows * ldb, 1))
    return d


B1, B2 = fill(c1), fill(c2)
A1, A2 = DevArray(c1, nbytes=4 * n * lda), DevArray(c2, nbytes=4 * n * lda)
w1 = DevArray(c1, nbytes=8 * n)
Z1 = DevArray(c1, nbytes=4 * (n // 2) * lda)


def eig_loop():
    for _ in range(reps):
        c1.check(c1.lib.sclens_hip_dev_gram_f32(c1.h, B1.p, n, 2048, ldb, 2048.0, A1.p, lda))
        c1.check(c1.lib.sclens_hip_dev_eigh_f32(c1.h, A1.p, n, lda, w1.p, 0, n // 2, Z1.p, lda))
    c1.sync()


def gram_loop():
    for _ in range(reps):
        c2.check(c2.lib.sclens_hip_dev_gram_f32(c2.h, B2.p, n, K, ldb, float(K), A2.p, lda))
    c2.sync()


eig_loop(); gram_loop()  # warm-up
t0 = time.perf_counter(); eig_loop(); t_e = time.perf_counter() - t0
t0 = time.perf_counter(); gram_loop(); t_g = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=eig_loop), threading.Thread(target=gram_loop)]
[t.start() for t in th]; [t.join() for t in th]
t_c = time.perf_counter() - t0
print(f"n={n} K={K} reps={reps}: eig alone {t_e / reps:.3f} s, gram alone {t_g / reps:.3f} s, serial sum {(t_e + t_g) / reps:.3f} s, "
      f"concurrent {t_c / reps:.3f} s per pair")
