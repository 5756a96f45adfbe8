"""Do concurrent decompositions overlap better when their phases are staggered? (GPU box)
Three host threads, each: Gram (MFMA-bound) + eigh with n/2 vectors (HBM-bound tridiagonalisation, latency-bound bisection /
inverse iteration, MFMA back-transform), `iters` times. Mode A: a barrier before every iteration (the lock-step rounds of
the sparsity search); mode B: free-running after staggered starts."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import Context
from devutil import DevArray, rup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = 2 * n
W, iters = 3, 6
mvec = n // 2 + 1
ldb, lda = rup(K, 32), rup(n, 32)
rng = np.random.default_rng(0)
Bp = np.zeros((n, ldb), np.float32)
Bp[:, :K] = rng.standard_normal((n, K)).astype(np.float32)
ctxs = [Context(0) for _ in range(W)]
bufs = [(DevArray(c, Bp), DevArray(c, nbytes=4 * n * lda), DevArray(c, nbytes=8 * n), DevArray(c, nbytes=4 * mvec * lda)) for c in ctxs]


def one(w):
    c = ctxs[w]
    dB, dA, dw, dZ = bufs[w]
    c.check(c.lib.sclens_hip_dev_gram_f32(c.h, dB.p, n, K, ldb, float(K), dA.p, lda))
    c.check(c.lib.sclens_hip_dev_eigh_f32(c.h, dA.p, n, lda, dw.p, 0, mvec, dZ.p, lda))
    c.sync()


for w in range(W):
    one(w)
for mode, stagger in (("lock-step rounds", None), ("free-running, starts staggered by 0.09 s", 0.09), ("free-running, no stagger", 0.0),
                      ("lock-step rounds", None), ("free-running, starts staggered by 0.09 s", 0.09)):
    bar = threading.Barrier(W)

    def work(w):
        if stagger is not None:
            time.sleep(w * stagger)
        for _ in range(iters):
            if stagger is None:
                bar.wait()
            one(w)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"n={n} {mode:45s} {W}x{iters} decompositions in {dt:.3f} s = {1e3 * dt / (W * iters):.1f} ms each", flush=True)
