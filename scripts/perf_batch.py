"""Tridiagonalisation of nb matrices of order n: solo, concurrent streams, lock-step batched (GPU box).
Usage: perf_batch.py n nb"""
import sys, time, os
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sclens_amd._lib import BatchGroup, Context
from devutil import DevArray, rup

n = int(sys.argv[1]); nb = int(sys.argv[2])
lda = rup(n, 32)
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 256)).astype(np.float32)
A0 = np.zeros((n, lda), np.float32); A0[:, :n] = B @ B.T / 256
ctxs = [Context(0) for _ in range(nb)]
bufs = [(DevArray(c, A0), DevArray(c, A0), DevArray(c, nbytes=8 * n), DevArray(c, nbytes=8 * n), DevArray(c, nbytes=8 * n)) for c in ctxs]

def one(b):
    c = ctxs[b]; src, dA, dd, de, dt = bufs[b]
    c.check(c.lib.sclens_hip_dev_memcpy(c.h, dA.p, src.p, 4 * n * lda, 3))
    c.check(c.lib.sclens_hip_dev_sytrd_f32(c.h, dA.p, n, lda, dd.p, de.p, dt.p))
    c.sync()

group = BatchGroup()
pool = ThreadPoolExecutor(max_workers=nb)
for mode in ("solo", "streams", "batched", "streams", "batched"):
    for c in ctxs:
        c.set_batch(group if mode == "batched" else None)
    one(0)
    t0 = time.perf_counter()
    if mode == "solo":
        one(0)
        cnt = 1
    else:
        if mode == "batched":
            group.expect(nb)
        list(pool.map(one, range(nb)))
        cnt = nb
    dt_ = time.perf_counter() - t0
    print(f"n={n} {mode:8s} members={cnt} wall={dt_*1e3:.1f} ms  per matrix {dt_*1e3/cnt:.1f} ms", flush=True)
