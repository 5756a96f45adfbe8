import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sclens_amd import api
from sclens_amd._lib import Context
from sclens_amd.synth import synth_counts
X = api._csc_f32(synth_counts(10000, 20000, seed=20240428))
ctx = Context(0)
for rep in range(3):
    t0 = time.perf_counter(); ses = api.Session(ctx, X); t1 = time.perf_counter()
    cs = [Context(0) for _ in range(2)]; t2 = time.perf_counter()
    ws = [ses.clone(c) for c in cs]; t3 = time.perf_counter()
    d = api.make_draws_native(X, seed=5); t4 = time.perf_counter()
    Lr = ws[0].null_spectrum(d.X_r); t5 = time.perf_counter()
    print(f"session {1e3*(t1-t0):.0f} ms, 2 contexts {1e3*(t2-t1):.0f} ms, 2 clones {1e3*(t3-t2):.0f} ms, draws {1e3*(t4-t3):.0f} ms, null_spectrum solo {1e3*(t5-t4):.0f} ms", flush=True)
    for w in ws: w.close()
    for c in cs: c.close()
    ses.close()
