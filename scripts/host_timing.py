import sys, time; sys.path.insert(0,'.')
import numpy as np
from sclens_amd import api
from sclens_amd.synth import synth_counts
X = synth_counts(10000,20000,seed=20240428)
ctx = api.default_context()
for rep in range(2):
    t0=time.perf_counter(); Xc = api._csc_f32(X); t1=time.perf_counter()
    d = api.make_draws_native(Xc, seed=5); t2=time.perf_counter()
    s = api.Session(ctx, Xc, d.z_idx1, d.z_idx2); t3=time.perf_counter()
    c2 = api.Context(0); w = s.clone(c2); t4=time.perf_counter()
    w.close(); c2.close(); s.close(); t5=time.perf_counter()
    print(f"csc {t1-t0:.3f} draws {t2-t1:.3f} session {t3-t2:.3f} clone {t4-t3:.3f} close {t5-t4:.3f}")
import cProfile, pstats
d = api.make_draws_native(X, seed=6)
pr = cProfile.Profile(); pr.enable(); r = api.sclens(X, draws=d, ctx=ctx, streams=3); pr.disable()
print("wall", r["wall_s"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
