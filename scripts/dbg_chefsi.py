"""context option `debug` = 1: per-sweep trace of the partial eigensolver (Ritz values, residuals, locked pairs) on a small case"""
import sys; sys.path.insert(0, '.')
from sclens_amd import api
from sclens_amd.synth import synth_counts

X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
d = api.make_draws_native(X, seed=9)
ctx = api.default_context()
ctx.set_option("debug", 1)
a = api.sclens(X, draws=d, n_perturb=2, max_search_iters=5, partial_eig=True, ctx=ctx, streams=1)
print(a["partial_eig"])
