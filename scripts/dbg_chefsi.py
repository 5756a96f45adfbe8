import sys; sys.path.insert(0,'.')
import numpy as np
from sclens_amd import api
from sclens_amd.synth import synth_counts
X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
d = api.make_draws_native(X, seed=9)
a = api.sclens(X, draws=d, n_perturb=2, max_search_iters=5, partial_eig=True)
print(a["partial_eig"])
