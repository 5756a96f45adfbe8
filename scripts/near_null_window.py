"""CPU study (float64): the near-null eigenvalue lambda_s of the binarised and of the perturbed binarised cells > genes matrices against
the session's positivity floor f * sqrt(n) * eps32 * lambda_max, f = 1 (round 6) and f = 8 (before). Harmful = Vr2 drops it and a perturbed
basis keeps it (a ~0 entry enters d_arr, profiles/r06_fuzz_null_floor.md); harmless = both drop, both keep, or only Vr2 keeps.
Usage: near_null_window.py <cases> <seed> <n_lo> <n_hi>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

cases, seed0, n_lo, n_hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(seed0)
harm = {1: 0, 8: 0}
near = 0
for c in range(cases):
    M = int(rng.integers(n_lo, n_hi))
    N = int(M * rng.uniform(1.03, 2.5))
    C = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 10 ** 6))
    mf, ms = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.8, 1.8))
    try:
        X = synth_counts(N, M, seed=seed, C=C, marker_frac=mf, marker_sd=ms, min_genes_per_cell=5, min_cells_per_gene=4)
    except Exception as e:
        print(c, "skip (synth):", e)
        continue
    X_ = O._as_csc_f32(X)
    N, M = X_.shape
    if N <= M:
        continue
    d = api.make_draws_native(X, seed=seed, host_sampler=True)
    coo = X_.tocoo()
    order = np.lexsort((coo.row, coo.col))
    r_, c_, v_ = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order]

    def lam_s(mat):
        w = sla.eigvalsh(O.wishart_matrix(O.logn_scale(O.pre_scale(mat)).T, 1))
        return w[0] / w[-1]

    b = lam_s(sp.csc_matrix((np.ones_like(v_), (r_, c_)), shape=(N, M), dtype=np.float32))
    ps = []
    for it in (0, 6, 12, 18):
        m = int(round((1 - (0.999 - 0.001 * it)) * M * N))
        if m > len(d.z_idx1):
            break
        idx = d.sampler("search", it, len(d.z_idx1), m)
        ps.append(lam_s(O._with_ones(N, M, r_, c_, v_, d.z_idx1, d.z_idx2, idx, binary=True)))
    f1 = 5.96e-8 * np.sqrt(M)
    tags = []
    for f in (1, 8):
        bad = b <= f * f1 and any(p > f * f1 for p in ps)
        harm[f] += bad
        tags.append(f"f={f}: {'HARMFUL' if bad else 'ok'}")
    close = min(abs(np.log(x / f1)) for x in [b] + ps) < np.log(1.3)
    near += close
    print(f"{c} N={N} M={M}: floor(1) {f1:.2e}; binarised {b:.2e}; perturbed {' '.join(f'{p:.2e}' for p in ps)}; {tags[0]}, {tags[1]}"
          f"{'; within 30 % of floor(1)' if close else ''}", flush=True)
print(f"{cases} cases: harmful with f = 1: {harm[1]}, with f = 8: {harm[8]}; within 30 % of floor(1): {near}")
