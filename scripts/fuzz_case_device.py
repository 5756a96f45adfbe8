"""One case of a fuzz_parity.py sweep, first search evaluation, taken apart on the GPU box: eigenvalues / eigenvectors of the
binarised matrix and of the perturbed binarised matrix from the device (drop-in get_eigvec on the oracle's scaled matrices, and the
session's own binary_basis / search_step) against the float64 oracle.
Usage: fuzz_case_device.py <cases> <sweep seed> <case index> [evaluation = 0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

cases, sweep_seed, only = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
it_eval = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(sweep_seed)
for c in range(cases):
    N, M = int(rng.integers(90, 420)), int(rng.integers(90, 420))
    C = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 10 ** 6))
    cent = "median" if rng.random() < 0.25 else "mean"
    streams = int(rng.integers(1, 4))
    mf, ms = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.8, 1.8))
    if c == only:
        break
X = synth_counts(N, M, seed=seed, C=C, marker_frac=mf, marker_sd=ms, min_genes_per_cell=5, min_cells_per_gene=4)
d = api.make_draws_native(X, seed=seed, host_sampler=True)
od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
X_ = O._as_csc_f32(X)
N, M = X_.shape
print(f"case {only}: N={N} M={M} C={C} seed={seed} {cent}, evaluation {it_eval}")
coo = X_.tocoo()
order = np.lexsort((coo.row, coo.col))
nz_row, nz_col, nz_val = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order]
z1, z2 = od.z_idx1, od.z_idx2
ls = O.logn_scale
binary = sp.csc_matrix((np.ones_like(nz_val), (nz_row, nz_col)), shape=(N, M), dtype=np.float32)
sb = ls(O.pre_scale(binary))
A = sb.T if N > M else sb
L64, V64 = O.get_eigvec(A, O.NULL_DROP)
n_2 = int(round(V64.shape[1] / 2))
p_ = 0.999 - 0.001 * it_eval
nnzidx = int(round((1 - p_) * M * N))
idx = od.sample("search", it_eval, len(z1), nnzidx)
pert = O._with_ones(N, M, nz_row, nz_col, nz_val, z1, z2, idx, binary=True)
sp_ = ls(O.pre_scale(pert))
A2 = sp_.T if N > M else sp_
L2, V2 = O.get_eigvec(A2, O.NULL_DROP)
lowhalf = slice(V2.shape[1] - n_2 - 1, V2.shape[1])
d64 = np.sort(np.abs(V64.T @ V2[:, lowhalf]).max(axis=0))
print(f"oracle: r = {V64.shape[1]} / {V2.shape[1]}, n_2 = {n_2}, d5 = {np.round(d64[:5], 5)}")
print(f"        smallest eigenvalues / lambda_max: binarised {L64[-3:] / L64[0]}, perturbed {L2[-3:] / L2[0]}")


def compare(name, Lref, Vref, Ld, Vd):
    r = min(Vref.shape[1], Vd.shape[1])
    cosv = np.abs(np.sum(Vref[:, :r] * Vd[:, :r].astype(np.float64), axis=0))
    w = np.argsort(cosv)[:5]
    gaps = np.minimum(np.abs(np.diff(Lref, prepend=np.inf)), np.abs(np.diff(Lref, append=-np.inf)))[:r] / Lref[0]
    print(f"  {name}: r {Vd.shape[1]} vs {Vref.shape[1]}; max |L - L64| / lambda_max = {np.abs(Ld[:r] - Lref[:r]).max() / Lref[0]:.2e}; "
          f"worst |cos| to the float64 vector: {np.round(cosv[w], 5)} at descending indices {w} (gap to the neighbour / lambda_max {gaps[w]}); "
          f"vectors with |cos| < 0.999: {(cosv < 0.999).sum()}")
    G = Vd.astype(np.float64).T @ Vd.astype(np.float64) - np.eye(Vd.shape[1])
    print(f"     orthonormality of the device vectors: max |V'V - I| = {np.abs(G).max():.2e}")


Ld, Vd = api.get_eigvec(np.ascontiguousarray(A, dtype=np.float32))
L2d, V2d = api.get_eigvec(np.ascontiguousarray(A2, dtype=np.float32))
compare("binarised (drop-in)", L64, V64, Ld, Vd)
compare("perturbed (drop-in)", L2, V2, L2d, V2d)
lh = slice(V2d.shape[1] - n_2 - 1, V2d.shape[1])
dd = np.sort(np.abs(Vd.astype(np.float64).T @ V2d[:, lh].astype(np.float64)).max(axis=0))
print(f"statistic from the drop-in's vectors (float64 products): d5 = {np.round(dd[:5], 5)}")
dm = np.sort(np.abs(V64.T @ V2d[:, lh].astype(np.float64)).max(axis=0))
print(f"   ... float64 Vr2 with the device's perturbed vectors: {np.round(dm[:5], 5)};  device Vr2 with float64 perturbed vectors: "
      f"{np.round(np.sort(np.abs(Vd.astype(np.float64).T @ V2[:, lowhalf]).max(axis=0))[:5], 5)}")

ctx = api.default_context()
ses = api.Session(ctx, X_, z1, z2)
try:
    Lb, r = ses.binary_basis()
    Lb = Lb[::-1]
    print(f"session: binary_basis r = {r}; max |L - L64| / lambda_max = {np.abs(Lb[:len(L64)] - L64).max() / L64[0]:.2e}; smallest kept / lambda_max {Lb[r - 3:r] / Lb[0]}")
    d5, r2 = ses.search_step(idx, int(round(r / 2)))
    print(f"session: search_step r = {r2}, d5 = {np.round(d5, 5)}")
finally:
    ses.close()
