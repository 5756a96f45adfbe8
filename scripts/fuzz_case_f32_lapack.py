"""A soft case of a fuzz_parity.py sweep on the CPU alone: the oracle's sparsity search in float64, and the same search with the
eigendecompositions in float32 LAPACK (ssyevr on the float32-rounded Gram matrix -- the arithmetic of the reference's own GPU path,
CUDA Float32 syevd, by an independent implementation) and with float64 eigendecompositions of the float32-ROUNDED Gram matrix (input
rounding only). Shows whether a 5e-3 difference in the search statistic between the device path and the float64 oracle is what fp32
arithmetic does to this statistic on that matrix, or an error of the device path.
Usage: fuzz_case_f32_lapack.py <cases> <sweep seed> <case index>[,<case index>...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla
from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

cases, sweep_seed = int(sys.argv[1]), int(sys.argv[2])
only = [int(x) for x in sys.argv[3].split(",")]
rng = np.random.default_rng(sweep_seed)
eigen64 = O.get_eigen


def eigen_f32(Y):
    Y = np.asarray(Y, dtype=np.float32)
    Y = (0.5 * (Y + Y.T)).astype(np.float32)
    L, V = sla.eigh(Y, driver="evr")
    return L.astype(np.float64), V.astype(np.float64)


def eigen_f64_of_rounded(Y):
    return eigen64(np.asarray(Y, dtype=np.float32).astype(np.float64))


for c in range(cases):
    N, M = int(rng.integers(90, 420)), int(rng.integers(90, 420))
    C = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 10 ** 6))
    cent = "median" if rng.random() < 0.25 else "mean"
    streams = int(rng.integers(1, 4))
    mf, ms = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.8, 1.8))
    if c not in only:
        continue
    X = synth_counts(N, M, seed=seed, C=C, marker_frac=mf, marker_sd=ms, min_genes_per_cell=5, min_cells_per_gene=4)
    d = api.make_draws_native(X, seed=seed, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    out = {}
    for name, fn in (("float64", eigen64), ("float32 LAPACK", eigen_f32), ("float64 of the fp32-rounded Gram", eigen_f64_of_rounded)):
        O.get_eigen = fn
        try:
            r = O.sclens(X, od, n_perturb=4, null_tol=O.NULL_DROP, centering=cent, max_search_iters=25)
        finally:
            O.get_eigen = eigen64
        out[name] = r
    print(f"case {c}: N={N} M={M} C={C} seed={seed} {cent}; p_th {out['float64']['p_th']:.5f}")
    ref = np.array([float(t[1]) for _, t in out["float64"]["search_trace"]])
    for name, r in out.items():
        v = np.array([float(t[1]) for _, t in r["search_trace"]])
        m = min(len(v), len(ref))
        print(f"  {name:34s} S={r['n_search']:2d} p_={r['p_']:.3f}  max |d5[1] - float64| = {np.abs(v[:m] - ref[:m]).max():.2e}   d5[1]: {np.round(v, 5)}")
