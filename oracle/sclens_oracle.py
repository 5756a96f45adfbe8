"""CPU oracle for the scLENS `sclens()` hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a float64 NumPy/SciPy restatement of the reference algorithm in
`/root/reference/src/scLENS.jl` (Mathbiomed/scLENS v2.0.1), CPU branch
(`device_="cpu"`).  It exists so that the HIP path in `sclens_amd/` can be
checked; nothing under `sclens_amd/` may import it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it.

PARITY UNPINNED: the reference is Julia, there is no `julia` binary in the build
image, the reference ships no tests / golden vectors, and the input of its only
committed numeric artefact (`out/pca.csv` <- `data/Z8eq.csv.gz`) is missing from
the mount (`.MISSING_LARGE_BLOBS:27`).  The arithmetic of the reference lives in
third-party dependencies that are not under /root/reference:
  * Julia stdlib LinearAlgebra 1.11 -> OpenBLAS_jll 0.3.27+1 (dsyrk/dgemm, LAPACK dsyevr)
    (Manifest.toml:1060-1063, :1263-1266) -- restated here with SciPy's LAPACK `dsyevr`
    (`scipy.linalg.eigh(driver="evr")`) and NumPy's BLAS;
  * StatsBase 0.34.3 (`sample`, `quantile`, `iqr`), Distributions 0.25.113 (`Normal`),
    NaNStatistics 0.6.42 (`histcounts`, `nanmaximum`), stdlib Random / SparseArrays /
    Statistics -- restated from their published semantics.
Every random draw the reference takes from Julia's unseeded global RNG is an
*injected* argument here, so that the oracle and
the HIP path can be fed identical draws.

Each function cites the reference lines it follows as `scLENS.jl:<lines>`.
Matrices are cells x genes (N x M), like the reference's DataFrame minus `:cell`.
Indices are 0-based here (the reference is 1-based).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# Normalisation (scLENS.jl:596-608, :650-658, :676-696, :291-305)
# --------------------------------------------------------------------------------------


def _as_csc_f32(X) -> sp.csc_matrix:
    """df2sparr (scLENS.jl:90-120): SparseMatrixCSC{Float32,UInt32}; explicit zeros dropped."""
    X = sp.csc_matrix(X, dtype=np.float32)
    X.eliminate_zeros()
    X.sort_indices()
    return X


def pre_scale(X: sp.csc_matrix) -> sp.csc_matrix:
    """`pre_scale = x -> log1p.(proj_l(x))` (scLENS.jl:650, :607).

    proj_l on a sparse Float32 matrix is spdiagm(1 ./ sum(x,dims=2)) * x, all Float32
    (SURVEY Appendix A4/A5); log1p touches stored entries only.
    """
    X = _as_csc_f32(X)
    rs = np.asarray(X.sum(axis=1)).ravel().astype(np.float32)
    inv = (np.float32(1.0) / rs).astype(np.float32)
    Y = X.copy()
    Y.data = np.log1p((inv[Y.indices] * Y.data).astype(np.float32)).astype(np.float32)
    return Y


def _sparse_col_mean_std(Y: sp.csc_matrix, dtype) -> Tuple[np.ndarray, np.ndarray]:
    """mean(X,dims=1) and std(X,dims=1) (corrected, zeros count) (scLENS.jl:597, :599, :682-683)."""
    N = Y.shape[0]
    d = Y.data.astype(np.float64)
    s1 = np.add.reduceat(np.append(d, 0.0), Y.indptr[:-1])
    s1[np.diff(Y.indptr) == 0] = 0.0
    mean = s1 / N
    # two-pass variance, zeros included
    dev = d - np.repeat(mean, np.diff(Y.indptr))
    s2 = np.add.reduceat(np.append(dev * dev, 0.0), Y.indptr[:-1])
    s2[np.diff(Y.indptr) == 0] = 0.0
    nz = np.diff(Y.indptr)
    s2 = s2 + (N - nz) * mean * mean
    std = np.sqrt(s2 / (N - 1))
    return mean.astype(dtype), std.astype(dtype)


def zscore_with_l2(Y: sp.csc_matrix, f32_std: bool = True) -> Tuple[np.ndarray, dict]:
    """zscore_with_l2 (scLENS.jl:596-605) and its inline Float64 twin (:682-693).

    `f32_std=True` reproduces the closure path: `std(X,dims=1)` of a Float32 matrix is
    Float32, then `1. ./ std_` promotes everything downstream to Float64 (Appendix A4).
    Returns the dense N x M Float64 matrix before the final centring, plus the
    intermediate vectors the reference stores in `rec_vals`.
    """
    N, M = Y.shape
    mean, std = _sparse_col_mean_std(Y, np.float32 if f32_std else np.float64)
    inv_std = 1.0 / std.astype(np.float64)
    Z = Y.astype(np.float64).multiply(inv_std[None, :]).tocsc()  # X * spdiagm(1 ./ std_)
    mu = np.asarray(Z.sum(axis=0)).ravel() / N  # mean(X_norm, dims=1)
    l2X2 = np.asarray(Z.multiply(Z).sum(axis=1)).ravel()  # l2X.^2
    l2mu = np.linalg.norm(mu)
    l2norm = np.sqrt(l2X2 - 2.0 * (Z @ mu) + l2mu * l2mu)  # scLENS.jl:603 / :690
    dense = (Z.toarray() - mu[None, :]) / (l2norm / l2norm.mean())[:, None]
    rec = {"mat2_mean": mean.astype(np.float64), "mat2_std": std.astype(np.float64), "norm_tgc": l2norm}
    return dense, rec


def logn_scale(Y: sp.csc_matrix) -> np.ndarray:
    """`logn_scale` for centering="mean": scaled_gdata(zscore_with_l2(x), "cent") (scLENS.jl:652, :300-305)."""
    dense, _ = zscore_with_l2(Y, f32_std=True)
    return dense - dense.mean(axis=0, keepdims=True)


def scaled_gdata_median(Xd: np.ndarray) -> np.ndarray:
    """scaled_gdata(X, position_="median") on a dense Float32 matrix (scLENS.jl:291-298 dense branch, :306, :328):
    per gene (x - median) / std, median and corrected std over all N cells, everything Float32."""
    Xd = np.asarray(Xd, dtype=np.float32)
    med = np.median(Xd, axis=0).astype(np.float32)  # mapslices(median, X, dims=1); even N: middle(a, b)
    sd = Xd.std(axis=0, ddof=1, dtype=np.float32)
    return (Xd - med[None, :]) / sd[None, :]


def norm_l(x: np.ndarray) -> np.ndarray:
    """norm_l, dense branch (scLENS.jl:608): every row scaled to the mean row L2 norm."""
    l2 = np.sqrt((x * x).sum(axis=1, dtype=x.dtype))
    return x / l2[:, None] * l2.mean(dtype=x.dtype)


def logn_scale_median(Y: sp.csc_matrix) -> np.ndarray:
    """`logn_scale` for centering="median" (scLENS.jl:653-654): norm_l(scaled_gdata(Matrix{Float32}(x), "median")).
    The reference stays Float32 from here through LAPACK (ssyevr); the oracle returns the Float32-scaled matrix
    promoted to float64 so that the Gram matrix and the eigendecomposition are the accurate ones."""
    dense = np.asarray(Y.todense(), dtype=np.float32)
    return norm_l(scaled_gdata_median(dense)).astype(np.float64)


def logn_scale_other(Y: sp.csc_matrix) -> np.ndarray:
    """`logn_scale` for any other `centering` string (scLENS.jl:655-657, after the printed warning):
    scaled_gdata(norm_l(scaled_gdata(Matrix{Float32}(x), "mean")), "cent") -- per gene (x - mean) / std, rows scaled to the mean
    row norm, columns centred, all on a dense Float32 copy. The same function of x as the "mean" branch (zscore_with_l2 + "cent"),
    which evaluates it through sparse Float64 identities; promoted to float64 on return like logn_scale_median."""
    Xd = np.asarray(Y.todense(), dtype=np.float32)
    mean = Xd.mean(axis=0, dtype=np.float32)
    sd = Xd.std(axis=0, ddof=1, dtype=np.float32)
    z = norm_l((Xd - mean[None, :]) / sd[None, :])
    return (z - z.mean(axis=0, dtype=np.float32)[None, :]).astype(np.float64)


def scale_main(X: sp.csc_matrix) -> Tuple[np.ndarray, dict]:
    """Inline Float64 normalisation of the data matrix with rec_vals (scLENS.jl:676-696)."""
    X = _as_csc_f32(X)
    tgc = np.asarray(X.astype(np.float64).sum(axis=1)).ravel()  # Vector{Float64}(sum(X_,dims=2))
    mat2 = X.astype(np.float64).tocsc()
    mat2.data = np.log1p(mat2.data / tgc[mat2.indices])
    dense, rec = zscore_with_l2(mat2, f32_std=False)
    cent = dense.mean(axis=0)
    rec["TGC"] = tgc
    rec["cent_"] = cent
    return dense - cent[None, :], rec


# --------------------------------------------------------------------------------------
# Device-dispatch layer, CPU branch (scLENS.jl:332-387)
# --------------------------------------------------------------------------------------


def wishart_matrix(X: np.ndarray, dims: int) -> np.ndarray:
    """_wishart_matrix, CPU branch (scLENS.jl:345-359). Divides by size(X,2) in both cases (Appendix A8)."""
    if dims == 2:
        return (X.T @ X) / X.shape[1]
    if dims == 1:
        return (X @ X.T) / X.shape[1]
    raise ValueError("dims must be 1 or 2")


def corr_mat(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """corr_mat, CPU branch (scLENS.jl:370-372)."""
    return X.T @ Y


def get_eigen(Y: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """_get_eigen, CPU branch (scLENS.jl:383-386): LAPACK syevr, ascending (Appendix A9)."""
    Y = 0.5 * (Y + Y.T) if not np.array_equal(Y, Y.T) else Y
    L, V = sla.eigh(Y, driver="evr")
    return L, V


# --------------------------------------------------------------------------------------
# Marchenko-Pastur / Tracy-Widom statistics (scLENS.jl:390-487)
# --------------------------------------------------------------------------------------


def mp_parameters(L: np.ndarray) -> dict:
    """_mp_parameters (scLENS.jl:390-408). mean of an empty vector is NaN (Appendix A12)."""
    L = np.asarray(L, dtype=np.float64)
    if L.size == 0:
        m1 = m2 = float("nan")
    else:
        m1 = float(np.mean(L))
        m2 = float(np.mean(L * L))
    gamma = m2 / (m1 * m1) - 1.0 if m1 == m1 and m1 != 0 else float("nan")
    sg = math.sqrt(gamma) if gamma == gamma and gamma >= 0 else float("nan")
    return {
        "moment_1": m1,
        "moment_2": m2,
        "gamma": gamma,
        "b_plus": m1 * (1 + sg) ** 2,
        "b_minus": m1 * (1 - sg) ** 2,
        "s": m1,
        "peak": m1 * (1.0 - gamma) ** 2.0 / (1.0 + gamma) if gamma == gamma else float("nan"),
        "sigma": m2,
    }


def marchenko_pastur(x: float, y: dict) -> float:
    """_marchenko_pastur (scLENS.jl:411-418)."""
    if y["b_minus"] < x < y["b_plus"]:
        return math.sqrt((y["b_plus"] - x) * (x - y["b_minus"])) / (2 * y["s"] * math.pi * y["gamma"] * x)
    return 0.0


def mp_pdf(x: float, L: np.ndarray) -> float:
    """_mp_pdf (scLENS.jl:420-422)."""
    return marchenko_pastur(x, mp_parameters(L))


def mp_calculation(L, Lr, eta=1.0, eps=1e-6, max_iter=10000):
    """_mp_calculation (scLENS.jl:424-459). Returns (L_mp, b_plus, b_minus)."""
    L = np.asarray(L, dtype=np.float64)
    mpp = mp_parameters(Lr)
    b_plus, b_minus = mpp["b_plus"], mpp["b_minus"]
    new = mp_parameters(L[(b_minus < L) & (L < b_plus)])
    new_b_plus, new_b_minus = new["b_plus"], new["b_minus"]
    it = 0
    while True:
        loss = (1 - new_b_plus / b_plus) ** 2
        it += 1
        if loss <= eps:
            break
        if it == max_iter:
            break
        gradient = new_b_plus - b_plus
        new_b_plus = b_plus + eta * gradient
        sel = L[(new_b_minus < L) & (L < new_b_plus)]
        b_plus, b_minus = new_b_plus, new_b_minus
        up = mp_parameters(sel)
        new_b_plus, new_b_minus = up["b_plus"], up["b_minus"]
    return L[(new_b_minus < L) & (L < new_b_plus)], new_b_plus, new_b_minus


def tw(L, L_mp):
    """_tw (scLENS.jl:461-467): p uses length of ALL eigenvalues (Appendix A13)."""
    gamma = mp_parameters(L_mp)["gamma"]
    p = len(L) / gamma
    sigma = 1 / p ** (2 / 3) * gamma ** (5 / 6) * (1 + math.sqrt(gamma)) ** (4 / 3)
    lambda_c = float(np.mean(L_mp)) * (1 + math.sqrt(gamma)) ** 2 + sigma
    return lambda_c, gamma, p, sigma


def mp_check(test_L, p_val=0.05) -> dict:
    """mp_check (scLENS.jl:469-487). 100 edges / 99 half-open bins (Appendix A14)."""
    test_L = np.asarray(test_L, dtype=np.float64)
    lo, hi = test_L.min() - 1, test_L.max() + 1
    edges = lo + (hi - lo) * np.arange(100) / 99.0  # LinRange(lo, hi, 100)
    # NaNStatistics.histcounts: bin index = floor((x - first) / step), half-open, last edge excluded
    idx = np.floor((test_L - lo) / ((hi - lo) / 99.0)).astype(np.int64)
    idx = idx[(idx >= 0) & (idx < 99)]
    count = np.bincount(idx, minlength=99).astype(np.float64)
    cdf_arr = np.cumsum(count / count.sum())
    centres = (edges[1:] + edges[:-1]) / 2
    par = mp_parameters(test_L)
    pdf = np.array([marchenko_pastur(float(x), par) for x in centres])
    c2 = np.cumsum(pdf)
    nc2 = c2 / c2.max()
    D = float(np.max(np.abs(cdf_arr - nc2)))
    c_alpha = math.sqrt(-0.5 * math.log(p_val))
    m = n = 99
    return {"ks_static": D, "pass": bool(D <= c_alpha * math.sqrt((m + n) / m / n))}


# --------------------------------------------------------------------------------------
# Spectral wrappers (scLENS.jl:489-594)
# --------------------------------------------------------------------------------------


def _normalize_cols(A: np.ndarray) -> np.ndarray:
    """mapslices(s -> s/norm(s), A, dims=1) (Appendix A16)."""
    return A / np.linalg.norm(A, axis=0, keepdims=True)


# Positive filter `L .> 0` (scLENS.jl:495, :515). Reference oddity (SURVEY 8a defect 6): for N <= M the centred
# matrix has one structurally zero eigenvalue whose computed value is +-1e-17 by rounding luck, so the reference
# keeps or drops that eigenvector at random (and with it one ~0 entry of d_arr at :742). NULL_TOL = 0 restates the
# reference literally; the HIP path (and the parity tests, via null_tol=NULL_DROP) always drop it.
NULL_DROP = 1e-9


def get_eigvec(X: np.ndarray, null_tol: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """get_eigvec (scLENS.jl:489-524), CPU branch; the catch expression (:507) is used for N > M.
    `null_tol`: eigenvalues <= null_tol * max(L) count as non-positive (0 = the reference's `L .> 0`)."""
    N, M = X.shape
    if N > M:
        L, V = get_eigen(wishart_matrix(X, 2))
    else:
        L, V = get_eigen(wishart_matrix(X, 1))
    pos = L > (null_tol * L.max() if null_tol > 0 else 0.0)
    L, V = L[pos], V[:, pos]
    order = np.argsort(-L, kind="stable")  # sortperm(L, rev=true) (Appendix A15)
    nL, nVs = L[order], V[:, order]
    if N > M:
        mul_X = nVs * np.sqrt(1.0 / nL)[None, :]
        return nL, _normalize_cols(X @ mul_X)
    return nL, nVs


def get_sigev(X: np.ndarray, Xr: np.ndarray):
    """get_sigev (scLENS.jl:526-594), CPU branch.

    Reference defect 1: for N > M the CPU branch calls `cu(...)` unguarded
    (:558-559); the obvious CPU equivalent `X*mul_X` (as in :507) is used. The dead noise
    eigenvectors (:545, :557, :560-564, :584, :592) are not formed; `noiseL` is returned.
    """
    n, m = X.shape
    dims = 2 if n > m else 1
    L, V = get_eigen(wishart_matrix(X, dims))
    Lr, _ = get_eigen(wishart_matrix(Xr, dims))
    L_mp, _, b_min = mp_calculation(L, Lr[:-1])  # Lr[1:end-1] drops the largest (Appendix A10)
    lambda_c = tw(L, L_mp)[0]
    sel = L > lambda_c
    sel_L, sel_V = L[sel], V[:, sel]
    noiseL = L[(b_min <= L) & (L <= lambda_c)]
    o = np.argsort(-sel_L, kind="stable")
    nL, nVs = sel_L[o], sel_V[:, o]
    noiseL = noiseL[np.argsort(-noiseL, kind="stable")]
    if n > m:
        mul_X = nVs * np.sqrt(1.0 / nL)[None, :]
        nVs = _normalize_cols(X @ mul_X)
    return nL, nVs, L, L_mp, lambda_c, noiseL


# --------------------------------------------------------------------------------------
# Null model and random draws (scLENS.jl:239-289, :668-673) -- injected RNG
# --------------------------------------------------------------------------------------


def random_nz(X: sp.csc_matrix, rng: np.random.Generator) -> sp.csc_matrix:
    """random_nz(pre_df, rmix=true) (scLENS.jl:261-289 -> :239-248), *intent* restated.

    Reference defect 4: `_random_matrix(dims=1)` pairs hash-ordered per-column
    samples with sorted column indices and lets `sparse` infer the size; the intent is: shuffle
    the stored values globally (:275), then give each column the same number of entries at
    rows drawn uniformly without replacement (:247), explicit N x M.
    """
    X = _as_csc_f32(X)
    N, M = X.shape
    vals = rng.permutation(X.data)
    indptr = X.indptr.copy()
    rows = np.empty_like(X.indices)
    for j in range(M):
        c = indptr[j + 1] - indptr[j]
        if c:
            rows[indptr[j] : indptr[j + 1]] = np.sort(rng.choice(N, size=c, replace=False))
    return sp.csc_matrix((vals, rows, indptr), shape=(N, M), dtype=np.float32)


def zero_candidates(X: sp.csc_matrix, rng: np.random.Generator) -> Tuple[np.ndarray, np.ndarray]:
    """Zero-candidate list (scLENS.jl:668-673): nnz uniform (i,j) draws, minus the stored set,
    de-duplicated in first-occurrence order (`setdiff`). Returns 0-based (z_idx1, z_idx2)."""
    X = _as_csc_f32(X)
    N, M = X.shape
    nnz = X.nnz
    i = rng.integers(0, N, size=nnz, dtype=np.int64)
    j = rng.integers(0, M, size=nnz, dtype=np.int64)
    key = i + j * N
    _, first = np.unique(key, return_index=True)
    first.sort()
    key = key[first]
    coo = X.tocoo()
    nzkey = coo.row.astype(np.int64) + coo.col.astype(np.int64) * N
    keep = ~np.isin(key, nzkey)
    key = key[keep]
    return (key % N).astype(np.uint32), (key // N).astype(np.uint32)


def noise_baseline(n: int, rng: np.random.Generator, trials: int = 5000) -> float:
    """p_th (scLENS.jl:709-712): mean over 5000 trials of max |N(0, 1/n)| over n draws."""
    acc = 0.0
    sd = math.sqrt(1.0 / n)
    for _ in range(trials):
        acc += float(np.max(np.abs(rng.standard_normal(n)))) * sd
    return acc / trials


# --------------------------------------------------------------------------------------
# Driver (scLENS.jl:649-832)
# --------------------------------------------------------------------------------------


@dataclass
class Draws:
    """All random draws of one `sclens()` call. 0-based indices."""

    z_idx1: np.ndarray  # R1 candidate rows
    z_idx2: np.ndarray  # R1 candidate cols
    X_r: sp.csc_matrix  # R2 null matrix
    p_th: float  # R3
    # R4/R5: callables (kind, iteration, population, m) -> index vector into the candidate list
    sampler: Callable[[str, int, int, int], np.ndarray] = None
    log: List[Tuple[str, int, np.ndarray]] = field(default_factory=list)

    def sample(self, kind: str, it: int, population: int, m: int) -> np.ndarray:
        idx = np.asarray(self.sampler(kind, it, population, m), dtype=np.int64)
        assert idx.shape == (m,)
        self.log.append((kind, it, idx))
        return idx


def make_draws(X, seed: int, p_th_trials: int = 5000) -> Draws:
    """Convenience: draw R1-R3 with NumPy PCG64 and set up a seeded sampler for R4/R5."""
    X = _as_csc_f32(X)
    rng = np.random.default_rng(seed)
    z1, z2 = zero_candidates(X, rng)
    Xr = random_nz(X, rng)
    p_th = noise_baseline(min(X.shape), rng, p_th_trials)

    def sampler(kind, it, population, m):
        r = np.random.default_rng([seed, 1 if kind == "search" else 2, it])
        return r.choice(population, size=m, replace=False)

    return Draws(z1, z2, Xr, p_th, sampler)


def _with_ones(N, M, rows, cols, vals, z1, z2, idx, binary: bool) -> sp.csc_matrix:
    """sparse(vcat(nz_row, z_idx1[s]), vcat(nz_col, z_idx2[s]), vcat(vals, ones), N, M)
    (scLENS.jl:735, :738, :774; Appendix A23)."""
    r = np.concatenate([rows, z1[idx].astype(np.int64)])
    c = np.concatenate([cols, z2[idx].astype(np.int64)])
    v = np.concatenate([np.ones_like(vals) if binary else vals, np.ones(len(idx), dtype=np.float32)])
    return sp.csc_matrix((v, (r, c)), shape=(N, M), dtype=np.float32)


def _quantile7(x: np.ndarray, q: float) -> float:
    return float(np.quantile(x, q, method="linear"))  # Julia quantile default = type 7 (Appendix A27)


def robustness(nV: np.ndarray, nV_set: Sequence[np.ndarray], th: float = 60.0) -> dict:
    """Robustness scoring (scLENS.jl:786-807)."""
    P = len(nV_set)
    th_ = math.cos(math.radians(th))
    a_b = np.stack([np.argmax(np.abs(nV.T @ j), axis=1) for j in nV_set], axis=1)  # k x P (first max)
    sub = [nV_set[s][:, a_b[:, s]] for s in range(P)]
    b_vec = []
    for i in range(P):
        for j in range(i + 1, P):
            b_vec.append(np.max(np.abs(sub[i].T @ sub[j]), axis=1))
    b_ = np.stack(b_vec, axis=1) if b_vec else np.zeros((nV.shape[1], 0))
    k = b_.shape[0]
    m_score = np.empty(k)
    sd_score = np.empty(k)
    for s in range(k):
        row = b_[s]
        q1, q3 = _quantile7(row, 0.25), _quantile7(row, 0.75)
        iqr = q3 - q1
        f = row[(q1 - 1.5 * iqr <= row) & (row <= q3 + 1.5 * iqr)]
        m_score[s] = np.median(f)
        sd_score[s] = np.std(f, ddof=1) if f.size > 1 else float("nan")
    sig_id = np.flatnonzero(m_score > th_)
    return {"a_b": a_b, "b_": b_, "rob_score": m_score, "m_scores": m_score, "sd_scores": sd_score, "sig_id": sig_id}


def sclens(
    X,
    draws: Draws,
    th: float = 60.0,
    p_step: float = 0.001,
    n_perturb: int = 20,
    max_search_iters: Optional[int] = None,
    keep_intermediates: bool = False,
    null_tol: float = 0.0,
    centering: str = "mean",
) -> Dict[str, object]:
    """sclens(inp_df; device_="cpu", centering="mean" | "median") (scLENS.jl:649-832).

    `X` is the cells x genes count matrix (what `df2sparr(inp_df)` returns, :662).
    `max_search_iters` is a test-only cap on the sparsity-search loop (None = reference behaviour).
    `null_tol`: see NULL_DROP above (0 = literal reference behaviour).
    """
    X_ = _as_csc_f32(X)
    N, M = X_.shape
    coo = X_.tocoo()  # findnz: column by column, rows ascending (Appendix A1)
    order = np.lexsort((coo.row, coo.col))
    nz_row, nz_col, nz_val = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order]
    z1, z2 = draws.z_idx1, draws.z_idx2

    if centering == "mean":
        ls = logn_scale  # :651-652
        scaled_X, rec_vals = scale_main(X_)  # :676-696
    elif centering == "median":
        ls = logn_scale_median  # :653-654
        scaled_X, rec_vals = ls(pre_scale(X_)), {}  # :697-698 (rec_vals stays empty)
    else:
        ls = logn_scale_other  # :655-657 (the reference prints a warning first)
        scaled_X, rec_vals = ls(pre_scale(X_)), {}  # :697-698
    Xr_scaled = ls(pre_scale(draws.X_r))  # :704
    nL, nV, L, L_mp, lambda_c, _ = get_sigev(scaled_X, Xr_scaled)  # :704
    mpC = mp_check(L_mp)  # :706
    p_th = draws.p_th  # :709-712

    # ---- sparsity search (:715-762) ----
    p_ = 0.999
    binary = sp.csc_matrix((np.ones_like(nz_val), (nz_row, nz_col)), shape=(N, M), dtype=np.float32)
    sb = ls(pre_scale(binary))
    Vr2 = get_eigvec(sb.T if N > M else sb, null_tol)[1]  # :717-721
    n_2 = int(round(Vr2.shape[1] / 2))  # round half to even (Appendix A17)
    tank = np.zeros((5, 0))
    it = 0
    search_trace = []
    while True:
        nnzidx = int(round((1 - p_) * M * N))  # :726 (Appendix A21)
        if len(z1) < nnzidx:
            p_ += p_step
            break
        idx = draws.sample("search", it, len(z1), nnzidx)
        pert = _with_ones(N, M, nz_row, nz_col, nz_val, z1, z2, idx, binary=True)
        sp_ = ls(pre_scale(pert))
        nV_2 = get_eigvec(sp_.T if N > M else sp_, null_tol)[1]  # :733-739
        C = corr_mat(Vr2, nV_2[:, nV_2.shape[1] - n_2 - 1 :])  # end-n_2:end -> n_2+1 columns (A17)
        d_arr = np.nanmax(np.abs(C), axis=0)  # :742 (A18)
        tmp_A = np.sort(d_arr)
        tank = np.hstack([tank, tmp_A[:5, None]])
        ppj = tank[1, :] if tank.shape[1] < 5 else tank[1, -5:]
        search_trace.append((p_, tmp_A[:5].copy()))
        it += 1
        if (np.sum(ppj < p_th) > 4) or (p_ < 0.9) or (max_search_iters is not None and it >= max_search_iters):
            p_ += 4 * p_step
            break
        p_ -= p_step

    # ---- perturbation ensemble (:767-778) ----
    min_s = nV.shape[1]
    min_pc = int(math.ceil(min_s * 1.5))
    nV_set, nL_set = [], []
    m_pert = int(round((1 - p_) * M * N))
    for t in range(n_perturb):
        idx = draws.sample("perturb", t, len(z1), m_pert)
        tmp_X = _with_ones(N, M, nz_row, nz_col, nz_val, z1, z2, idx, binary=False)
        tL, tV = get_eigvec(ls(pre_scale(tmp_X)), null_tol)
        c = min(min_pc, tV.shape[1])
        nV_set.append(tV[:, :c])
        nL_set.append(tL[:c])

    res: Dict[str, object] = {"L": L, "L_mp": L_mp, "lambda_c": lambda_c, "p_": p_, "p_th": p_th,
                              "n_search": it, "search_trace": search_trace, "signal_ev": nL,
                              "signal_evec": nV, "pass": mpC["pass"], "ks_static": mpC["ks_static"],
                              "rec_vals": rec_vals, "min_pc": min_pc}
    if min_s == 0:  # :780-784
        return res
    rob = robustness(nV, nV_set, th)  # :786-807
    sig_id = rob["sig_id"]
    Xout0 = nV * np.sqrt(nL)[None, :]  # :810
    Xout1 = nV[:, sig_id] * np.sqrt(nL[sig_id])[None, :]  # :811
    gene_basis = ((1.0 / np.sqrt(nL))[:, None] * nV.T) @ scaled_X / math.sqrt(M)  # :818 (A29)
    res.update({"pca": Xout0, "pca_n1": Xout1, "sig_id": sig_id, "robustness_scores": rob,
                "gene_basis": gene_basis, "nL_set": nL_set})
    if keep_intermediates:
        res["nV_set"] = nV_set
        res["scaled_X"] = scaled_X
    return res


def get_denoised(res: Dict[str, object]) -> np.ndarray:
    """get_denoised_df (scLENS.jl:889-931), CPU branch: returns the N x M matrix `rcov_mean` (:926)."""
    g_mat = np.asarray(res["gene_basis"])[np.asarray(res["sig_id"]), :]
    Xout0 = np.asarray(res["pca_n1"], dtype=np.float32).astype(np.float64)  # Matrix{Float32}(pca_n1) (:891)
    M = np.asarray(res["gene_basis"]).shape[1]
    d_mean = (Xout0 @ g_mat) * math.sqrt(M)
    rv = res["rec_vals"]
    norm_tgc = np.ravel(rv["norm_tgc"])
    r1 = d_mean + np.ravel(rv["cent_"])[None, :]
    r2 = r1 * (norm_tgc / norm_tgc.mean())[:, None]
    r3 = r2 * np.ravel(rv["mat2_std"])[None, :] + np.ravel(rv["mat2_mean"])[None, :]
    r4 = np.exp(r3) - 1.0
    r4[r4 < 0] = 0.0
    r4 /= r4.sum(axis=1, keepdims=True)
    return r4 * np.ravel(rv["TGC"]).mean()


# --------------------------------------------------------------------------------------
# QC used only to build fixtures from the bundled datasets (scLENS.jl:160-236, defaults)
# --------------------------------------------------------------------------------------


def preprocess_counts(X, gene_names: Sequence[str], min_tp_c=0, min_tp_g=0, max_tp_c=np.inf, max_tp_g=np.inf,
                      min_genes_per_cell=200, max_genes_per_cell=0, min_cells_per_gene=15, mito_percent=5.0,
                      ribo_percent=0.0):
    """preprocess(tmp_df; ...) on a cells x genes count matrix (scLENS.jl:160-236), same keyword arguments and defaults.
    Returns (filtered matrix with genes sorted by mean count, their names, indices of the kept cells) or None
    ("There is no high quality cells and genes", :233). The step before the hot path; also used so that
    fixtures start from the same QC'd matrix as `example.jl`."""
    import re

    X = np.asarray(X.todense() if sp.issparse(X) else X, dtype=np.float32)
    names = np.asarray(gene_names)
    n_cell_counts = (X != 0).sum(axis=0)  # :181
    gsum = X.sum(axis=0, dtype=np.float32)
    fg = (gsum > min_tp_g) & (gsum < max_tp_g) & (n_cell_counts >= min_cells_per_gene)  # :183-186
    n_gene_counts = (X != 0).sum(axis=1)
    csum = X.sum(axis=1, dtype=np.float32)
    mito = np.array([bool(re.match(r"(?i)^mt-.", g)) for g in names])  # :194
    ribo = np.array([bool(re.match(r"(?i)^RP[SL].", g)) for g in names])  # :195
    with np.errstate(invalid="ignore", divide="ignore"):  # Float32 ratio compared in Float64 (Julia promotion)
        b4 = np.ones(len(csum), bool) if mito_percent == 0 else \
            (X[:, mito].sum(axis=1, dtype=np.float32) / csum).astype(np.float64) < mito_percent / 100
        b5 = np.ones(len(csum), bool) if ribo_percent == 0 else \
            (X[:, ribo].sum(axis=1, dtype=np.float32) / csum).astype(np.float64) < ribo_percent / 100
    b6 = np.ones(len(csum), bool) if max_genes_per_cell == 0 else n_gene_counts < max_genes_per_cell
    fc = (csum > min_tp_c) & (csum < max_tp_c) & (n_gene_counts >= min_genes_per_cell) & b4 & b5 & b6  # :215
    if not (fc.any() and fg.any()):
        return None
    oo = X[fc][:, fg]
    nn = oo.sum(axis=0, dtype=np.float32) != 0  # :219
    oo = oo[:, nn]
    g = names[fg][nn]
    s_idx = np.argsort(oo.mean(axis=0, dtype=np.float32), kind="stable")  # :223
    return oo[:, s_idx], g[s_idx], np.flatnonzero(fc)
