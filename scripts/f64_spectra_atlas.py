"""float64 spectra of the data and null matrices of the ATLAS configuration (BASELINE.json configs[4]: 1 000 000 cells x 30 000 genes)
as a FIXTURE (no GPU; ~3.5 h on 8 cores, 45 GB of host memory, 50 GB of slab files in the cache directory).

The same quantities as scripts/f64_spectra.py (what the reference's CPU branch computes for `get_sigev(X, Xr)`, scLENS.jl:526-541:
eigenvalues of the Wishart matrices of the scaled data matrix -- inline Float64 twin :676-696 -- and of the scaled null matrix --
`logn_scale(pre_scale(X_r))` :650-652, :701-702 --, the MP fixed point :424-459, the TW threshold :461-467, the signal count :539), for a
matrix that is never held in one piece: the data matrix is `sclens_amd.atlas.synth_slabs` (seed 20240427 + 4), the null matrix
`sclens_amd.atlas.null_slabs` (the library's R2 generator on the WHOLE matrix, draw seed 1000), both as 8 row slabs on disk. The
normalisation statistics that span all cells (scLENS.jl:597-603) are accumulated slab by slab in float64 -- per-gene sum, then per-gene
squared deviations (two passes, as `oracle._sparse_col_mean_std`), then the cells' norms and the centring vector --, the Gram matrix
(1/M) S'S as a sum over cell blocks of 4 000 rows (the identity behind scLENS.jl:332-361 for dims = 2), eigenvalues by LAPACK dsyevd.
`--selftest` pins this slab-wise arithmetic against the oracle's dense path (`wishart_matrix(scale_main(X))`,
`wishart_matrix(logn_scale(pre_scale(X_r)))`) on a small matrix cut into 3 slabs.

Output: tests/golden/cfg5_f64_spectra.npz (L, Lr ascending, lambda_c, k, b_minus / b_plus, seeds, timings).
Usage: f64_spectra_atlas.py [--selftest] [--n-total N --m M --world W --out file.npz]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from oracle import sclens_oracle as O  # checker
from sclens_amd import atlas

DRAW_SEED = 1000
SYNTH_SEED = 20240427 + 4


def _lg_of(slab: sp.csc_matrix, closure: bool) -> sp.csc_matrix:
    """log-normalised stored entries of one slab (cells are whole inside a slab, so TGC is local): the closure path
    `log1p.(proj_l(x))` in Float32 (scLENS.jl:650, :607) or the inline Float64 twin (:676-681); returned as float64 CSC"""
    if closure:
        return O.pre_scale(slab).astype(np.float64).tocsc()
    X = O._as_csc_f32(slab)
    tgc = np.asarray(X.astype(np.float64).sum(axis=1)).ravel()
    Y = X.astype(np.float64).tocsc()
    Y.data = np.log1p(Y.data / tgc[Y.indices])
    return Y


def gram_of_slabs_f64(S: atlas.SlabSet, closure: bool, chunk: int = 4000, blk: int = 6000, log=None):
    """(1 / M) B'B for B = scaled_gdata(zscore_with_l2(Y), "cent") of the whole matrix (scLENS.jl:596-605, :300-305; :682-696 for the
    data matrix), accumulated over the slabs. Returns the LOWER triangle (block columns of `blk`) of the M x M matrix."""
    N, M = S.N_total, S.M
    col = lambda v: np.asarray(v).ravel()
    t0 = time.perf_counter()
    # pass 1: per-gene sums and entry counts (mean(X, dims=1), :597 / :682)
    s1, nzc = np.zeros(M), np.zeros(M)
    for g in range(S.world):
        Y = _lg_of(S.slab(g), closure)
        s1 += col(Y.sum(axis=0))
        nzc += np.diff(Y.indptr)
    mean = s1 / N
    # pass 2: squared deviations, implicit zeros included (std(X, dims=1), :599 / :683)
    s2 = np.zeros(M)
    for g in range(S.world):
        Y = _lg_of(S.slab(g), closure)
        dev = Y.data - np.repeat(mean, np.diff(Y.indptr))
        a = np.add.reduceat(np.append(dev * dev, 0.0), Y.indptr[:-1])
        a[np.diff(Y.indptr) == 0] = 0.0
        s2 += a
    s2 += (N - nzc) * mean * mean
    std = np.sqrt(s2 / (N - 1))
    if closure:  # std of a Float32 matrix is Float32 (Appendix A4), everything after it Float64
        std = std.astype(np.float32).astype(np.float64)
    inv_std = 1.0 / std
    mu = (s1 * inv_std) / N  # mean(X_norm, dims=1)
    l2mu2 = float(mu @ mu)
    # pass 3: the cells' norms (:601-603 / :688-690) and sum_i Z_ij / l_i
    l2 = []
    tz = np.zeros(M)
    for g in range(S.world):
        Z = _lg_of(S.slab(g), closure).multiply(inv_std[None, :]).tocsr()
        l = np.sqrt(col(Z.multiply(Z).sum(axis=1)) - 2.0 * (Z @ mu) + l2mu2)
        l2.append(l)
        tz += col(Z.T @ (1.0 / l))
    lall = np.concatenate(l2)
    lmean = lall.mean()
    inv_s = [lmean / l for l in l2]  # 1 ./ (l2norm ./ mean(l2norm))
    cent = (lmean * tz - mu * sum(float(v.sum()) for v in inv_s)) / N  # mean(., dims=1) of the row-scaled, centred matrix (:300-305)
    if log:
        log(f"  statistics done ({time.perf_counter() - t0:.0f} s); mean cell norm {lmean:.6f}")
    # pass 4: Gram matrix over cell blocks
    G = np.zeros((M, M))
    done = 0
    for g in range(S.world):
        Z = _lg_of(S.slab(g), closure).multiply(inv_std[None, :]).tocsr()
        for a in range(0, Z.shape[0], chunk):
            b = min(Z.shape[0], a + chunk)
            D = Z[a:b].toarray()
            D -= mu[None, :]
            D *= inv_s[g][a:b, None]
            D -= cent[None, :]
            for j0 in range(0, M, blk):  # lower block columns only (NumPy's own dsyrk path crashes in the bundled OpenBLAS at this order)
                G[j0:, j0:j0 + blk] += D[:, j0:].T @ D[:, j0:j0 + blk]
            done += b - a
        if log:
            log(f"  gram rows {done}/{N} ({time.perf_counter() - t0:.0f} s)")
    G /= M
    return G, {"mean": mean, "std": std, "mu": mu, "cent": cent, "lmean": lmean}


def selftest():
    import tempfile

    d = tempfile.mkdtemp()
    N, M = 6000, 150
    S = atlas.synth_slabs(N, M, 5, 3, cache_dir=d, workers=1)
    assert all(np.diff(S.slab(g).indptr).min() > 0 for g in range(3))  # (no empty gene: its std would be 0)
    R = atlas.null_slabs(S, 3, cache_dir=d)
    X = O._as_csc_f32(sp.vstack([S.slab(g) for g in range(3)]).tocsc())
    Xr = O._as_csc_f32(sp.vstack([R.slab(g) for g in range(3)]).tocsc())
    low = np.tril_indices(M)
    ref = O.wishart_matrix(O.scale_main(X)[0], 2)
    got, st = gram_of_slabs_f64(S, closure=False, chunk=97, blk=64)
    e1 = np.abs(got[low] - ref[low]).max() / np.abs(ref).max()
    rec = O.scale_main(X)[1]
    assert np.allclose(st["std"], rec["mat2_std"], rtol=1e-12) and np.allclose(st["cent"], rec["cent_"], rtol=1e-9, atol=1e-15)
    ref = O.wishart_matrix(O.logn_scale(O.pre_scale(Xr)), 2)
    got, _ = gram_of_slabs_f64(R, closure=True, chunk=97, blk=64)
    e2 = np.abs(got[low] - ref[low]).max() / np.abs(ref).max()
    assert e1 < 1e-12 and e2 < 1e-12, (e1, e2)
    print(f"selftest: slab-wise float64 Gram == oracle (data {e1:.1e}, null {e2:.1e})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--n-total", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=30_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--out")
    a = ap.parse_args()
    selftest()
    if a.selftest:
        return
    N, M = a.n_total, a.m
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = a.out or os.path.join(root, "tests", "golden", "cfg5_f64_spectra.npz")
    T0 = time.perf_counter()
    log = lambda m: print(f"[{time.perf_counter() - T0:7.0f} s] {m}", flush=True)
    S = atlas.synth_slabs(N, M, SYNTH_SEED, a.world, workers=a.workers, log=log)
    log(f"data matrix {N} x {M}, nnz {S.nnz_total}")
    R = atlas.null_slabs(S, DRAW_SEED, log=log)
    log(f"null matrix nnz {R.nnz_total}")
    spectra, times = {}, {}
    for name, slabs, closure in (("L", S, False), ("Lr", R, True)):
        t = time.perf_counter()
        G, _ = gram_of_slabs_f64(slabs, closure, log=log)
        times[name + "_gram_s"] = time.perf_counter() - t
        log(f"{name}: Gram done; dsyevd (values only) ...")
        t = time.perf_counter()
        spectra[name] = sla.eigh(G, lower=True, eigvals_only=True, driver="evd", overwrite_a=True, check_finite=False)
        times[name + "_eig_s"] = time.perf_counter() - t
        del G
        log(f"{name}: [{spectra[name][0]:.3e}, {spectra[name][-1]:.6f}]")
        np.savez(out + ".partial.npz", **spectra)
    L, Lr = spectra["L"], spectra["Lr"]
    L_mp, b_plus, b_minus = O.mp_calculation(L, Lr[:-1])
    lam_c = float(O.tw(L, L_mp)[0])
    k = int(np.sum(L > lam_c))
    log(f"lambda_c {lam_c:.9f}, k {k}")
    np.savez(out, L=L, Lr=Lr, lambda_c=lam_c, k=k, b_plus=float(b_plus), b_minus=float(b_minus), N=N, M=M, synth_seed=SYNTH_SEED,
             draw_seed=DRAW_SEED, nnz=int(S.nnz_total), **{q: float(v) for q, v in times.items()})
    os.remove(out + ".partial.npz")
    log(f"written {out}")


if __name__ == "__main__":
    main()
