"""float64 spectra of the data and null matrices of a BASELINE.json configuration, as a FIXTURE (no GPU; ~1.5 h on 8 cores for cfg4).

What the reference's CPU branch computes for `get_sigev(X, Xr)` (scLENS.jl:526-541): eigenvalues of the Wishart matrices of the scaled
data matrix (inline Float64 twin, :676-696) and of the scaled null matrix (`logn_scale(pre_scale(X_r))`, :650-652, :701-702), the MP
fixed point (:424-459), the TW threshold (:461-467) and the signal count `sum(L .> lambda_c)` (:539). Here with the oracle's
functions (test infrastructure: oracle/sclens_oracle.py), the dense scaled matrices formed in row chunks so that 100 000 x 30 000
fits a 64 GB host; `--selftest` pins the chunked products against the oracle's own `wishart_matrix(scale_main(X))` /
`wishart_matrix(logn_scale(pre_scale(X_r)))` at a small size.

Output (`tests/golden/<cfg>_f64_spectra.npz`): L (data, ascending), Lr (null, ascending), lambda_c, k, b_minus / b_plus of the MP fit, the seeds.
`tests/test_gpu_bench_size.py::test_spectrum_of_the_shipped_arithmetic_against_float64` compares the device's spectra with it on every
GPU run, so the check no longer depends on a one-off run of an older build.

Usage: f64_spectra.py [cfg4|cfg3|tiny_gt] [--out file.npz] [--selftest]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

CFG = {"tiny_gt": (900, 400, 0), "cfg3": (50000, 30000, 2), "cfg4": (100000, 30000, 3)}
DRAW_SEED = 1000  # bench.py: seed_base + step 0


def gram_of_scaled_f64(Y: sp.csc_matrix, f32_std: bool, chunk: int = 4000, log=None) -> np.ndarray:
    """(1 / M) S' S for S = zscore_with_l2(Y) centred per gene (scLENS.jl:596-605 + :300-305; the inline twin :682-696 when
    f32_std is False), cells > genes: get_sigev (:526-541) takes `_wishart_matrix(X; dims=2)` = X'X / size(X, 2) = / GENES (:352-359).
    The same operations as oracle.zscore_with_l2 on the sparse side; the dense N x M matrix exists only `chunk` rows at a time."""
    N, M = Y.shape
    _, std = O._sparse_col_mean_std(Y, np.float32 if f32_std else np.float64)
    inv_std = 1.0 / std.astype(np.float64)
    Z = Y.astype(np.float64).multiply(inv_std[None, :]).tocsr()
    mu = np.asarray(Z.sum(axis=0)).ravel() / N
    l2X2 = np.asarray(Z.multiply(Z).sum(axis=1)).ravel()
    l2mu = np.linalg.norm(mu)
    l2norm = np.sqrt(l2X2 - 2.0 * (Z @ mu) + l2mu * l2mu)
    inv_s = 1.0 / (l2norm / l2norm.mean())
    cent = (np.asarray(Z.T @ inv_s).ravel() - mu * inv_s.sum()) / N  # column means of (Z - mu) / s
    G = np.zeros((M, M))
    t0 = time.perf_counter()
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        D = Z[a:b].toarray()
        D -= mu[None, :]
        D *= inv_s[a:b, None]
        D -= cent[None, :]
        # column blocks through dgemm (NumPy maps `D.T @ D` onto dsyrk, which crashes in the bundled OpenBLAS at order 30 000)
        for j0 in range(0, M, 6000):
            G[:, j0:j0 + 6000] += D.T @ D[:, j0:j0 + 6000]
        if log and (a // chunk) % 5 == 0:
            log(f"  gram rows {b}/{N} ({time.perf_counter() - t0:.0f} s)")
    G /= M
    return G


def data_gram_f64(X, **kw):
    """the inline Float64 twin (scLENS.jl:676-681): TGC and log1p in Float64, then :682-696"""
    X = O._as_csc_f32(X)
    tgc = np.asarray(X.astype(np.float64).sum(axis=1)).ravel()
    mat2 = X.astype(np.float64).tocsc()
    mat2.data = np.log1p(mat2.data / tgc[mat2.indices])
    return gram_of_scaled_f64(mat2, f32_std=False, **kw)


def null_gram_f64(Xr, **kw):
    """the closure path (scLENS.jl:650-652): Float32 proj_l + log1p, Float32 std, everything after in Float64"""
    return gram_of_scaled_f64(O.pre_scale(Xr), f32_std=True, **kw)


def selftest():
    X = api._csc_f32(synth_counts(700, 300, seed=5, C=4))
    Xr = O.random_nz(X, np.random.default_rng(3))
    ref = O.wishart_matrix(O.scale_main(X)[0], 2)
    got = data_gram_f64(X, chunk=128)
    e1 = np.abs(got - ref).max() / np.abs(ref).max()
    ref = O.wishart_matrix(O.logn_scale(O.pre_scale(Xr)), 2)
    got = null_gram_f64(Xr, chunk=128)
    e2 = np.abs(got - ref).max() / np.abs(ref).max()
    assert e1 < 1e-12 and e2 < 1e-12, (e1, e2)
    print(f"selftest: chunked float64 Gram == oracle (data {e1:.1e}, null {e2:.1e})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", nargs="?", default="cfg4", choices=list(CFG))
    ap.add_argument("--out")
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    selftest()
    if a.selftest:
        return
    N, M, idx = CFG[a.cfg]
    assert N > M
    out = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"{a.cfg}_f64_spectra.npz")
    T0 = time.perf_counter()
    log = lambda m: print(f"[{time.perf_counter() - T0:7.0f} s] {m}", flush=True)
    X = api._csc_f32(synth_counts(N, M, seed=20240427 + idx))
    log(f"matrix {N} x {M}, nnz {X.nnz}")
    d = api.make_draws_native(X, seed=DRAW_SEED, host_sampler=True)
    Xr = api._csc_f32(api._resolve(d.X_r))
    log(f"null matrix nnz {Xr.nnz}")
    spectra, times = {}, {}
    for name, fn, mat in (("L", data_gram_f64, X), ("Lr", null_gram_f64, Xr)):
        t = time.perf_counter()
        G = fn(mat, log=log)
        times[name + "_gram_s"] = time.perf_counter() - t
        log(f"{name}: Gram done; dsyevd (values only) ...")
        t = time.perf_counter()
        spectra[name] = sla.eigh(G, eigvals_only=True, driver="evd", overwrite_a=True, check_finite=False)
        times[name + "_eig_s"] = time.perf_counter() - t
        del G
        log(f"{name}: [{spectra[name][0]:.3e}, {spectra[name][-1]:.6f}]")
        np.savez(out + ".partial.npz", **spectra)
    L, Lr = spectra["L"], spectra["Lr"]
    L_mp, b_plus, b_minus = O.mp_calculation(L, Lr[:-1])
    lam_c = float(O.tw(L, L_mp)[0])
    k = int(np.sum(L > lam_c))
    log(f"lambda_c {lam_c:.9f}, k {k}")
    np.savez(out, L=L, Lr=Lr, lambda_c=lam_c, k=k, b_plus=float(b_plus), b_minus=float(b_minus), N=N, M=M, synth_seed=20240427 + idx,
             draw_seed=DRAW_SEED, nnz=int(X.nnz), **{q: float(v) for q, v in times.items()})
    os.remove(out + ".partial.npz")
    log(f"written {out}")


if __name__ == "__main__":
    main()
