"""float64 arbiter of ONE evaluation of the sparsity search at a BASELINE.json configuration (hours of host
time, no GPU): the statistic of scLENS.jl:742-747 -- d_arr[j] = max_i |Vr2_i' nV2_j| over the lower half of the perturbed binarised
matrix's eigenvectors, of which the stop rule (:756) compares the SECOND SMALLEST with p_th -- computed with LAPACK dsyevr on
float64 Gram matrices of the float64-scaled matrices, for the draws the device used (same seed: the library's host twins of the
device-side candidate draw and sampler give the same candidates and the same samples). It is the oracle's path
(oracle/sclens_oracle.py: pre_scale, zscore_with_l2, logn_scale, get_eigvec, corr_mat) with the dense scaled matrix formed in row
chunks, so that 100 000 x 30 000 fits the host (the dense float64 matrix would be 24 GB; here: two 7.2 GB Gram / vector
matrices at a time); `--selftest` pins the chunked form against the oracle's own functions at a small size.

Usage: check_search_step_f64.py cfg seed it [it ...] [--out file.json] [--threads n] [--device-null-rule]
  (--threads: the bundled OpenBLAS crashes in dsyrk / dsyr2k at order 30 000 with 2 .. 7 threads on the build box; 1 or all 8 work)
  cfg: cfg4 | cfg3 | tiny_gt ; seed: the draw seed of the sclens() call (bench.py: seed_base + step); it: 0-based search iterations
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.shard import search_schedule
from sclens_amd.synth import synth_counts

CFG = {"tiny_gt": (900, 400, 0), "cfg3": (50000, 30000, 2), "cfg4": (100000, 30000, 3)}


def scaled_gram_f64(P: sp.csc_matrix, chunk: int = 4000, log=None) -> np.ndarray:
    """(1 / N) S' S for S = logn_scale(pre_scale(P)) (scLENS.jl:650-652, :596-605, :300-305; the Wishart matrix of
    get_eigvec(S') for cells > genes, :345-360 with divisor = cells), S formed in chunks of rows in float64"""
    N, M = P.shape
    Y = O.pre_scale(P)  # Float32 proj_l + log1p on the stored entries
    mean, std = O._sparse_col_mean_std(Y, np.float32)  # std(X, dims=1) of a Float32 matrix is Float32 (Appendix A4)
    inv_std = 1.0 / std.astype(np.float64)
    Z = Y.astype(np.float64).multiply(inv_std[None, :]).tocsr()
    mu = np.asarray(Z.sum(axis=0)).ravel() / N
    l2X2 = np.asarray(Z.multiply(Z).sum(axis=1)).ravel()
    l2norm = np.sqrt(l2X2 - 2.0 * (Z @ mu) + float(mu @ mu))
    s = l2norm / l2norm.mean()
    inv_s = 1.0 / s
    cent = (np.asarray(Z.T @ inv_s).ravel() - mu * inv_s.sum()) / N  # column means of (Z - mu) / s
    G = np.zeros((M, M))
    t0 = time.perf_counter()
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        D = Z[a:b].toarray()
        D -= mu[None, :]
        D *= inv_s[a:b, None]
        D -= cent[None, :]
        # column blocks through dgemm: NumPy maps `D.T @ D` onto dsyrk, which segfaults inside the bundled OpenBLAS at order 30 000
        # (reproduced in isolation on the build box, round 4)
        for j0 in range(0, M, 6000):
            G[:, j0:j0 + 6000] += D.T @ D[:, j0:j0 + 6000]
        if log and (a // chunk) % 5 == 0:
            log(f"  gram rows {b}/{N} ({time.perf_counter() - t0:.0f} s)")
    G /= N
    return G


def selftest():
    X = api._csc_f32(synth_counts(700, 300, seed=5, C=4))
    P = sp.csc_matrix((np.ones_like(X.data), X.indices, X.indptr), shape=X.shape, dtype=np.float32)
    S = O.logn_scale(O.pre_scale(P))
    ref = O.wishart_matrix(S.T, 1)
    got = scaled_gram_f64(P, chunk=128)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-12, err
    print("selftest: chunked float64 Gram == oracle wishart_matrix(logn_scale(pre_scale(P))') to", err)


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", nargs="?", default="cfg4", choices=list(CFG))
    ap.add_argument("seed", nargs="?", type=int, default=1019)
    ap.add_argument("its", nargs="*", type=int)
    ap.add_argument("--out")
    ap.add_argument("--threads", type=int)
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--device-null-rule", action="store_true",
                    help="count an eigenvalue as positive iff it exceeds eps32 sqrt(n) lambda_max (the device's rounding floor, DESIGN.md "
                         "section 2 position 6) instead of the oracle's NULL_DROP = 1e-9 lambda_max: at 100 000 x 30 000 the binarised "
                         "matrix has ONE eigenvalue of 1.7e-9 lambda_max, which the two rules treat differently")
    a = ap.parse_args()
    if a.selftest:
        return selftest()
    cfg, seed, its, opts = a.cfg, a.seed, list(a.its), ({"out": a.out} if a.out else {})
    if a.threads:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=a.threads)
    N, M, idx = CFG[cfg]
    assert N > M, "cells > genes configurations only"
    out = {"config": cfg, "seed": seed, "iterations": its, "host_cores": os.cpu_count(), "results": [],
           "null_rule": "device (sessions since the end of round 6): eps32 sqrt(n) lambda_max" if a.device_null_rule else "oracle: NULL_DROP = 1e-9 lambda_max"}
    floor = (lambda w: 1.0 * 5.96e-8 * np.sqrt(len(w)) * w.max()) if a.device_null_rule else (lambda w: O.NULL_DROP * w.max())
    T0 = time.perf_counter()
    log = lambda m: print(f"[{time.perf_counter() - T0:7.0f} s] {m}", flush=True)
    selftest()
    X = api._csc_f32(synth_counts(N, M, seed=20240427 + idx))
    log(f"matrix {N} x {M}, nnz {X.nnz}")
    d = api.make_draws_native(X, seed=seed, host_sampler=True)  # host twins of the device draws: the same candidates, the same samples
    z1, z2 = api._resolve(d.z_idx1), api._resolve(d.z_idx2)
    log(f"{len(z1)} zero candidates, p_th {d.p_th:.6f}")
    out["p_th"] = float(d.p_th)
    out["n_cand"] = int(len(z1))
    P0 = sp.csc_matrix((np.ones_like(X.data), X.indices, X.indptr), shape=X.shape, dtype=np.float32)
    # Vr2 (:717-721): every eigenvector of the unperturbed binarised matrix with a positive eigenvalue
    G = scaled_gram_f64(P0, log=log)
    log("Gram of the binarised matrix done; dsyevr (all vectors) ...")
    L, V = sla.eigh(G, driver="evr", overwrite_a=True, check_finite=False)
    del G
    pos_ = L > floor(L)
    Vr2 = V[:, pos_]
    del V
    r = int(pos_.sum())
    n_2 = int(round(r / 2))
    log(f"Vr2: r = {r}, n_2 = {n_2}, lambda in [{L[0]:.3e}, {L[-1]:.3e}]")
    out["r"], out["n_2"] = r, n_2
    p_list = search_schedule(0.001)
    coo = X.tocoo()
    for it in its:
        p_ = p_list[it]
        m = int(round((1 - p_) * M * N))
        sidx = d.sampler("search", it, len(z1), m)
        rows = np.concatenate([coo.row.astype(np.int64), z1[sidx].astype(np.int64)])
        cols = np.concatenate([coo.col.astype(np.int64), z2[sidx].astype(np.int64)])
        P = sp.csc_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(N, M), dtype=np.float32)
        assert P.nnz == X.nnz + m, "candidates are disjoint from the stored entries and unique"
        log(f"iteration {it}: p_ = {p_:.3f}, {m} sampled zeros")
        G = scaled_gram_f64(P, log=log)
        log("  dsyevr (all values, lower-half vectors) ...")
        w = sla.eigh(G, driver="evr", eigvals_only=True, check_finite=False)
        npos = int((w > floor(w)).sum())
        lo = M - npos  # ascending index of the smallest positive eigenvalue
        # nV_2[:, end-n_2:end] of the DESCENDING order = the n_2 + 1 smallest positive eigenvalues (Appendix A17)
        w2, V2 = sla.eigh(G, driver="evr", subset_by_index=[lo, lo + n_2], overwrite_a=True, check_finite=False)
        del G
        C = Vr2.T @ V2
        d_arr = np.abs(C).max(axis=0)
        del C, V2
        d5 = np.sort(d_arr)[:5]
        log(f"  d5 (float64) = {d5.tolist()}  second smallest - p_th = {d5[1] - d.p_th:+.6f}")
        out["results"].append({"it": it, "p_": p_, "m": m, "positive": npos, "d5_f64": d5.tolist(),
                               "second_smallest_minus_p_th": float(d5[1] - d.p_th), "below_p_th": bool(d5[1] < d.p_th)})
        if "out" in opts:
            open(opts["out"], "w").write(json.dumps(out, indent=1) + "\n")
    out["wall_s"] = round(time.perf_counter() - T0, 1)
    print(json.dumps(out, indent=1))
    if "out" in opts:
        open(opts["out"], "w").write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
