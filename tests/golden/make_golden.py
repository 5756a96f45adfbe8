"""Generates the golden fixtures under tests/golden/ with the oracle (run in the BUILD container only; the bundled
datasets live under /root/reference/data, which does not exist on the GPU box).

PARITY UNPINNED: the reference (Julia) cannot be executed here and ships no golden vectors, so these fixtures pin the
oracle's own outputs: they guard the oracle against regressions and give the HIP path fixed, committed targets.
Each fixture stores the INPUT count matrix (CSC), every injected random draw, and the expected outputs. Eigenvectors
are stored as |V| column signs removed by the consumer (sign-invariant comparisons only).

    python tests/golden/make_golden.py [fixture names ...]
"""
import gzip
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import sclens_oracle as O  # noqa: E402
from sclens_amd.synth import synth_counts  # noqa: E402


def run_case(name, X, seed, n_perturb, centering="mean"):
    d = O.make_draws(X, seed, 2000)
    res = O.sclens(X, d, n_perturb=n_perturb, keep_intermediates=True, null_tol=O.NULL_DROP, centering=centering)
    search = [idx for kind, it, idx in d.log if kind == "search"]
    pert = [idx for kind, it, idx in d.log if kind == "perturb"]
    X = sp.csc_matrix(X)
    rob = res["robustness_scores"]
    out = dict(
        N=X.shape[0], M=X.shape[1], indptr=X.indptr.astype(np.int64), indices=X.indices.astype(np.int32),
        data=X.data.astype(np.float32),
        z1=d.z_idx1, z2=d.z_idx2, xr_indptr=d.X_r.indptr.astype(np.int64), xr_indices=d.X_r.indices.astype(np.int32),
        xr_data=d.X_r.data.astype(np.float32), p_th=np.float64(d.p_th),
        search_len=np.array([len(s) for s in search]), search_idx=np.concatenate(search).astype(np.uint32),
        pert_len=np.array([len(s) for s in pert]), pert_idx=np.concatenate(pert).astype(np.uint32),
        L=res["L"], n_L_mp=np.int64(len(res["L_mp"])), lambda_c=np.float64(res["lambda_c"]), signal_ev=res["signal_ev"],
        signal_evec=res["signal_evec"].astype(np.float32), p_=np.float64(res["p_"]), n_search=np.int64(res["n_search"]),
        search_trace=np.array([a for _, a in res["search_trace"]]), a_b=rob["a_b"], b_=rob["b_"], rob_score=rob["rob_score"],
        sig_id=res["sig_id"], nL_set=np.array(res["nL_set"]), mp_pass=np.bool_(res["pass"]), ks_static=np.float64(res["ks_static"]),
        gene_basis=res["gene_basis"].astype(np.float32), centering=np.str_(centering),
    )
    if centering == "mean":  # centering="median" leaves rec_vals empty (scLENS.jl:697-698)
        out.update(rec_TGC=res["rec_vals"]["TGC"], rec_mat2_mean=res["rec_vals"]["mat2_mean"],
                   rec_mat2_std=res["rec_vals"]["mat2_std"], rec_norm_tgc=res["rec_vals"]["norm_tgc"],
                   rec_cent=res["rec_vals"]["cent_"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, X.shape, "k =", len(res["signal_ev"]), "sig_id =", res["sig_id"], "S =", res["n_search"], "p_ =", res["p_"],
          f"{os.path.getsize(path) / 1e6:.2f} MB")


def run_case_native(name, X, seed, n_perturb):
    """the same for a matrix whose injected samples would not fit a fixture (3838 x 9083: 1e7 sampled indices): every draw comes from the
    library's own seeded generators (api.make_draws_native(..., host_sampler=True): C++ host code, no GPU), which the test regenerates
    from the seed -- the fixture holds the input matrix, the seed and the oracle's outputs"""
    from sclens_amd import api

    X = sp.csc_matrix(X)
    d = api.make_draws_native(api._csc_f32(X), seed=seed, host_sampler=True)
    od = O.Draws(api._resolve(d.z_idx1), api._resolve(d.z_idx2), api._resolve(d.X_r), d.p_th, d.sampler)
    res = O.sclens(X, od, n_perturb=n_perturb, keep_intermediates=False, null_tol=O.NULL_DROP)
    rob = res["robustness_scores"]
    out = dict(N=X.shape[0], M=X.shape[1], indptr=X.indptr.astype(np.int64), indices=X.indices.astype(np.int32), data=X.data.astype(np.float32),
               seed=np.int64(seed), n_perturb=np.int64(n_perturb), p_th=np.float64(d.p_th), n_cand=np.int64(len(od.z_idx1)),
               L=res["L"], n_L_mp=np.int64(len(res["L_mp"])), lambda_c=np.float64(res["lambda_c"]), signal_ev=res["signal_ev"],
               signal_evec=res["signal_evec"].astype(np.float32), p_=np.float64(res["p_"]), n_search=np.int64(res["n_search"]),
               search_trace=np.array([a for _, a in res["search_trace"]]), a_b=rob["a_b"], b_=rob["b_"], rob_score=rob["rob_score"],
               sig_id=res["sig_id"], nL_set=np.array(res["nL_set"]), mp_pass=np.bool_(res["pass"]), ks_static=np.float64(res["ks_static"]),
               gene_basis=res["gene_basis"].astype(np.float32), rec_TGC=res["rec_vals"]["TGC"], rec_mat2_mean=res["rec_vals"]["mat2_mean"],
               rec_mat2_std=res["rec_vals"]["mat2_std"], rec_norm_tgc=res["rec_vals"]["norm_tgc"], rec_cent=res["rec_vals"]["cent_"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, X.shape, "k =", len(res["signal_ev"]), "sig_id =", res["sig_id"], "S =", res["n_search"], "p_ =", res["p_"],
          f"{os.path.getsize(path) / 1e6:.2f} MB")


def known_answer_spectra():
    """Analytic MP cases for _mp_calculation/_tw: white Wishart -> 0 signals; rank-3 spikes above the BBP threshold -> 3."""
    rng = np.random.default_rng(42)
    n, K = 400, 1600
    out = {}
    for name, spikes in (("white", []), ("spiked", [6.0, 9.0, 14.0])):
        B = rng.standard_normal((n, K))
        for q, s in enumerate(spikes):
            u = rng.standard_normal(n)
            u /= np.linalg.norm(u)
            B += np.sqrt(s) * np.outer(u, rng.standard_normal(K))
        L = np.linalg.eigvalsh(B @ B.T / K)
        Br = rng.standard_normal((n, K))
        Lr = np.linalg.eigvalsh(Br @ Br.T / K)
        L_mp, bp, bm = O.mp_calculation(L, Lr[:-1])
        lam, gamma, p, sigma = O.tw(L, L_mp)
        chk = O.mp_check(L_mp)
        out[name + "_L"] = L
        out[name + "_Lr"] = Lr
        out[name + "_expect"] = np.array([len(L_mp), bp, bm, lam, gamma, p, sigma, chk["ks_static"], float(chk["pass"]),
                                          float(np.sum(L > lam))])
        print(name, "k =", int(np.sum(L > lam)), "lambda_c =", lam)
    np.savez_compressed(os.path.join(HERE, "mp_known_answers.npz"), **out)


def median_case_matrix(N, M):
    """synthetic counts in which a tenth of the genes is expressed in most cells (non-zero medians)"""
    X = synth_counts(N, M, seed=1, C=5, marker_frac=0.2, marker_sd=1.5).toarray()
    rng = np.random.default_rng(11)
    dg = rng.choice(M, size=M // 10, replace=False)
    X[:, dg] += rng.poisson(2.0, size=(N, len(dg))).astype(X.dtype)
    return sp.csc_matrix(X)


def main():
    only = set(sys.argv[1:])  # optional: names of the fixtures to (re)generate; default all

    def want(name):
        return not only or name in only

    if want("synth_300x500"):
        run_case("synth_300x500", synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5), seed=7, n_perturb=6)
    if want("synth_600x250"):
        run_case("synth_600x250", synth_counts(600, 250, seed=1, C=5, marker_frac=0.2, marker_sd=1.5), seed=7, n_perturb=6)
    if want("synth_300x500_median"):
        run_case("synth_300x500_median", median_case_matrix(300, 500), seed=7, n_perturb=5, centering="median")
    if want("mp_known_answers"):
        known_answer_spectra()
    ref = "/root/reference/data/Real_Zheng_data/z_data_785.csv.gz"
    if want("zheng_785") and os.path.exists(ref):
        import pandas as pd

        df = pd.read_csv(ref)
        Xq, genes, cells = O.preprocess_counts(df.iloc[:, 1:].to_numpy(dtype=np.float32), list(df.columns[1:]))
        print("z_data_785 after QC:", Xq.shape)
        run_case("zheng_785", sp.csc_matrix(Xq), seed=11, n_perturb=5)
    ref2 = "/root/reference/data/Real_Zheng_data/z_data_3869.csv.gz"  # BASELINE.md's closest analogue of the missing Z8eq (8 balanced types)
    if want("zheng_3869") and os.path.exists(ref2):
        import pandas as pd

        df = pd.read_csv(ref2)
        Xq, genes, cells = O.preprocess_counts(df.iloc[:, 1:].to_numpy(dtype=np.float32), list(df.columns[1:]))
        print("z_data_3869 after QC:", Xq.shape)
        run_case_native("zheng_3869", sp.csc_matrix(Xq), seed=17, n_perturb=6)


if __name__ == "__main__":
    main()
