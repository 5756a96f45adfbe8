"""-m gpu: medium-size parity against the oracle and size-independent properties at sizes the oracle cannot reach quickly
(spectrum identities, orthonormality, determinism, ranges)."""
import numpy as np
import pytest

from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd.synth import synth_counts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,C", [(1200, 2000, 8), (2000, 800, 6)])
def test_medium_parity_full_search(ctx, N, M, C):
    X = synth_counts(N, M, seed=N, C=C, marker_frac=0.12, marker_sd=1.3)
    d = api.make_draws_native(X, seed=3, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=6, null_tol=O.NULL_DROP)
    res = api.sclens(X, draws=api.make_draws_native(X, seed=3), n_perturb=6, ctx=ctx, streams=3)
    k = len(ref["signal_ev"])
    assert len(res["signal_ev"]) == k >= C - 2
    assert np.allclose(res["signal_ev"], ref["signal_ev"], rtol=2e-4)
    assert np.abs(res["L"] - ref["L"]).max() < 2e-4 * ref["L"].max()
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    tr, trr = np.array([a for _, a in res["search_trace"]]), np.array([a for _, a in ref["search_trace"]])
    assert np.abs(tr - trr).max() < 3e-3
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.abs(res["robustness_scores"]["rob_score"] - ref["robustness_scores"]["rob_score"]).max() < 5e-3
    cos = np.abs(np.sum(res["signal_evec"].astype(np.float64) * ref["signal_evec"], axis=0))
    gaps = np.minimum(np.abs(np.diff(ref["signal_ev"], prepend=np.inf)), np.abs(np.diff(ref["signal_ev"], append=ref["lambda_c"])))
    assert np.all(cos[gaps > 0.02 * ref["signal_ev"]] > 1 - 2e-3)


def test_large_size_properties(ctx):
    N, M = 3000, 5000
    X = synth_counts(N, M, seed=77, C=8, marker_frac=0.08, marker_sd=1.2)
    d = api.make_draws_native(X, seed=5)
    a = api.sclens(X, draws=d, n_perturb=6, ctx=ctx, streams=3, keep_intermediates=True)
    b = api.sclens(X, draws=d, n_perturb=6, ctx=ctx, streams=2)
    # determinism: different concurrency, same bits
    assert np.array_equal(a["L"], b["L"]) and a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
    L = a["L"]
    assert np.all(np.diff(L) >= 0)  # ascending like eigen() / syevd!
    assert L[0] > -1e-5 * L[-1]  # Gram matrix is PSD up to rounding
    # trace identity: sum of eigenvalues of XX'/M = ||X||_F^2 / M, and every centred+scaled row has squared norm ~ M
    S = O.scale_main(X)[0]
    assert abs(L.sum() - (S * S).sum() / M) < 1e-4 * L.sum()
    V = a["signal_evec"].astype(np.float64)
    k = V.shape[1]
    assert np.abs(V.T @ V - np.eye(k)).max() < 1e-4  # orthonormal signal vectors
    assert 0.9 <= a["p_"] < 1.0 and 5 <= a["n_search"] <= 110
    rs = a["robustness_scores"]["rob_score"]
    assert np.all((rs >= 0) & (rs <= 1 + 1e-6))
    for Vt in a["nV_set"]:
        assert np.abs(np.linalg.norm(Vt, axis=0) - 1).max() < 1e-4
    # pca = signal_evec * sqrt(signal_ev) (scLENS.jl:810)
    assert np.allclose(a["pca"], a["signal_evec"] * np.sqrt(a["signal_ev"])[None, :], rtol=1e-5, atol=1e-6)
    # gene_basis rows reproduce V' X / sqrt(lambda M) on the oracle's scaled matrix
    gb = ((1.0 / np.sqrt(a["signal_ev"]))[:, None] * V.T) @ S / np.sqrt(M)
    assert np.abs(a["gene_basis"] - gb).max() < 5e-3 * np.abs(gb).max()


def test_wishart_spectrum_identity_on_device(ctx):
    """XX'/M and X'X/M share their non-zero spectrum (both computed through the C ABI drop-ins)."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((300, 500)).astype(np.float32)
    L1, _ = api._get_eigen(api._wishart_matrix(X, dims=1, ctx=ctx), ctx=ctx)
    L2, _ = api._get_eigen(api._wishart_matrix(X, dims=2, ctx=ctx), ctx=ctx)
    assert np.abs(L1 - L2[-300:]).max() < 2e-5 * L1.max()
    assert np.abs(L2[:200]).max() < 2e-5 * L1.max()


def test_full_size_cfg2_properties(ctx):
    """BASELINE.json configs[1] (10 000 x 20 000, what bench.py times), where the float64 oracle needs hours: properties that
    do not depend on the size. (1) determinism: 3 vs 2 concurrent streams give the same bits; (2) the spectrum is ascending,
    PSD up to rounding, and its sum equals ||X||_F^2 / M of the scaled matrix (trace identity; X from the scaling drop-in);
    (3) every signal eigenpair satisfies the eigen-equation of the Gram matrix XX'/M to fp32 accuracy, with X applied on the
    host in float64 (two matrix-vector products per vector, no Gram matrix needed); (4) decision outputs are in range and
    the result dictionary is internally consistent."""
    N, M = 10000, 20000
    X = synth_counts(N, M, seed=20240427 + 1, C=8)
    d = api.make_draws_native(X, seed=1000)
    a = api.sclens(X, draws=d, ctx=ctx, streams=3, n_perturb=20)
    b = api.sclens(X, draws=d, ctx=ctx, streams=2, n_perturb=20)
    assert np.array_equal(a["L"], b["L"]) and a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
    assert np.array_equal(a["signal_evec"], b["signal_evec"])
    L = a["L"]
    assert L.shape == (N,) and np.all(np.diff(L) >= 0) and L[0] > -1e-5 * L[-1]
    S, rec = api.logn_scale(X, "mean", inline_f64=True, ctx=ctx)  # the scaled data matrix the path decomposes
    S = S.astype(np.float64)
    assert abs(L.sum() - (S * S).sum() / M) < 1e-4 * L.sum()
    assert np.abs(S.mean(axis=0)).max() < 1e-5  # column-centred (scLENS.jl:695-696)
    for key in ("TGC", "mat2_mean", "mat2_std", "norm_tgc", "cent_"):
        assert np.allclose(np.ravel(a["rec_vals"][key]), np.ravel(rec[key]), rtol=1e-12, atol=0)
    V, lam = a["signal_evec"].astype(np.float64), a["signal_ev"]
    k = V.shape[1]
    assert k == len(lam) >= 6 and np.all(np.diff(lam) < 0) and np.all(lam > a["lambda_c"])
    assert np.abs(V.T @ V - np.eye(k)).max() < 1e-4
    GV = S @ (S.T @ V) / M
    resid = np.abs(GV - V * lam[None, :]).max(axis=0)
    assert np.all(resid < 2e-4 * L[-1] / np.sqrt(N) * 50), resid  # |Gv - lambda v|_inf, v has entries ~ 1/sqrt(N)
    assert np.all(np.linalg.norm(GV - V * lam[None, :], axis=0) < 5e-4 * L[-1])
    assert 0.9 <= a["p_"] < 1.0 and 5 <= a["n_search"] <= 110
    rs = a["robustness_scores"]["rob_score"]
    assert rs.shape == (k,) and np.all((rs >= 0) & (rs <= 1 + 1e-6))
    assert np.array_equal(a["sig_id"], np.flatnonzero(rs > 0.5))
    assert a["partial_eig"][0] + a["partial_eig"][1] == 20
    assert np.allclose(a["pca"], a["signal_evec"] * np.sqrt(lam)[None, :].astype(np.float32), rtol=1e-5, atol=1e-6)
