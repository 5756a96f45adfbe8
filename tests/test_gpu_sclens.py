"""-m gpu: parity of the HIP path (through the C ABI) against the float64 oracle on the same injected draws.

Tolerances: the device path is fp32 (Gram, tridiagonalisation, back-transform) with fp64 statistics, the oracle
is the reference's Float64 CPU path. Eigenvalues: 2e-4 relative to the largest; eigenvectors: |cos| >= 1 - 2e-3
for separated signals; integer/decision outputs (signal count, p_, search length, sig_id, a_b) exact.
"""
import os

import numpy as np
import pytest

from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd._lib import Context
from sclens_amd.synth import synth_counts

pytestmark = pytest.mark.gpu


def _abs_cos(U, V):
    """|cos| between matching columns of two unit-column matrices."""
    return np.abs(np.sum(U.astype(np.float64) * V.astype(np.float64), axis=0))


@pytest.mark.parametrize("N,M", [(90, 150), (200, 70)])
def test_dropin_wishart_eigen_corr(ctx, N, M):
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, M)).astype(np.float32)
    for dims in (1, 2):
        Y = api._wishart_matrix(X, dims=dims, ctx=ctx)
        ref = O.wishart_matrix(X.astype(np.float64), dims)
        assert Y.shape == ref.shape
        assert np.abs(Y - ref).max() < 2e-6 * np.sqrt(max(N, M)) * np.abs(ref).max()
        assert np.array_equal(Y, Y.T)
    Y = O.wishart_matrix(X.astype(np.float64), 1)
    L, V = api._get_eigen(Y, ctx=ctx)
    Lr, _ = O.get_eigen(Y)
    assert np.all(np.diff(L) >= 0)
    assert np.abs(L - Lr).max() < 2e-5 * Lr.max()
    assert np.abs(Y @ V - V * L[None, :]).max() < 2e-5 * Lr.max()
    assert np.abs(V.T @ V - np.eye(N)).max() < 2e-5
    A = rng.standard_normal((N, 17)).astype(np.float32)
    B = rng.standard_normal((N, 23)).astype(np.float32)
    Cm = api.corr_mat(A, B, ctx=ctx)
    assert np.abs(Cm - A.astype(np.float64).T @ B.astype(np.float64)).max() < 1e-4
    assert api._wishart_matrix(X, device="tpu") is None  # unknown device string: reference returns nothing


@pytest.mark.parametrize("N,M", [(120, 200), (260, 90)])
def test_dropin_get_eigvec(ctx, N, M):
    X = synth_counts(N, M, seed=3, C=4, marker_frac=0.3, marker_sd=1.5)
    S = O.logn_scale(O.pre_scale(X))
    nL_ref, nV_ref = O.get_eigvec(S, O.NULL_DROP)
    nL, nV = api.get_eigvec(S.astype(np.float32), keep_top=6, ctx=ctx)
    assert len(nL) == len(nL_ref)  # structural null eigenvalue dropped on both sides (SURVEY 8a defect 6)
    assert np.abs(nL - nL_ref).max() < 2e-4 * nL_ref[0]
    assert nV.shape == (N, 6)
    assert np.all(_abs_cos(nV[:, :3], nV_ref[:, :3]) > 1 - 2e-3)
    assert np.abs(np.linalg.norm(nV, axis=0) - 1).max() < 1e-4


CASES = {
    "cells_le_genes": dict(N=300, M=500, seed=1, draws_seed=7),
    "cells_gt_genes": dict(N=600, M=250, seed=1, draws_seed=7),
}


@pytest.fixture(scope="module", params=list(CASES))
def run_pair(request, ctx):
    c = CASES[request.param]
    X = synth_counts(c["N"], c["M"], seed=c["seed"], C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws(X, seed=c["draws_seed"], p_th_trials=300)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=6, keep_intermediates=True, null_tol=O.NULL_DROP)
    res = api.sclens(X, draws=d, n_perturb=6, ctx=ctx, keep_intermediates=True)
    return X, ref, res


def test_sclens_spectrum_and_threshold(run_pair):
    X, ref, res = run_pair
    assert np.abs(res["L"] - ref["L"]).max() < 2e-4 * ref["L"].max()
    assert abs(res["lambda_c"] - ref["lambda_c"]) < 2e-4 * ref["lambda_c"]
    assert len(res["signal_ev"]) == len(ref["signal_ev"]) > 0  # retained-signal count identical
    assert np.all(np.diff(res["signal_ev"]) < 0)  # descending, same ordering
    assert np.allclose(res["signal_ev"], ref["signal_ev"], rtol=2e-4)
    assert len(res["L_mp"]) == len(ref["L_mp"])
    assert res["pass"] == ref["pass"]
    for key in ("TGC", "mat2_mean", "mat2_std", "norm_tgc", "cent_"):
        a, b = np.ravel(res["rec_vals"][key]), np.ravel(ref["rec_vals"][key])
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12), key


def test_sclens_signal_vectors(run_pair):
    X, ref, res = run_pair
    cos = _abs_cos(res["signal_evec"], ref["signal_evec"])
    assert np.all(cos > 1 - 2e-3), cos
    s = np.sign(np.sum(res["signal_evec"] * ref["signal_evec"], axis=0))
    assert np.abs(res["pca"] * s[None, :] - ref["pca"]).max() < 5e-3 * np.abs(ref["pca"]).max()
    assert np.abs(res["gene_basis"] * s[:, None] - ref["gene_basis"]).max() < 5e-3 * np.abs(ref["gene_basis"]).max()


def test_sclens_sparsity_search(run_pair):
    X, ref, res = run_pair
    assert res["n_search"] == ref["n_search"]
    assert res["p_"] == ref["p_"]  # same decision sequence -> bit-identical p_
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2
        assert np.abs(d1 - d2).max() < 3e-3, (d1, d2)


def test_sclens_robustness(run_pair):
    X, ref, res = run_pair
    rr, ro = res["robustness_scores"], ref["robustness_scores"]
    assert np.array_equal(rr["a_b"], ro["a_b"])
    assert np.abs(rr["b_"] - ro["b_"]).max() < 3e-3
    assert np.abs(rr["rob_score"] - ro["rob_score"]).max() < 3e-3
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    for t in range(len(ref["nV_set"])):
        a, b = res["nV_set"][t], ref["nV_set"][t]
        k = len(ref["signal_ev"])
        assert np.all(_abs_cos(a[:, :k], b[:, :k]) > 1 - 3e-3)
        assert np.allclose(res["nL_set"][t][:k], ref["nL_set"][t][:k], rtol=3e-4)
        assert np.allclose(res["nL_set"][t], ref["nL_set"][t], rtol=2e-3)


def test_device_sampler_equals_host_sampler(ctx):
    """*_seeded session calls draw the sample on the device with the same keyed permutation the host function
    evaluates: both routes must give bit-identical results."""
    X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d_dev = api.make_draws_native(X, seed=21)
    d_host = api.make_draws_native(X, seed=21, host_sampler=True)
    a = api.sclens(X, draws=d_dev, n_perturb=4, ctx=ctx, max_search_iters=6)
    b = api.sclens(X, draws=d_host, n_perturb=4, ctx=ctx, max_search_iters=6)
    assert a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
    for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
        assert np.array_equal(t1, t2)
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
    assert np.array_equal(a["L"], b["L"])  # the whole path is bitwise reproducible


def test_native_draws_agree_with_oracle(ctx):
    """End-to-end with the library's own draw generators (what bench.py times), oracle fed the identical draws."""
    X = synth_counts(600, 250, seed=2, C=4, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=5, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=5, null_tol=O.NULL_DROP)
    res = api.sclens(X, draws=api.make_draws_native(X, seed=5), n_perturb=5, ctx=ctx)
    assert len(res["signal_ev"]) == len(ref["signal_ev"])
    assert res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"]
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.abs(res["robustness_scores"]["rob_score"] - ref["robustness_scores"]["rob_score"]).max() < 3e-3


@pytest.mark.parametrize("implicit", [0, 1])
@pytest.mark.parametrize("N,M", [(300, 500), (600, 250)])
def test_partial_eigensolver_matches_full_solver(ctx, N, M, implicit, opt):
    """Ensemble members via the leading-eigenpair subspace iteration vs the full eigensolver: same decisions,
    eigenvalues to 3e-4, signal eigenvectors to |cos| >= 1 - 3e-3. implicit = 1: the iteration applies the Gram matrix as
    two passes over the scaled matrix and never forms it (the default from n = 16 000)."""
    opt(implicit_min_n=1 if implicit else 1000000000)
    X = synth_counts(N, M, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=9)
    a = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, keep_intermediates=True, max_search_iters=6, partial_eig=True)
    b = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, keep_intermediates=True, max_search_iters=6, partial_eig=False)
    assert a["partial_eig"][0] == 5 and a["partial_eig"][1] == 0, a["partial_eig"]  # used, no fallback
    assert b["partial_eig"] == (0, 0)
    k = len(b["signal_ev"])
    for t in range(5):
        assert np.allclose(a["nL_set"][t][:k], b["nL_set"][t][:k], rtol=3e-4)  # signals
        assert np.allclose(a["nL_set"][t], b["nL_set"][t], rtol=2e-3)  # bulk-edge tail (not part of the reference's outputs)
        assert np.all(_abs_cos(a["nV_set"][t][:, :k], b["nV_set"][t][:, :k]) > 1 - 3e-3)
    assert np.array_equal(a["robustness_scores"]["a_b"], b["robustness_scores"]["a_b"])
    assert np.abs(a["robustness_scores"]["rob_score"] - b["robustness_scores"]["rob_score"]).max() < 3e-3
    assert np.array_equal(a["sig_id"], b["sig_id"])
    # members whose matching picked a tail vector (index >= k) were solved again with the gap-aware tail target: every picked
    # column of such a member then agrees with the full solver's as a signal column does
    assert isinstance(a["tail_redo"], list) and b["tail_redo"] == []
    ab = a["robustness_scores"]["a_b"]
    for t in a["tail_redo"]:
        assert np.any(ab[:, t] >= k)
        cols = np.unique(ab[:, t])
        assert np.all(_abs_cos(a["nV_set"][t][:, cols], b["nV_set"][t][:, cols]) > 1 - 2e-2)


@pytest.mark.parametrize("implicit", [0, 1])
@pytest.mark.parametrize("N,M,sd", [(300, 500, 1.5), (600, 250, 1.5), (600, 250, 0.8)])
def test_certified_ensemble_tail_gives_the_outputs_of_the_full_solver(ctx, N, M, sd, implicit, opt):
    """ensemble_tail = "certified" (the default from order 16 000): the eigenpairs k .. min_pc-1 of a member are left unconverged and
    the matching (scLENS.jl:788) is accepted per member only with the proof that no vector outside the first k could have been picked
    (session_robustness: best_k^2 > 1 - sum_{j<k} c_ij^2); members without it are solved again with the tail converged. Every OUTPUT
    equals the full solver's: a_b, sig_id exactly, scores and signal pairs to the usual tolerances; the certificate the library
    reports equals the one recomputed here from the vectors (weak markers, sd = 0.8: members without proof do occur)."""
    opt(implicit_min_n=1 if implicit else 1000000000)
    X = synth_counts(N, M, seed=1, C=5, marker_frac=0.2, marker_sd=sd)
    d = api.make_draws_native(X, seed=9)
    kw = dict(draws=d, n_perturb=5, ctx=ctx, keep_intermediates=True, max_search_iters=6)
    a = api.sclens(X, partial_eig=True, ensemble_tail="certified", **kw)
    b = api.sclens(X, partial_eig=False, **kw)
    if "robustness_scores" not in b:
        pytest.skip("no signal in this matrix")
    assert a["ensemble_tail"] == "certified" and b["ensemble_tail"] == "converged"
    assert a["partial_eig"][0] >= 5 and a["partial_eig"][1] == 0, a["partial_eig"]
    k = len(b["signal_ev"])
    assert k >= 1
    ab = a["robustness_scores"]["a_b"]
    assert np.array_equal(ab, b["robustness_scores"]["a_b"])
    assert np.abs(a["robustness_scores"]["rob_score"] - b["robustness_scores"]["rob_score"]).max() < 3e-3
    assert np.array_equal(a["sig_id"], b["sig_id"])
    nV = b["signal_evec"].astype(np.float64)
    for t in range(5):
        assert np.allclose(a["nL_set"][t][:k], b["nL_set"][t][:k], rtol=3e-4)
        assert np.all(_abs_cos(a["nV_set"][t][:, :k], b["nV_set"][t][:, :k]) > 1 - 3e-3)
        # the certificate, from the full solver's vectors: where it holds the member was not solved again, and vice versa up to the margin
        c = np.abs(nV.T @ b["nV_set"][t][:, :k].astype(np.float64))
        proof = np.all(c.max(axis=1) ** 2 > 1.0 - (c ** 2).sum(axis=1) + 2.5e-2)  # the library's margin is 2e-2
        if proof:
            assert t not in a["tail_redo"], (t, a["tail_redo"])
        if t in a["tail_redo"]:  # solved again: every picked column agrees with the full solver's
            cols = np.unique(ab[:, t])
            assert np.all(_abs_cos(a["nV_set"][t][:, cols], b["nV_set"][t][:, cols]) > 1 - 2e-2)
    print(f"[certified tail {N}x{M} sd {sd} implicit {implicit}] k {k}, members solved again: {a['tail_redo']}")


def test_matching_certificate_of_the_session_equals_its_definition(ctx):
    """session_robustness with "chefsi_tail_free": match_uncertain:t against the definition evaluated here from the downloaded vectors
    -- member t is certain iff every signal i has best_k^2 > 1 - sum_{j<k} c_ij^2 (+ 2e-2) with c = nV' V_t[:, :k] and its argmax over
    all min_pc columns lies among the first k. Session A: members of its own matrix (the signals survive the perturbation: certain).
    Session B, another matrix of the same shape: A's members imported into its slots -- B's signals have nothing to do with them, so
    no proof can exist: both answers occur and both equal the definition."""
    XA = api._csc_f32(synth_counts(600, 250, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    XB = api._csc_f32(synth_counts(600, 250, seed=2, C=4, marker_frac=0.25, marker_sd=1.5))
    d = api.make_draws_native(XA, seed=9)
    ref = api.sclens(XA, draws=d, n_perturb=2, ctx=ctx, max_search_iters=6, partial_eig=False, streams=1)
    k, min_pc = len(ref["signal_ev"]), ref["min_pc"]
    m_pert = int(round((1 - ref["p_"]) * XA.shape[0] * XA.shape[1]))

    def definition(nV, V, kk):
        c = np.abs(nV.T @ V)
        slack = c[:, :kk].max(axis=1) ** 2 - (1.0 - (c[:, :kk] ** 2).sum(axis=1) + 2e-2)
        return int(np.any(c.argmax(axis=1) >= kk) or np.any(slack <= 0)), bool(np.any(np.abs(slack) < 1e-4))

    sesA = api.Session(ctx, XA, api._resolve(d.z_idx1), api._resolve(d.z_idx2))
    sesB = ctxB = None
    buf = 0
    try:
        sesA.set_int("chefsi", 1)
        sesA.null_spectrum(api._resolve(d.X_r))
        sesA.data_spectrum(True)
        nVA = sesA.signal_vectors(k).astype(np.float64)
        sesA.set_int("chefsi_tail_free", 1)
        assert sesA.get_int("chefsi_tail_free") == 1
        V = []
        for t in range(2):
            _, c = sesA.perturb_seeded(t, api.sample_seed_for(d.sample_seed, "perturb", t), m_pert, min_pc)
            assert c == min_pc
            V.append(sesA.get_perturbed(t, c).astype(np.float64))
        sesA.robustness(k, 2)
        gotA = [sesA.get_int(f"match_uncertain:{t}") for t in range(2)]
        assert sesA.get_int("match_uncertain_count") == sum(gotA)
        # session B: its own signals, A's members
        dB = api.make_draws_native(XB, seed=10)
        ctxB = Context(ctx.device)  # one live session per context
        sesB = api.Session(ctxB, XB, api._resolve(dB.z_idx1), api._resolve(dB.z_idx2))
        sesB.null_spectrum(api._resolve(dB.X_r))
        LB, _ = sesB.data_spectrum(True)
        kB = min(k, 3)
        nVB = sesB.signal_vectors(kB).astype(np.float64)
        buf = ctx.malloc(4 * min_pc * sesA.slot_ld())
        for t in range(2):
            sesA.export_slot(t, min_pc, buf)
            sesB.import_slot(t, min_pc, min_pc, buf)
        sesB.robustness(kB, 2)
        gotB = [sesB.get_int(f"match_uncertain:{t}") for t in range(2)]
        assert sesB.get_int("match_uncertain_count") == sum(gotB)
    finally:
        if buf:
            ctx.free(buf)
        sesA.close()
        if sesB is not None:
            sesB.close()
        if ctxB is not None:
            ctxB.close()
    wantA = [definition(nVA, V[t], k) for t in range(2)]
    wantB = [definition(nVB, V[t], kB) for t in range(2)]
    print(f"[matching certificate] k {k}: own members {gotA} (definition {[w for w, _ in wantA]}); foreign members {gotB} (definition {[w for w, _ in wantB]})")
    for got, want in ((gotA, wantA), (gotB, wantB)):
        for t in range(2):
            assert want[t][1] or got[t] == want[t][0], (t, got, want)
    assert gotA == [0, 0] and gotB == [1, 1]


def test_partial_eigensolver_tail_gap_option(ctx):
    """session option "chefsi_tail_gap_milli": the tail pairs k .. min_pc-1 of an ensemble member held to a gap-aware residual
    target, residual <= 0.05 (theta_q - theta_block_end), i.e. sin(angle to the true vector) <= 0.05 (what api.sclens switches on
    for a member whose tail vector gets matched). Every column whose eigenvalue is separated from its neighbours by 1 % then agrees
    with the full solver's to |cos| >= 1 - 5e-3, tail columns included."""
    X = api._csc_f32(synth_counts(600, 250, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=9)
    ref = api.sclens(X, draws=d, n_perturb=2, ctx=ctx, keep_intermediates=True, max_search_iters=6, partial_eig=False, streams=1)
    k, min_pc = len(ref["signal_ev"]), ref["min_pc"]
    ses = api.Session(ctx, X, api._resolve(d.z_idx1), api._resolve(d.z_idx2))
    try:
        ses.set_int("chefsi", 1)
        ses.null_spectrum(api._resolve(d.X_r))
        ses.data_spectrum(True)
        ses.signal_vectors(k)
        m_pert = int(round((1 - ref["p_"]) * X.shape[0] * X.shape[1]))
        outs = {}
        for milli in (0, 50):
            ses.set_int("chefsi_tail_gap_milli", milli)
            assert ses.get_int("chefsi_tail_gap_milli") == milli
            nl, c = ses.perturb_seeded(0, api.sample_seed_for(d.sample_seed, "perturb", 0), m_pert, min_pc)
            outs[milli] = (nl, ses.get_perturbed(0, c))
        assert ses.get_int("chefsi_used") == 2 and ses.get_int("chefsi_fallback") == 0
    finally:
        ses.close()
    nLf, Vf = ref["nL_set"][0], ref["nV_set"][0]
    for milli, (nl, V) in outs.items():
        c = min(V.shape[1], Vf.shape[1])
        assert c > k
        assert np.allclose(nl[:c], nLf[:c], rtol=2e-3)
        cos = _abs_cos(V[:, :c], Vf[:, :c])
        assert np.all(cos[:k] > 1 - 3e-3), (milli, cos)
        if milli:
            lam = np.asarray(nLf[: c + 1], dtype=np.float64)
            sep = np.minimum(np.abs(np.diff(lam, prepend=np.inf)), np.abs(np.diff(lam, append=-np.inf)))[:c] > 1e-2 * lam[:c]
            assert np.all(cos[sep] > 1 - 5e-3), (cos, sep)


def test_fused_dense_write_gives_the_same_bits(ctx, opt):
    """k_dense_fused (one pass per gene: background + stored entries + sampled candidates through LDS chunks) against the separate
    fill + scatter kernels (context option dense_fused = 0): the dense scaled matrices are the same, so is everything downstream --
    counts-only patterns (data / null), and the union pattern with its unordered candidate tail (ensemble members, full solver)."""
    X = api._csc_f32(synth_counts(900, 260, seed=3, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=17)
    out = []
    for flag in ("1", "0"):
        opt(dense_fused=int(flag))
        out.append(api.sclens(X, draws=d, n_perturb=3, ctx=ctx, keep_intermediates=True, max_search_iters=5, partial_eig=False, streams=1))
        S, _ = api.logn_scale(X, "mean", inline_f64=True, ctx=ctx)
        out[-1]["S"] = S
    a, b = out
    assert np.array_equal(a["S"], b["S"])
    assert np.array_equal(a["L"], b["L"]) and np.array_equal(a["signal_evec"], b["signal_evec"])
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
    for t in range(3):
        assert np.array_equal(a["nV_set"][t], b["nV_set"][t]) and np.array_equal(a["nL_set"][t], b["nL_set"][t])


def test_two_streams_give_identical_results(ctx):
    """streams=2 (two worker sessions on separate HIP streams, speculative search rounds of 2) must reproduce the
    serial run bit for bit: same samples per iteration, deterministic kernels, results consumed in order."""
    X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=13)
    a = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=1)
    b = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=2)
    assert a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
    for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
        assert p1 == p2 and np.array_equal(t1, t2)
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
    assert np.array_equal(a["sig_id"], b["sig_id"])
    for t in range(5):
        assert np.array_equal(a["nL_set"][t], b["nL_set"][t])
    for kw in ({"streams": 3}, {"streams": None}):  # None = the host's own choice by order
        c = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, **kw)
        assert a["p_"] == c["p_"] and a["n_search"] == c["n_search"]
        for (p1, t1), (p3, t3) in zip(a["search_trace"], c["search_trace"]):
            assert p1 == p3 and np.array_equal(t1, t3)
        assert np.array_equal(a["L"], c["L"])
        assert np.array_equal(a["robustness_scores"]["b_"], c["robustness_scores"]["b_"])


def test_get_denoised_df(run_pair, ctx):
    """the denoised reconstruction (scLENS.jl:889-931) on the device vs the oracle, both from the oracle's
    sclens() result (so the comparison isolates this function)."""
    X, ref, res = run_pair
    want = O.get_denoised(ref)
    got = api.get_denoised_df({"sig_id": ref["sig_id"], "gene_basis": ref["gene_basis"], "pca_n1": ref["pca_n1"],
                               "rec_vals": ref["rec_vals"]}, ctx=ctx)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-4 * want.max()
    assert np.allclose(got.sum(axis=1), np.ravel(ref["rec_vals"]["TGC"]).mean(), rtol=1e-4)
    # and end to end from the device result (eigenvector signs cancel in pca_n1 * gene_basis)
    got2 = api.get_denoised_df(res, ctx=ctx)
    assert np.abs(got2 - want).max() < 2e-2 * want.max()


@pytest.mark.parametrize("N,M,dense_frac", [(151, 260, 0.0), (300, 500, 0.1), (420, 130, 0.3), (64, 40, 0.6)])
def test_logn_scale_dropin_mean_and_median(ctx, N, M, dense_frac):
    """sclens_hip_scale_csc_f32 (logn_scale(pre_scale(x)), scLENS.jl:650-654) against the oracle for both centrings;
    `dense_frac` of the genes are expressed in most cells, so their medians are non-zero (odd and even N, so one and
    two middle order statistics), and the inline Float64 twin returns the reference's rec_vals."""
    X = synth_counts(N, M, seed=N + M, C=4, marker_frac=0.3, marker_sd=1.5).toarray()
    rng = np.random.default_rng(5)
    dense_genes = rng.choice(M, size=int(dense_frac * M), replace=False)
    X[:, dense_genes] += rng.poisson(3.0, size=(N, len(dense_genes))).astype(X.dtype)
    Xs = O._as_csc_f32(X)
    want = O.logn_scale(O.pre_scale(Xs))
    got = api.logn_scale(Xs, "mean", ctx=ctx)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    want_m = O.logn_scale_median(O.pre_scale(Xs))
    got_m = api.logn_scale(Xs, "median", ctx=ctx)
    assert np.abs(got_m - want_m).max() < 2e-5 * np.abs(want_m).max()
    if dense_frac > 0:
        med = np.median(np.asarray(O.pre_scale(Xs).todense()), axis=0)
        assert (med > 0).sum() >= len(dense_genes) // 2  # the radix-select branch was exercised
    want_i, rec = O.scale_main(Xs)
    got_i, rec_d = api.logn_scale(Xs, "mean", inline_f64=True, ctx=ctx)
    assert np.abs(got_i - want_i).max() < 2e-6 * np.abs(want_i).max()
    for key in ("TGC", "mat2_mean", "mat2_std", "norm_tgc", "cent_"):
        assert np.allclose(np.ravel(rec_d[key]), np.ravel(rec[key]), rtol=1e-9, atol=1e-12), key


@pytest.mark.parametrize("N,M", [(300, 500), (600, 250)])
def test_sclens_median_centering(ctx, N, M):
    """centering="median" (scLENS.jl:653-654) end to end against the oracle on the same draws."""
    X = synth_counts(N, M, seed=1, C=5, marker_frac=0.2, marker_sd=1.5).toarray()
    rng = np.random.default_rng(11)
    dg = rng.choice(M, size=M // 10, replace=False)
    X[:, dg] += rng.poisson(2.0, size=(N, len(dg))).astype(X.dtype)  # genes with non-zero medians
    d = api.make_draws(X, seed=7, p_th_trials=300)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=5, keep_intermediates=True, null_tol=O.NULL_DROP, centering="median")
    res = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, keep_intermediates=True, centering="median", streams=2)
    assert res["rec_vals"] == {} and ref["rec_vals"] == {}
    assert np.abs(res["L"] - ref["L"]).max() < 2e-4 * ref["L"].max()
    assert abs(res["lambda_c"] - ref["lambda_c"]) < 2e-4 * ref["lambda_c"]
    k = len(ref["signal_ev"])
    assert len(res["signal_ev"]) == k > 0
    assert np.allclose(res["signal_ev"], ref["signal_ev"], rtol=2e-4)
    assert np.all(_abs_cos(res["signal_evec"], ref["signal_evec"]) > 1 - 2e-3)
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.abs(d1 - d2).max() < 3e-3
    rr, ro = res["robustness_scores"], ref["robustness_scores"]
    assert np.array_equal(rr["a_b"], ro["a_b"])
    assert np.abs(rr["rob_score"] - ro["rob_score"]).max() < 3e-3
    assert np.array_equal(res["sig_id"], ref["sig_id"])


def test_sclens_unsupported_centering_string_runs_the_reference_fallback(ctx, capsys):
    """Any other `centering` string (scLENS.jl:655-657): the reference prints a warning and runs
    scaled_gdata(norm_l(scaled_gdata(x, "mean")), "cent") -- the mean branch's function of x evaluated in Float32 -- and leaves
    rec_vals empty (:697-698). Device: same warning, mean path, empty rec_vals; against the oracle's restatement of that branch."""
    X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws(X, seed=7, p_th_trials=300)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=4, null_tol=O.NULL_DROP, centering="mode", max_search_iters=6)
    res = api.sclens(X, draws=d, n_perturb=4, ctx=ctx, centering="mode", streams=2, max_search_iters=6)
    assert "not supported in the current algorithm" in capsys.readouterr().out
    assert res["rec_vals"] == {} and ref["rec_vals"] == {}
    assert np.abs(res["L"] - ref["L"]).max() < 2e-4 * ref["L"].max()
    k = len(ref["signal_ev"])
    assert len(res["signal_ev"]) == k > 0 and np.allclose(res["signal_ev"], ref["signal_ev"], rtol=2e-4)
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    assert np.array_equal(res["sig_id"], ref["sig_id"])


def test_late_candidate_attachment_equals_upfront_session(ctx):
    """A session created from the counts alone, with the union pattern (counts + zero candidates) attached after the
    first decompositions (sclens_hip_pattern_create / _session_set_pattern / adopt(4)), must give bit-identical search
    and ensemble results to a session that was created with the candidates."""
    from sclens_amd._lib import Context

    X = api._csc_f32(synth_counts(260, 410, seed=4, C=4, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=17, host_sampler=True)
    z1, z2 = d.z_idx1, d.z_idx2
    N, M = X.shape
    m = int(round(0.004 * N * M))
    idx = d.sampler("search", 0, len(z1), m)

    def run(late):
        ses = api.Session(ctx, X) if late else api.Session(ctx, X, z1, z2)
        c2 = Context(ctx.device)
        w = ses.clone(c2)
        try:
            L, _ = ses.data_spectrum()
            _, r = w.binary_basis()
            ses.signal_vectors(3)
            if late:
                with pytest.raises(api.SclensHipError):  # no candidates yet: the sample is out of range
                    w.search_step(idx, int(round(r / 2)))
                pat = api.Pattern(c2, X, z1, z2)
                ses.set_pattern(pat)
                pat.close()  # ownership has moved to the session
                w.adopt(ses, 4)
            ses.adopt(w, 1)
            d5a, _ = w.search_step(idx, int(round(r / 2)))
            d5b, _ = ses.search_step_seeded(123, m, int(round(r / 2)))
            nLp, nc = ses.perturb(0, idx, 4)
            V = ses.get_perturbed(0, nc)
            return L, d5a, d5b, nLp, V
        finally:
            w.close()
            c2.close()
            ses.close()

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_midsize_parity_native_draws(ctx):
    """A size with ten reduction panels and multi-work-unit symv launches (n = 1200) against the oracle, library draws
    (what bench.py times), three concurrent streams: decisions exact, spectra and scores within the fp32 tolerances."""
    X = synth_counts(1200, 2400, seed=3, C=6, marker_frac=0.1, marker_sd=1.2)
    d = api.make_draws_native(X, seed=31, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=6, null_tol=O.NULL_DROP, keep_intermediates=True)
    res = api.sclens(X, draws=api.make_draws_native(X, seed=31, async_null=True, async_candidates=True), n_perturb=6, ctx=ctx,
                     streams=3, keep_intermediates=True)
    assert np.abs(res["L"] - ref["L"]).max() < 2e-4 * ref["L"].max()
    k = len(ref["signal_ev"])
    assert len(res["signal_ev"]) == k > 0
    assert np.allclose(res["signal_ev"], ref["signal_ev"], rtol=2e-4)
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.abs(d1 - d2).max() < 3e-3
    gaps = np.minimum(np.abs(np.diff(ref["signal_ev"], prepend=np.inf)), np.abs(np.diff(ref["signal_ev"], append=ref["lambda_c"])))
    sep = gaps > 0.02 * ref["signal_ev"]
    assert np.all(_abs_cos(res["signal_evec"], ref["signal_evec"])[sep] > 1 - 2e-3)
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.abs(res["robustness_scores"]["rob_score"] - ref["robustness_scores"]["rob_score"]).max() < 5e-3
    assert res["partial_eig"][0] + res["partial_eig"][1] == 6


@pytest.mark.parametrize("world,streams", [(2, 2), (3, 1), (4, 2)])
def test_multi_rank_sclens_equals_single_rank(ctx, world, streams):
    """api.sclens(shard=...) with `world` ranks simulated as threads of this process (one context each; collectives by
    devutil.ThreadShard): the data / null / binarised decompositions are spread over the ranks and their results
    broadcast, search rounds of world x streams evaluations, ensemble members t % world. Rank 0's result must equal the
    single-rank run bit for bit (same kernels on the same inputs; results consumed in iteration order)."""
    import threading

    from devutil import ThreadShard
    from sclens_amd._lib import Context

    X = api._csc_f32(synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=13)
    ref = api.sclens(X, draws=d, n_perturb=6, ctx=ctx, streams=streams)
    group = ThreadShard.Group(world)
    out, err = [None] * world, [None] * world

    def work(r):
        c = Context(ctx.device)
        try:
            out[r] = api.sclens(X, draws=d, n_perturb=6, ctx=c, streams=streams, shard=ThreadShard(group, r))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.bar.abort()
        finally:
            c.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(e is None for e in err)
    res = out[0]
    assert np.array_equal(res["L"], ref["L"]) and res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"]
    for (p1, t1), (p2, t2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.array_equal(t1, t2)
    assert np.array_equal(res["signal_evec"], ref["signal_evec"])
    assert np.array_equal(res["robustness_scores"]["b_"], ref["robustness_scores"]["b_"])
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.array_equal(res["gene_basis"], ref["gene_basis"])
    for t in range(6):
        assert np.array_equal(res["nL_set"][t], ref["nL_set"][t])
    for r in range(1, world):  # the other ranks return the shared part of the result
        assert np.array_equal(out[r]["L"], ref["L"]) and out[r]["p_"] == ref["p_"] and "pca" not in out[r]


def test_eight_rank_rehearsal_of_the_whole_call(ctx):
    """SURVEY 8e-i/ii/iv at the world size of the target node (8 x MI355X), rehearsed on one GPU with thread ranks: 20 ensemble members
    spread 3 + 3 + 3 + 3 + 2 + 2 + 2 + 2 over the ranks (t mod 8), one gather at the end; search rounds of 8 evaluations, more than
    the search takes, so that the evaluations past the stopping one are computed and DISCARDED (the overshoot path of
    consume_search_round); the data / null / binarised decompositions on ranks 0 / 1 / 2 with their results broadcast. Rank 0's result
    equals the single-rank run bit for bit; every rank returns the shared decisions (scLENS.jl:725-761, :771-778)."""
    import threading

    from devutil import ThreadShard
    from sclens_amd._lib import Context

    world, P = 8, 20
    X = api._csc_f32(synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=13)
    ref = api.sclens(X, draws=d, n_perturb=P, ctx=ctx, streams=1)
    assert ref["n_search"] % world != 0  # the last round is cut short by the stop rule: results beyond it are discarded
    group = ThreadShard.Group(world)
    out, err = [None] * world, [None] * world

    def work(r):
        c = Context(ctx.device)
        try:
            out[r] = api.sclens(X, draws=d, n_perturb=P, ctx=c, streams=1, shard=ThreadShard(group, r))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.bar.abort()
        finally:
            c.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(e is None for e in err)
    res = out[0]
    assert np.array_equal(res["L"], ref["L"]) and res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"]
    for (p1, t1), (p2, t2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.array_equal(t1, t2)
    assert np.array_equal(res["signal_evec"], ref["signal_evec"])
    assert np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"])
    assert np.array_equal(res["robustness_scores"]["b_"], ref["robustness_scores"]["b_"])
    assert np.array_equal(res["sig_id"], ref["sig_id"]) and np.array_equal(res["gene_basis"], ref["gene_basis"])
    for t in range(P):
        assert np.array_equal(res["nL_set"][t], ref["nL_set"][t])
    for r in range(1, world):
        assert np.array_equal(out[r]["L"], ref["L"]) and out[r]["p_"] == ref["p_"] and out[r]["n_search"] == ref["n_search"]


def _run_ranks(X, d, world, device, **kw):
    import threading

    from devutil import ThreadShard
    from sclens_amd._lib import Context

    group = ThreadShard.Group(world)
    out, err = [None] * world, [None] * world

    def work(r):
        c = Context(device)
        try:
            out[r] = api.sclens(X, draws=d, ctx=c, shard=ThreadShard(group, r), **kw)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.bar.abort()
        finally:
            c.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(e is None for e in err)
    return out


def test_multi_rank_corner_cases(ctx):
    """Eight ranks with one stream each on a cells > genes matrix (more ranks than first decompositions, more search slots
    than iterations), and a matrix without signals on three ranks (every rank must take the early return of
    scLENS.jl:780-784 together)."""
    import scipy.sparse as sp

    X = api._csc_f32(synth_counts(600, 250, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=3)
    ref = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=1)
    out = _run_ranks(X, d, 8, ctx.device, n_perturb=5, streams=1)
    assert np.array_equal(out[0]["L"], ref["L"]) and out[0]["p_"] == ref["p_"] and out[0]["n_search"] == ref["n_search"]
    assert np.array_equal(out[0]["sig_id"], ref["sig_id"])
    assert np.array_equal(out[0]["robustness_scores"]["b_"], ref["robustness_scores"]["b_"])
    assert all(o["p_"] == ref["p_"] for o in out)
    rng = np.random.default_rng(0)
    Xn = sp.csc_matrix(rng.poisson(0.12, size=(150, 260)).astype(np.float32))
    Xn = Xn[np.asarray(Xn.sum(axis=1)).ravel() > 0][:, np.asarray(Xn.sum(axis=0)).ravel() > 0]
    Xn = api._csc_f32(Xn[:, np.diff(Xn.tocsc().indptr) >= 2].tocsc())
    dn = api.make_draws_native(Xn, seed=1)
    refn = api.sclens(Xn, draws=dn, n_perturb=3, ctx=ctx, max_search_iters=5)
    outn = _run_ranks(Xn, dn, 3, ctx.device, n_perturb=3, max_search_iters=5, streams=2)
    assert len(outn[0].get("signal_ev", [])) == len(refn.get("signal_ev", []))
    assert np.array_equal(outn[0]["L"], refn["L"]) and all(o["p_"] == refn["p_"] for o in outn)


def test_default_call_hands_its_device_memory_back(ctx):
    """`keep_warm=False` (the default) must leave nothing of the call on the device -- the caller's context outlives the call, so
    its grow-only workspaces go back to the pool first (release_scratch "everything") and the pool's idle blocks to the driver after
    that (sclens_hip_trim): the library's live bytes are what they were before the call, its idle cache is empty. With keep_warm=True the
    blocks stay cached for the next call of the same shape."""
    import ctypes as C

    from sclens_amd._lib import Context

    def stats():
        c, l, h, m = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        assert ctx.lib.sclens_hip_pool_stats(ctx.device, C.byref(c), C.byref(l), C.byref(h), C.byref(m)) == 0
        return c.value, l.value

    X = api._csc_f32(synth_counts(900, 400, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    c2 = Context(ctx.device)
    try:
        ctx.trim_pool()
        _, live0 = stats()
        api.sclens(X, draws=api.make_draws_native(X, seed=3), n_perturb=3, ctx=c2, streams=2, keep_warm=True)
        cached_w, live_w = stats()
        assert cached_w > 0 and live_w > live0  # warm: the call's blocks wait in the pool, the context keeps its workspaces
        api.sclens(X, draws=api.make_draws_native(X, seed=3), n_perturb=3, ctx=c2, streams=2)
        cached, live = stats()
        assert cached == 0 and live == live0, (cached, live, live0)
    finally:
        c2.close()


def test_shared_inverse_iteration_workspaces_give_the_same_bits(ctx):
    """stein_shared = 1: the six [n][batch] workspaces of the inverse iteration exist once per device and the contexts of a call take turns
    (an event recorded after one user's last launch, awaited by the next): same bits as the per-context workspaces, with two and three
    streams, and nothing of the block left on the device after a default (keep_warm=False) call."""
    import ctypes as C

    from sclens_amd._lib import Context

    def live():
        c, l, h, m = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        assert ctx.lib.sclens_hip_pool_stats(ctx.device, C.byref(c), C.byref(l), C.byref(h), C.byref(m)) == 0
        return c.value, l.value

    X = synth_counts(600, 900, seed=2, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=17)
    a = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=2)
    c2 = Context(ctx.device)
    c2.set_option("stein_shared", 1)
    try:
        ctx.trim_pool()
        _, live0 = live()
        for streams in (2, 3, 1):
            b = api.sclens(X, draws=d, n_perturb=5, ctx=c2, streams=streams)
            assert a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
            for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
                assert p1 == p2 and np.array_equal(t1, t2)
            assert np.array_equal(a["L"], b["L"]) and np.array_equal(a["signal_evec"], b["signal_evec"])
            assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
            assert np.array_equal(a["sig_id"], b["sig_id"])
            cached, l1 = live()
            assert cached == 0 and l1 == live0, (cached, l1, live0)
    finally:
        c2.close()


def _sweep_case(sweep_seed, index):
    """parameters of case `index` of `scripts/fuzz_parity.py <cases> <sweep_seed>` (the same draws in the same order)"""
    rng = np.random.default_rng(sweep_seed)
    for c in range(index + 1):
        N, M = int(rng.integers(90, 420)), int(rng.integers(90, 420))
        C = int(rng.integers(2, 7))
        seed = int(rng.integers(1, 10 ** 6))
        cent = "median" if rng.random() < 0.25 else "mean"
        streams = int(rng.integers(1, 4))
        mf, ms = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.8, 1.8))
    return N, M, C, seed, cent, streams, mf, ms


@pytest.mark.parametrize("sweep_seed,index,shape", [(41, 100, (292, 260)), (41, 121, (259, 233)), (51, 117, (371, 329))])
def test_small_genuine_eigenvalue_of_a_nearly_square_matrix_stays_in_the_binarised_basis(ctx, sweep_seed, index, shape):
    """cells > genes and nearly square: the binarised matrix has a GENUINE eigenvalue of ~7e-6 lambda_max (no structural zero on this
    side). The reference's `L .> 0` (scLENS.jl:495) keeps it; a positivity floor of 8 sqrt(n) eps32 lambda_max (7.7e-6 at n = 260) dropped
    it from Vr2 but not from the perturbed matrices' bases, one ~0 entry entered d_arr (:742) and the search ended an evaluation early --
    the cases the random sweeps of rounds 5 / 6 turned up (profiles/r06_fuzz_null_floor.md). Every evaluation's statistic and the search
    length against the oracle."""
    N, M, C, seed, cent, streams, mf, ms = _sweep_case(sweep_seed, index)
    assert (N, M) == shape and cent == "mean"
    X = synth_counts(N, M, seed=seed, C=C, marker_frac=mf, marker_sd=ms, min_genes_per_cell=5, min_cells_per_gene=4)
    d = api.make_draws_native(X, seed=seed, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=4, null_tol=O.NULL_DROP, max_search_iters=25)
    res = api.sclens(X, draws=d, n_perturb=4, ctx=ctx, streams=streams, max_search_iters=25)
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.abs(d1 - d2).max() < 1e-3, (p1, d1, d2)
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"])
