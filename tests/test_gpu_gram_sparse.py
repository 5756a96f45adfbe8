"""-m gpu: the Gram matrix of a COUNT-VALUED scaled matrix from its sparse structure (csrc/gram_sparse.hip,
sclens_hip_gram_counts_f32 mode 1, context option gram_sparse) against the float64 oracle, against the dense product it replaces, and
inside a whole sclens() call. The scaled matrix is sparse + rank two (the identity of scLENS.jl:601-603 / :688-690); the sparse part is
contracted with fp32 products accumulated in 64-bit fixed point (order-independent: bitwise reproducible), the rank-two terms in fp64."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,seed", [(2000, 300, 3), (1500, 1300, 4), (9000, 2600, 5)])
@pytest.mark.parametrize("f32path", [True, False])
def test_sparse_structured_gram_matches_float64(ctx, N, M, seed, f32path):
    X = api._csc_f32(synth_counts(N, M, seed=seed, C=4, marker_frac=0.15, marker_sd=1.3))
    S = O.logn_scale(O.pre_scale(X)) if f32path else O.scale_main(X)[0]
    ref = O.wishart_matrix(S, 2)  # S'S / M, float64
    got = api._gram_counts(X, 1, f32path=f32path, ctx=ctx).astype(np.float64)
    dense = api._gram_counts(X, 0, f32path=f32path, ctx=ctx).astype(np.float64)
    scale = np.abs(ref).max()
    e_sparse, e_dense = np.abs(got - ref).max() / scale, np.abs(dense - ref).max() / scale
    print(f"[gram_sparse {N}x{M} f32path={f32path}] max error / largest entry: sparse {e_sparse:.2e}, dense product {e_dense:.2e}")
    # the contraction is exact to 2^-41 per product; what remains is the fp32 rounding of the operands and of the result
    assert e_sparse < 1e-6 and e_sparse < 3 * e_dense + 2e-7
    assert np.array_equal(got, got.T)  # exactly symmetric
    again = api._gram_counts(X, 1, f32path=f32path, ctx=ctx).astype(np.float64)
    assert np.array_equal(got, again)  # integer accumulation: the order of the atomics does not show


def test_sparse_structured_gram_of_a_binarised_matrix(ctx):
    N, M = 3000, 700
    X = api._csc_f32(synth_counts(N, M, seed=9, C=5, marker_frac=0.2, marker_sd=1.2))
    P = sp.csc_matrix((np.ones_like(X.data), X.indices, X.indptr), shape=X.shape, dtype=np.float32)
    ref = O.wishart_matrix(O.logn_scale(O.pre_scale(P)), 2) * (M / N)  # divisor = cells, as the sparsity search takes it (Appendix A8)
    got = api._gram_counts(X, 1, binary=True, divisor=N, ctx=ctx).astype(np.float64)
    assert np.abs(got - ref).max() < 1e-6 * np.abs(ref).max()


def test_sclens_with_the_sparse_structured_gram(ctx, opt):
    """a whole call with every Gram product of a count-valued or binarised matrix taken from the sparse structure: the decisions and
    results of the default path (which forms the scaled matrix and multiplies)"""
    X = api._csc_f32(synth_counts(900, 400, seed=1, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=23)
    ref = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=1)
    opt(gram_sparse=1, gram_bits=0)
    res = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=1)
    assert res["gram_sparse_used"] >= 1 + res["n_search"]  # (counted from the binarised basis on, like gram_bits_used)
    assert np.abs(res["L"] - ref["L"]).max() < 2e-5 * ref["L"].max()
    assert len(res["signal_ev"]) == len(ref["signal_ev"]) > 0
    assert res["n_search"] == ref["n_search"] and res["p_"] == ref["p_"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.abs(d1 - d2).max() < 2e-3
    assert np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"])
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.abs(res["robustness_scores"]["rob_score"] - ref["robustness_scores"]["rob_score"]).max() < 3e-3


def test_sparse_form_is_chosen_by_its_multiply_adds(ctx, opt):
    """context option gram_sparse = -1 (the default): from order gram_sparse_min_n the sparse form is taken where its multiply-adds,
    sum_i r_i^2 / 2 at the measured rate, undercut the dense product -- a 3 % dense matrix takes it, a 30 % dense one does not"""
    opt(gram_sparse=-1, gram_sparse_min_n=256, gram_bits=0)
    kw = dict(n_perturb=2, max_search_iters=5, streams=1, ctx=ctx)  # five: the smallest cap that leaves p_ < 1 (:756-760)
    Xs = api._csc_f32(synth_counts(6000, 500, seed=2, C=4, sparsity=0.97, min_genes_per_cell=5, marker_frac=0.2, marker_sd=1.5))
    Xd = api._csc_f32(synth_counts(6000, 500, seed=2, C=4, sparsity=0.70, marker_frac=0.2, marker_sd=1.5))
    assert Xs.nnz < 0.05 * 6000 * 500 and Xd.nnz > 0.25 * 6000 * 500
    rs = api.sclens(Xs, draws=api.make_draws_native(Xs, seed=5), **kw)
    rd = api.sclens(Xd, draws=api.make_draws_native(Xd, seed=5), **kw)
    assert rs["gram_sparse_used"] >= 1 and rd["gram_sparse_used"] == 0  # (counted from the binarised basis on; the union pattern of the
    # search carries twice the slots, four times the multiply-adds: there the estimate may go either way)
