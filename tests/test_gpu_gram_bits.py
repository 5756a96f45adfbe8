"""-m gpu: the Gram matrix of a binarised, scaled count matrix formed on the fp16 MFMA as a weighted co-occurrence product
(csrc/gram_bits.hip, the sparsity search's path for large N > M problems) against the float64 oracle and against the general
path (scaled matrix + fp32 product), kernel level and end to end.

Tolerance: both device paths accumulate in fp32; entries are compared relative to the largest entry of the Gram matrix
(its diagonal), 2e-5, and the fp16 path may not be worse than twice the general path's own error (+ 1e-6)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd.synth import synth_counts

pytestmark = pytest.mark.gpu


def _binarised(N, M, seed):
    X = synth_counts(N, M, seed=seed, C=5, marker_frac=0.2, marker_sd=1.5).tocsc()
    X.data[:] = 1.0
    return X


@pytest.mark.parametrize("N,M,terms", [(700, 300, 2), (1500, 520, 2), (2113, 1030, 2), (1500, 520, 3)])
def test_gram_bits_matches_oracle_and_fp32_path(ctx, N, M, terms):
    """sizes: M not a multiple of the 256-gene tile, N not a multiple of the 64-cell stage (zero-padded cells), more than one
    tile row; 2 and 3 fp16 pieces of the cell weights."""
    Xb = _binarised(N, M, seed=N + M)
    S = np.asarray(O.logn_scale(O.pre_scale(Xb)), dtype=np.float64)  # N x M, the closure path of scLENS.jl:650-652
    want = S.T @ S / N
    with ctx.options(gram_bits_terms=terms):
        got = api._gram_binary(Xb, use_bits=True, ctx=ctx)
        with ctx.options(split_pipe=0):  # the two-buffer stage loop of round 4: the same products in the same order
            assert np.array_equal(got, api._gram_binary(Xb, use_bits=True, ctx=ctx))
    dense = api._gram_binary(Xb, use_bits=False, ctx=ctx)
    scale = np.abs(want).max()
    e_bits, e_dense = np.abs(got - want).max() / scale, np.abs(dense - want).max() / scale
    assert np.array_equal(got, got.T)  # exactly symmetric (mirrored stores)
    assert e_bits < 2e-5, (e_bits, e_dense)
    assert e_bits < 2 * e_dense + 1e-6, (e_bits, e_dense)
    # spectrum: what the search consumes
    wl, gl = np.linalg.eigvalsh(want), np.linalg.eigvalsh(got.astype(np.float64))
    assert np.abs(wl - gl).max() < 2e-5 * wl.max()


def test_gram_bits_other_divisor_and_dense_cells(ctx):
    """a divisor other than N and cells that express most genes (large weights spread: TGC from ~20 to ~M)."""
    N, M = 900, 400
    rng = np.random.default_rng(3)
    dens = np.concatenate([np.full(N // 3, 0.05), np.full(N // 3, 0.4), np.full(N - 2 * (N // 3), 0.95)])
    P = (rng.random((N, M)) < dens[:, None]).astype(np.float32)
    P[np.arange(N), rng.integers(0, M, size=N)] = 1.0  # every cell expresses something
    P[rng.integers(0, N, size=M), np.arange(M)] = 1.0  # every gene is seen
    Xb = sp.csc_matrix(P)
    S = np.asarray(O.logn_scale(O.pre_scale(Xb)), dtype=np.float64)
    want = S.T @ S / 123.0
    got = api._gram_binary(Xb, use_bits=True, divisor=123.0, ctx=ctx)
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()


def test_sclens_with_gram_bits_matches_oracle(ctx):
    """End to end, N > M, the fp16 Gram path forced on (context option "gram_bits"): same decisions as the float64 oracle
    on the same draws -- search length, p_, signal count -- and the search statistics within the fp32 tolerance."""
    N, M = 600, 250
    X = synth_counts(N, M, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws(X, seed=7, p_th_trials=300)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=6, keep_intermediates=True, null_tol=O.NULL_DROP)
    ctx.set_option("gram_bits", 1)
    try:
        res = api.sclens(X, draws=d, n_perturb=6, ctx=ctx, keep_intermediates=True, streams=1)
    finally:
        ctx.set_option("gram_bits", -1)
    assert res["gram_bits_used"] >= res["n_search"] + 1  # binary basis + every search step took the fp16 path
    assert res["n_search"] == ref["n_search"]
    assert res["p_"] == ref["p_"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2
        assert np.abs(d1 - d2).max() < 3e-3, (d1, d2)
    assert len(res["signal_ev"]) == len(ref["signal_ev"]) > 0
    assert list(res["sig_id"]) == list(ref["sig_id"])


def test_pattern_image_written_once_gives_the_same_bits(ctx, opt):
    """k_mask_fused (round 5: the 0/1 image of a binarised matrix written once from LDS chunks of 8 192 cells) against memset +
    scatter (context option dense_fused = 0): N spans three chunks with a partial last one, the union pattern has the unordered
    candidate tail; the search statistics of every evaluation and the binarised basis' eigenvalue count are bitwise the same."""
    N, M = 20000, 300
    X = synth_counts(N, M, seed=4, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=23)
    out = []
    for flag in (1, 0):
        opt(gram_bits=1, dense_fused=flag)
        out.append(api.sclens(X, draws=d, n_perturb=2, ctx=ctx, max_search_iters=4, streams=1))
    a, b = out
    assert a["gram_bits_used"] >= a["n_search"] + 1 and a["gram_bits_used"] == b["gram_bits_used"]
    assert a["n_search"] == b["n_search"] and a["p_"] == b["p_"]
    for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
        assert p1 == p2 and np.array_equal(t1, t2)


@pytest.mark.parametrize("pipe", [1, 0])
@pytest.mark.parametrize("n,p,q", [(500, 300, 130), (1000, 777, 260), (2050, 520, 515), (96, 40, 33)])
def test_corr_colmax_split_fp16(ctx, n, p, q, pipe, opt):
    """max_i |X_i' Y_j| from split fp16 images against float64 and against the fp32 product: unit columns with entries over
    several orders of magnitude (localised + delocalised vectors), n not a multiple of the 32-deep stage."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, p)) * np.exp(rng.normal(0, 2.0, size=(n, 1)))
    Y = rng.standard_normal((n, q)) * np.exp(rng.normal(0, 2.0, size=(n, 1)))
    Y[:, 0] = X[:, 3]  # one perfectly correlated pair
    X /= np.linalg.norm(X, axis=0)
    Y /= np.linalg.norm(Y, axis=0)
    want = np.abs(X.astype(np.float32).astype(np.float64).T @ Y.astype(np.float32).astype(np.float64)).max(axis=0)
    opt(split_pipe=pipe)  # 1: the stage loop as a software pipeline (round 5), 0: the two-buffer loop; the same products in the same order
    got = api._corr_colmax(X, Y, use_split=True, ctx=ctx)
    if pipe:
        opt(split_pipe=0)
        assert np.array_equal(got, api._corr_colmax(X, Y, use_split=True, ctx=ctx))  # the same bits
    ref32 = api._corr_colmax(X, Y, use_split=False, ctx=ctx)
    assert abs(got[0] - 1.0) < 2e-6
    assert np.abs(got - want).max() < 2e-6, np.abs(got - want).max()
    assert np.abs(got - want).max() < 2 * np.abs(ref32 - want).max() + 1e-6
