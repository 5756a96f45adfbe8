"""-m gpu: the device QC filter (sclens_hip_preprocess_csc / _gather; scLENS.jl:160-236) against the oracle's
restatement: integer / index work, so the comparison is exact."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import sclens_oracle as O
from sclens_amd import api

pytestmark = pytest.mark.gpu


def _counts(N, M, seed, rate=0.3):
    rng = np.random.default_rng(seed)
    lib = rng.lognormal(0.0, 0.6, size=(N, 1))
    gene = rng.lognormal(np.log(rate), 1.0, size=(1, M))
    X = rng.poisson(lib * gene).astype(np.float32)
    names = np.array([f"g{j}" for j in range(M)], dtype=object)
    mt = rng.choice(M, size=max(2, M // 30), replace=False)
    for t, j in enumerate(mt[: len(mt) // 2]):
        names[j] = ("MT-" if t % 2 else "mt-") + f"X{t}"
    for t, j in enumerate(mt[len(mt) // 2:]):
        names[j] = ("RPS" if t % 2 else "rpl") + f"{t}"
    return X, names


CASES = [
    dict(),  # reference defaults
    dict(min_genes_per_cell=40, min_cells_per_gene=30, mito_percent=3.0, ribo_percent=4.0),
    dict(min_genes_per_cell=10, min_cells_per_gene=5, mito_percent=0.0, ribo_percent=0.0, max_genes_per_cell=120),
    dict(min_tp_c=50, max_tp_c=400, min_tp_g=20, max_tp_g=2000, min_genes_per_cell=20, min_cells_per_gene=10),
]


@pytest.mark.parametrize("N,M,seed", [(700, 900, 1), (1500, 400, 2), (257, 129, 3)])
@pytest.mark.parametrize("kw", CASES)
def test_preprocess_matches_oracle_exactly(ctx, N, M, seed, kw):
    X, names = _counts(N, M, seed)
    want = O.preprocess_counts(X, names, **kw)
    got = api.preprocess(sp.csc_matrix(X), names, ctx=ctx, **kw)
    if want is None:
        assert got is None
        return
    Xw, gw, cw = want
    Xg, gg, cg = got
    assert Xg.shape == Xw.shape and Xg.dtype == np.float32
    assert np.array_equal(cg, cw)
    assert list(gg) == list(gw)  # same genes in the same (mean-sorted, tie-stable) order
    assert np.array_equal(Xg.toarray(), Xw)
    assert Xg.has_sorted_indices or np.all(np.diff(Xg.indices[Xg.indptr[0]:Xg.indptr[1]]) > 0)
    assert np.all(Xg.data != 0)


def test_preprocess_then_sclens_runs_on_the_filtered_matrix(ctx):
    """example.jl:17-24 call order: preprocess -> sclens (file ingest is out of scope)."""
    X, names = _counts(500, 700, 5, rate=0.5)
    out = api.preprocess(X, names, cell_names=[f"c{i}" for i in range(500)], min_genes_per_cell=50, min_cells_per_gene=20,
                         ctx=ctx)
    Xf, genes, cells = out
    assert str(cells[0]).startswith("c")
    assert np.all(np.diff(Xf.indptr) >= 20) and np.all(np.asarray((Xf != 0).sum(axis=1)).ravel() > 0)
    res = api.sclens(Xf, seed=3, n_perturb=3, ctx=ctx, max_search_iters=6)
    assert res["L"].shape == (min(Xf.shape),)


def test_preprocess_edge_cases(ctx):
    X, names = _counts(120, 90, 7)
    assert api.preprocess(X, names, min_genes_per_cell=10 ** 6, ctx=ctx) is None  # nothing passes (scLENS.jl:232-234)
    # explicit zeros in the stored pattern are not counts
    Xs = sp.csc_matrix(X)
    Xs.data[::7] = 0.0
    a = api.preprocess(Xs, names, min_genes_per_cell=5, min_cells_per_gene=5, ctx=ctx)
    b = O.preprocess_counts(Xs.toarray(), names, min_genes_per_cell=5, min_cells_per_gene=5)
    assert np.array_equal(a[0].toarray(), b[0]) and list(a[1]) == list(b[1])
    # gather without a preceding stats call: state error, not a crash
    lib = ctx.lib
    buf = np.zeros(4, dtype=np.int64)
    rc = lib.sclens_hip_preprocess_gather(ctx.h, api.ptr(buf, C.c_int64), api.ptr(buf.astype(np.int32), C.c_int32),
                                          api.ptr(buf.astype(np.float32), C.c_float))
    assert rc == 7
    # malformed CSC: error code
    cp = np.array([0, 3, 2], dtype=np.int64)
    rc = lib.sclens_hip_preprocess_csc(ctx.h, 4, 2, api.ptr(cp, C.c_int64), api.ptr(np.zeros(3, np.int32), C.c_int32),
                                       api.ptr(np.ones(3, np.float32), C.c_float), None, None, 0.0, 0.0, 1e300, 1e300, 1, 0, 1,
                                       5.0, 0.0, api.ptr(np.zeros(4, np.uint8), C.c_uint8), api.ptr(np.zeros(2, np.int64), C.c_int64),
                                       C.byref(C.c_int64()), C.byref(C.c_int64()), C.byref(C.c_int64()))
    assert rc == 1


def test_device_handover_preprocess_to_sclens(ctx):
    """SURVEY 8f-3 remainder: the filtered matrix stays in HBM (sclens_hip_preprocess_keep) and the session + the union pattern
    are built from it in place. Same filtered matrix bit for bit as the host round trip, and sclens() on the handle gives the
    same result as sclens() on the downloaded matrix (same seed: identical draws, identical kernels)."""
    from sclens_amd.synth import synth_counts

    X = synth_counts(700, 900, seed=3, C=5, marker_frac=0.2, marker_sd=1.5).toarray()  # clustered: a few signals survive
    names = np.array([("mt-x%d" % j) if j % 97 == 0 else ("g%d" % j) for j in range(X.shape[1])], dtype=object)
    kw = dict(min_genes_per_cell=50, min_cells_per_gene=20)
    Xh, gh, ch = api.preprocess(X, names, ctx=ctx, **kw)
    out = api.preprocess(X, names, ctx=ctx, keep_on_device=True, **kw)
    Xd, gd, cd = out
    try:
        assert isinstance(Xd, api.DeviceCounts) and Xd.shape == Xh.shape and Xd.nnz == Xh.nnz
        assert list(gd) == list(gh) and np.array_equal(cd, ch)
        cp, rv, nz = Xd.download()
        assert np.array_equal(cp, Xh.indptr) and np.array_equal(rv, Xh.indices) and np.array_equal(nz, Xh.data)
        a = api.sclens(Xd, seed=5, n_perturb=3, max_search_iters=6, ctx=ctx, streams=2)
        d = api.make_draws_native(Xh, 5, device_candidates=True)
        b = api.sclens(Xh, draws=d, n_perturb=3, max_search_iters=6, ctx=ctx, streams=2)
        assert np.array_equal(a["L"], b["L"]) and a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
        for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
            assert p1 == p2 and np.array_equal(t1, t2)
        assert len(a["signal_ev"]) == len(b["signal_ev"]) >= 2
        assert np.array_equal(a["sig_id"], b["sig_id"])
        assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])
        assert np.array_equal(a["gene_basis"], b["gene_basis"])
    finally:
        Xd.close()
    # the same handle from host arrays, and the error paths of the hand-over
    up = api.DeviceCounts.upload(ctx, Xh)
    try:
        assert up.shape == Xh.shape and np.array_equal(up.to_scipy().toarray(), Xh.toarray())
        cp_only = up.download(rowval=False, nzval=False)
        assert cp_only[1] is None and np.array_equal(cp_only[0], Xh.indptr)
    finally:
        up.close()
    h = C.c_void_p()
    assert ctx.lib.sclens_hip_preprocess_keep(ctx.h, C.byref(h)) == 7  # no preceding preprocess call: state error
    assert ctx.lib.sclens_hip_session_create_from_counts(ctx.h, None, C.byref(h)) == 1
