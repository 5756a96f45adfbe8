"""-m gpu: the sparse pattern built on the device (pattern_dev.hip) against the host builder and the
host R1 generator: the device arrays are bit-identical, the device-drawn candidate list is the list of
sclens_draw_zero_candidates for the same seed, and sclens() decides the same either way."""
import ctypes as C

import numpy as np
import pytest

from sclens_amd import _lib, api
from sclens_amd.synth import synth_counts

pytestmark = pytest.mark.gpu

COUNTS = lambda p, X: [X.shape[1] + 1, p_nU(p, X), X.shape[0] + 1, p_nU(p, X), p_nU(p, X), p.ncand, p_nU(p, X)]


def p_nU(p, X):
    return int(X.nnz + p.ncand)


def _arrays(p, X):
    return [p.download(w, c) for w, c in enumerate(COUNTS(p, X))]


def _host_candidates(X, seed):
    lib = _lib.load()
    cp = np.ascontiguousarray(X.indptr, dtype=np.int64)
    rv = np.ascontiguousarray(X.indices, dtype=np.int32)
    z1, z2 = np.empty(X.nnz, dtype=np.uint32), np.empty(X.nnz, dtype=np.uint32)
    cnt = C.c_int64(0)
    assert lib.sclens_draw_zero_candidates(X.shape[0], X.shape[1], api.ptr(cp, C.c_int64), api.ptr(rv, C.c_int32), seed,
                                           api.ptr(z1, C.c_uint32), api.ptr(z2, C.c_uint32), C.byref(cnt)) == 0
    return z1[: cnt.value].copy(), z2[: cnt.value].copy()


@pytest.mark.parametrize("N,M", [(300, 500), (900, 260), (64, 70)])
def test_device_pattern_equals_host_pattern(ctx, N, M, opt):
    X = api._csc_f32(synth_counts(N, M, seed=N + M, C=4, marker_frac=0.2))
    z1, z2 = _host_candidates(X, 77)
    assert len(z1) > 0 and len(set(zip(z1.tolist(), z2.tolist()))) == len(z1)  # distinct pairs, none a stored entry
    dense = X.toarray()
    assert not np.any(dense[z1, z2] != 0)
    opt(host_pattern=1)
    ph = api.Pattern(ctx, X, z1, z2)
    opt(host_pattern=0)
    pd_ = api.Pattern(ctx, X, z1, z2)       # device build from the same host list
    pr = api.Pattern.drawn(ctx, X, 77)       # device build with the list drawn on the device
    try:
        assert pr.ncand == len(z1)
        d1, d2 = pr.candidates()
        assert np.array_equal(d1, z1) and np.array_equal(d2, z2)
        ah, ad, ar = _arrays(ph, X), _arrays(pd_, X), _arrays(pr, X)
        for w, (a, b, c) in enumerate(zip(ah, ad, ar)):
            assert np.array_equal(a, b), f"array {w}: device build differs from the host build"
            assert np.array_equal(a, c), f"array {w}: device-drawn build differs from the host build"
        # counts-only pattern (what session_create and the null matrix use)
        opt(host_pattern=1)
        p0h = api.Pattern(ctx, X, np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        opt(host_pattern=0)
        p0d = api.Pattern(ctx, X, np.zeros(0, np.uint32), np.zeros(0, np.uint32))
        for a, b in zip(_arrays(p0h, X), _arrays(p0d, X)):
            assert np.array_equal(a, b)
        p0h.close()
        p0d.close()
    finally:
        for p in (ph, pd_, pr):
            p.close()


def test_sclens_with_device_drawn_candidates(ctx):
    """the whole path with R1 drawn on the device equals the run that gets the same list from the host generator"""
    X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    a = api.sclens(X, draws=api.make_draws_native(X, seed=21), n_perturb=4, ctx=ctx, max_search_iters=6)
    b = api.sclens(X, draws=api.make_draws_native(X, seed=21, device_candidates=True), n_perturb=4, ctx=ctx, max_search_iters=6)
    assert a["p_"] == b["p_"] and a["n_search"] == b["n_search"]
    for (p1, t1), (p2, t2) in zip(a["search_trace"], b["search_trace"]):
        assert p1 == p2 and np.array_equal(t1, t2)
    assert np.array_equal(a["L"], b["L"]) and np.array_equal(a["sig_id"], b["sig_id"])
    assert np.array_equal(a["robustness_scores"]["b_"], b["robustness_scores"]["b_"])


def test_null_matrix_in_page_locked_memory(ctx, monkeypatch):
    """(opt-in, SCLENS_PINNED_DRAWS=1) the null matrix drawn into page-locked host blocks (sclens_hip_host_alloc, above 4M stored entries) is the matrix drawn into
    ordinary memory, its pattern on the device is the same, and a block is handed out again once the matrix built on it is gone"""
    import gc

    def block(arr):  # the page-locked block under an array (views of views of the ctypes buffer the pool wrapped)
        while isinstance(arr, np.ndarray):
            arr = arr.base
        return arr._sclens_block

    X = api._csc_f32(synth_counts(30000, 1500, seed=9, C=3))
    assert X.nnz >= (1 << 22), X.nnz
    monkeypatch.setenv("SCLENS_PINNED_DRAWS", "0")
    plain = api.make_draws_native(X, seed=77).X_r
    monkeypatch.setenv("SCLENS_PINNED_DRAWS", "1")
    api._pinned.trim()
    pinned = api.make_draws_native(X, seed=77).X_r
    assert api._pinned.idle == 0 and block(pinned.indices).nbytes >= 4 * X.nnz  # really on page-locked blocks, both in use
    assert np.array_equal(pinned.indptr, plain.indptr) and np.array_equal(pinned.indices, plain.indices)
    assert np.array_equal(pinned.data, plain.data)
    pa, pb = api.Pattern(ctx, plain, [], []), api.Pattern(ctx, pinned, [], [])
    try:
        for w, c in enumerate(COUNTS(pa, plain)):
            assert np.array_equal(pa.download(w, c), pb.download(w, c)), w
    finally:
        pa.close()
        pb.close()
    addrs = {block(pinned.indices).addr, block(pinned.data).addr}
    del pinned, pb
    gc.collect()
    assert api._pinned.idle > 0  # both blocks are back on the free list ...
    again = api.make_draws_native(X, seed=78).X_r
    assert {block(again.indices).addr, block(again.data).addr} == addrs  # ... and are handed out again
    del again
    gc.collect()
    api._pinned.trim()
    assert api._pinned.idle == 0
