"""CPU: the host control flow of `api.sclens` -- worker sessions on threads, speculative search rounds consumed in order, the
slot traffic of the ensemble, guard band, scoring and result assembly -- on a float64 NumPy stand-in for the device session
(tests/fake_session.py: the oracle's operations cut at the session entry points). On the same draws the result must be the
oracle's; sessions must never see overlapping calls; everything `sclens()` opens must be closed when it returns or raises.
The kernels are not involved (they have their own `-m gpu` parity tests); the C++ host statistics are the real ones."""
import numpy as np
import pytest

import fake_session as F
from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd.synth import synth_counts


def _pair(monkeypatch, N, M, streams, n_perturb=4, centering="mean", ensemble_tail="auto", **kw):
    X = synth_counts(N, M, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    ref = O.sclens(X, O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler), n_perturb=n_perturb, null_tol=O.NULL_DROP,
                   centering=centering, keep_intermediates=True, **kw)
    kw = dict(kw, ensemble_tail=ensemble_tail)
    main = F.install(monkeypatch)
    res = api.sclens(X, draws=d, n_perturb=n_perturb, ctx=main, streams=streams, centering=centering, keep_intermediates=True, **kw)
    return ref, res


def _same(ref, res):
    assert np.allclose(res["L"], ref["L"], rtol=1e-12, atol=1e-13)
    assert abs(res["lambda_c"] - ref["lambda_c"]) < 1e-9 * ref["lambda_c"]  # C++ host statistics against the oracle's
    assert len(res["signal_ev"]) == len(ref["signal_ev"]) > 0
    assert np.allclose(res["signal_ev"], ref["signal_ev"], rtol=1e-12)
    assert res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"] and res["pass"] == ref["pass"]
    for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.allclose(d1, d2, rtol=1e-10, atol=1e-13)
    ra, rb = res["robustness_scores"], ref["robustness_scores"]
    assert np.array_equal(ra["a_b"], rb["a_b"]) and np.allclose(ra["b_"], rb["b_"], atol=1e-10)
    assert np.allclose(ra["rob_score"], rb["rob_score"], atol=1e-10) and np.array_equal(res["sig_id"], ref["sig_id"])
    assert res["min_pc"] == ref["min_pc"]
    k = len(ref["signal_ev"])
    for t in range(len(ref["nL_set"])):
        got, want = np.asarray(res["nL_set"][t], dtype=np.float64), np.asarray(ref["nL_set"][t], dtype=np.float64)
        have = ~np.isnan(got)  # certified mode: the tail of a member that was not solved again is NaN, its values are kept as estimates
        assert have[:k].all() and np.allclose(got[have], want[have], rtol=1e-10)
        if not have.all():
            assert np.allclose(res["nL_tail_ritz_estimates"][t], want[k:], rtol=1e-10)
        assert np.allclose(np.abs(res["nV_set"][t]), np.abs(ref["nV_set"][t]), atol=1e-5)  # the fake hands out float32
    assert np.allclose(res["pca"], ref["pca"], atol=1e-5 * np.abs(ref["pca"]).max())
    assert np.allclose(res["gene_basis"], ref["gene_basis"], atol=1e-5 * np.abs(ref["gene_basis"]).max())


def _all_closed():
    assert F.FakeSession.live == 0 and F.FakePattern.live == 0
    assert F.FakeContext.live == 1  # the caller's own context


@pytest.mark.parametrize("streams", [1, 2, 3])
@pytest.mark.parametrize("N,M", [(150, 220), (260, 110)])
def test_sclens_host_flow_reproduces_the_oracle(monkeypatch, N, M, streams):
    ref, res = _pair(monkeypatch, N, M, streams)
    _same(ref, res)
    _all_closed()


def test_capped_search_median_centering_and_five_members(monkeypatch):
    ref, res = _pair(monkeypatch, 150, 220, 2, n_perturb=5, centering="median", max_search_iters=5)  # the smallest cap that leaves p_ < 1
    _same(ref, res)
    assert res["n_search"] == 5 and res["rec_vals"] == {}
    _all_closed()


def test_certified_ensemble_tail_redoes_the_members_the_session_names(monkeypatch):
    """ensemble_tail = "certified": every worker gets chefsi_tail_free = 1 before the ensemble; after the matching the members the
    session reports as uncertain (match_uncertain:t) are solved again on the main session with the tail converged (tail_free back to
    0, gap-aware target on), and the matching is redone; same result as the oracle, nothing left open"""
    log = []
    orig_set, orig_get = F.FakeSession.set_int, F.FakeSession.get_int
    monkeypatch.setattr(F.FakeSession, "set_int", lambda self, name, value: (log.append((id(self), name, value)), orig_set(self, name, value))[1])
    monkeypatch.setattr(F.FakeSession, "get_int", lambda self, name: 1 if name in ("match_uncertain:1", "chefsi_used") else orig_get(self, name))
    ref, res = _pair(monkeypatch, 150, 220, 2, ensemble_tail="certified")
    _same(ref, res)
    _all_closed()
    assert res["ensemble_tail"] == "certified" and res["tail_redo"] == [1]
    frees = [(sid, v) for sid, name, v in log if name == "chefsi_tail_free"]
    assert len({sid for sid, v in frees if v == 1}) == 2 and frees[-1][1] == 0  # both workers switched on; the main session off for the redo
    gaps = [v for _, name, v in log if name == "chefsi_tail_gap_milli"]
    assert gaps == [50, 0]


def test_a_failure_in_a_worker_closes_everything(monkeypatch):
    X = synth_counts(150, 220, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    main = F.install(monkeypatch)

    def boom(self):
        raise RuntimeError("binary basis failed")

    monkeypatch.setattr(F.FakeSession, "binary_basis", boom)
    with pytest.raises(RuntimeError, match="binary basis failed"):
        api.sclens(X, draws=d, n_perturb=3, ctx=main, streams=2)
    _all_closed()


def test_a_failed_search_evaluation_closes_everything(monkeypatch):
    X = synth_counts(150, 220, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    main = F.install(monkeypatch)
    seen = []
    orig = F.FakeSession.search_step

    def third_call_fails(self, sample, n_2):
        seen.append(1)
        if len(seen) == 3:
            raise RuntimeError("search evaluation failed")
        return orig(self, sample, n_2)

    monkeypatch.setattr(F.FakeSession, "search_step", third_call_fails)
    with pytest.raises(RuntimeError, match="search evaluation failed"):
        api.sclens(X, draws=d, n_perturb=3, ctx=main, streams=2)
    assert len(seen) <= 4  # the round that failed ends; no further round starts
    _all_closed()


# ---- several ranks (threads of this process): the spread first phase, rounds of world x streams evaluations, ensemble t % world
class _FakeThreadShard:
    """devutil.ThreadShard with the two device-buffer exchanges acting on the fakes' address space"""

    def __new__(cls, group, rank):
        from devutil import ThreadShard

        class S(ThreadShard):
            def bcast_dev(self, ctx, dev_ptr, count_f32, src):
                got = self._exchange(F.FakeContext.mem.get(dev_ptr) if self.rank == src else None)[src]
                if self.rank != src:
                    F.FakeContext.mem[dev_ptr] = got

            def allgather_dev(self, ctx, send_ptr, recv_ptr, count_f32):
                nbytes = 4 * int(count_f32)
                mine = {a - send_ptr: o for a, o in list(F.FakeContext.mem.items()) if send_ptr <= a < send_ptr + nbytes}
                for r, part in enumerate(self._exchange(mine)):
                    for off, o in part.items():
                        F.FakeContext.mem[recv_ptr + r * nbytes + off] = o

        return S(group, rank)


@pytest.mark.parametrize("world,streams", [(2, 1), (2, 2), (3, 2)])
def test_multi_rank_host_flow_reproduces_the_oracle(monkeypatch, world, streams):
    import threading

    from devutil import ThreadShard

    X = synth_counts(150, 220, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    ref = O.sclens(X, O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler), n_perturb=5, null_tol=O.NULL_DROP, keep_intermediates=True)
    F.install(monkeypatch).close()
    group = ThreadShard.Group(world)
    out, err = [None] * world, [None] * world

    def work(r):
        c = F.FakeContext(0)
        try:
            out[r] = api.sclens(X, draws=d, n_perturb=5, ctx=c, streams=streams, shard=_FakeThreadShard(group, r), keep_intermediates=(r == 0))
        except BaseException as e:  # noqa: BLE001 - reported below
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
    _same(ref, out[0])  # rank 0 scores: the whole result
    for r in range(1, world):  # the other ranks return the shared part
        assert out[r]["p_"] == ref["p_"] and out[r]["n_search"] == ref["n_search"] and "pca" not in out[r]
        assert np.allclose(out[r]["L"], ref["L"], rtol=1e-12, atol=1e-13)
    assert F.FakeSession.live == 0 and F.FakePattern.live == 0 and F.FakeContext.live == 0


@pytest.mark.parametrize("where", ["binary_basis", "search_step", "perturb"])
def test_a_rank_local_failure_stops_every_rank_instead_of_hanging_the_others(monkeypatch, where):
    """a rank-local error used to leave the other ranks waiting in the next broadcast / gather for ever. Now every phase
    that is followed by a collective ends with Shard.all_ok: the failing rank raises its own exception, the others a RuntimeError
    that names it -- nobody is released by the test (no barrier abort here), everything opened is closed."""
    import threading

    from devutil import ThreadShard

    X = synth_counts(150, 220, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    F.install(monkeypatch).close()
    world, bad_rank = 2, 1
    group = ThreadShard.Group(world)
    tid_of_rank = {}
    orig = getattr(F.FakeSession, where)

    def boom(self, *a, **k):
        if threading.get_ident() in tid_of_rank.get(bad_rank, ()):  # only the sessions driven by rank 1's threads
            raise RuntimeError(f"{where} failed on rank {bad_rank}")
        return orig(self, *a, **k)

    monkeypatch.setattr(F.FakeSession, where, boom)
    err = [None] * world

    def work(r):
        tid_of_rank.setdefault(r, set()).add(threading.get_ident())
        c = F.FakeContext(0)
        try:
            api.sclens(X, draws=d, n_perturb=4, ctx=c, streams=1, shard=_FakeThreadShard(group, r))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
        finally:
            c.close()

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for a peer that has failed"
    assert isinstance(err[bad_rank], RuntimeError) and f"{where} failed on rank {bad_rank}" in str(err[bad_rank])
    assert isinstance(err[0], RuntimeError) and "rank(s) [1] failed" in str(err[0])
    assert F.FakeSession.live == 0 and F.FakePattern.live == 0 and F.FakeContext.live == 0


# ---- the row-sharded (atlas) mode: cells divided over thread ranks ------------------------------------------------------------
@pytest.mark.parametrize("world,distribute", [(2, True), (3, True), (2, False)])
def test_row_sharded_host_flow_reproduces_the_oracle(monkeypatch, world, distribute):
    """sclens_amd/atlas.py on the stand-in: rounds of `world` evaluations / members with their roots, the stop rule on gathered
    statistics, the gathers of the cell-side results -- every rank must return the oracle's result on the whole matrix"""
    import threading

    from devutil import ThreadShard
    from sclens_amd import atlas

    N, M = 260, 110
    X = api._csc_f32(synth_counts(N, M, seed=3, C=4, marker_frac=0.25, marker_sd=1.5))
    d = api.make_draws_native(X, seed=19)                       # host candidates, device-side sample seeds
    dh = api.make_draws_native(X, seed=19, host_sampler=True)   # the same samples materialised for the oracle
    ref = O.sclens(X, O.Draws(dh.z_idx1, dh.z_idx2, dh.X_r, dh.p_th, dh.sampler), n_perturb=5, null_tol=O.NULL_DROP)
    F.install(monkeypatch).close()
    F.FakeShardedSession.registry.clear()
    monkeypatch.setattr(atlas, "Session", F.FakeShardedSession)
    group = ThreadShard.Group(world)
    out, err = [None] * world, [None] * world
    Xr = X.tocsr()

    def work(r):
        c = F.FakeContext(0)
        try:
            a, b = atlas.row_block(r, world, N)
            out[r] = atlas.sclens_row_sharded(Xr[a:b].tocsc(), a, N, d, ThreadShard(group, r), n_perturb=5, ctx=c, distribute=distribute)
        except BaseException as e:  # noqa: BLE001 - reported below
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
    for r in range(world):
        res = out[r]
        assert res["distributed"] == distribute and res["row_block"] == atlas.row_block(r, world, N)
        assert np.allclose(res["L"], ref["L"], rtol=1e-12, atol=1e-13) and len(res["signal_ev"]) == len(ref["signal_ev"]) > 0
        assert res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"]
        for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
            assert p1 == p2 and np.allclose(d1, d2, rtol=1e-10, atol=1e-13)
        assert np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"])
        assert np.allclose(res["robustness_scores"]["b_"], ref["robustness_scores"]["b_"], atol=1e-10)
        assert np.array_equal(res["sig_id"], ref["sig_id"])
        assert res["signal_evec"].shape == ref["signal_evec"].shape  # gathered over the ranks: all cells
        assert np.allclose(res["signal_evec"], ref["signal_evec"], atol=1e-6)
        assert np.allclose(res["pca"], ref["pca"], atol=1e-5 * np.abs(ref["pca"]).max())
        for key in ("TGC", "norm_tgc", "mat2_mean", "mat2_std", "cent_"):
            assert np.allclose(np.ravel(res["rec_vals"][key]), np.ravel(ref["rec_vals"][key]), rtol=1e-12)
        for t in range(5):
            assert np.allclose(res["nL_set"][t], ref["nL_set"][t], rtol=1e-10)
    assert F.FakeSession.live == 0 and F.FakeContext.live == 0


@pytest.mark.parametrize("streams", [1, 2, 3])
@pytest.mark.parametrize("where", ["data_spectrum", "null_spectrum_pattern", "signal_vectors", "binary_basis", "search_step", "perturb",
                                   "robustness", "gene_basis"])
def test_an_error_anywhere_propagates_and_leaves_nothing_open(monkeypatch, where, streams):
    """a session call that fails at any stage of sclens() -- on the main session or on a worker thread -- reaches the caller as
    the original exception, after every worker session, pattern and auxiliary context of the call has been closed"""
    X = synth_counts(150, 220, seed=3, C=4, marker_frac=0.25, marker_sd=1.5)
    d = api.make_draws(X, seed=11, p_th_trials=200)
    main = F.install(monkeypatch)
    real = getattr(F.FakeSession, where)
    n_calls = {"n": 0}

    def failing(self, *a, **kw):
        n_calls["n"] += 1
        if n_calls["n"] >= (2 if where in ("search_step", "perturb") else 1):  # the second search step / member: mid-loop
            raise RuntimeError(f"{where} failed")
        return real(self, *a, **kw)

    monkeypatch.setattr(F.FakeSession, where, failing)
    with pytest.raises(RuntimeError, match=f"{where} failed"):
        api.sclens(X, draws=d, n_perturb=4, ctx=main, streams=streams)
    _all_closed()


def test_randomised_sweep_of_the_host_flow(monkeypatch):
    """a dozen random shapes / seeds / stream counts / centrings: decisions and traces equal to the oracle's in every one"""
    rng = np.random.default_rng(2024)
    for case in range(12):
        N, M = int(rng.integers(90, 260)), int(rng.integers(90, 260))
        streams = int(rng.integers(1, 4))
        centering = "median" if case % 4 == 3 else "mean"
        X = synth_counts(N, M, seed=100 + case, C=int(rng.integers(2, 6)), marker_frac=0.25, marker_sd=1.5)
        d = api.make_draws(X, seed=int(rng.integers(1, 10**6)), p_th_trials=100)
        ref = O.sclens(X, O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler), n_perturb=3, null_tol=O.NULL_DROP,
                       centering=centering, max_search_iters=8)
        main = F.install(monkeypatch)
        res = api.sclens(X, draws=d, n_perturb=3, ctx=main, streams=streams, centering=centering, max_search_iters=8)
        tag = (case, N, M, streams, centering)
        assert len(res.get("signal_ev", [])) == len(ref["signal_ev"]), tag
        assert res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"], tag
        for (p1, d1), (p2, d2) in zip(res["search_trace"], ref["search_trace"]):
            assert p1 == p2 and np.allclose(d1, d2, rtol=1e-9, atol=1e-12), tag
        if len(ref["signal_ev"]):
            assert np.array_equal(res["sig_id"], ref["sig_id"]), tag
            assert np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"]), tag
        _all_closed()
        main.close()
