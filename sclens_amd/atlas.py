"""Row-sharded sclens() for the atlas configuration: cells > genes, every rank holds a contiguous block of
cells. The Gram matrix X'X (genes x genes) is a sum over cells, so each decomposition is: local normalisation statistics
(three O(M) all-reduces + one scalar), local partial Gram matrix, one sum of 4 M^2 bytes over the ranks, the eigen-solver, and
the recovery GEMM X_g V, which is row-local. All ranks run the same control flow (the serial loop of scLENS.jl:715-778); the
library calls the reducers of `Shard` (RCCL inside the library when there is a communicator) wherever the path has a real
exchange.

Two ways to place the eigen-solver: replicated on identical inputs (all-reduce of the Gram matrix; `distribute=False`), or
-- the default for several ranks with device-side samples -- a ROUND of `world` independent decompositions (consecutive
sparsities of the search, consecutive ensemble members) whose partial Gram matrices are each reduced onto ONE rank, which solves
it while the others solve theirs (`search_round_seeded`, `perturb_round_seeded`). The first three decompositions (data, null,
binarised) run on one rank each as well (session option `solve_root`; their results are shared inside the library). Memory scales with the ranks either way: the sparse pattern, the candidates (drawn locally:
no rank holds foreign ones), the dense scaled matrix and the cell-side vectors are divided by their number. For cells <= genes,
or when the matrix fits one GPU, use api.sclens(shard=...), which distributes whole decompositions instead.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp

from . import _lib
from .api import (Context, Draws, Session, _csc_f32, _mp_calculation, _resolve, _robust_scores, _tw, cut_with_guard_band,
                  default_context, mp_check, sample_seed_for)
from .shard import Shard, consume_search_round, search_schedule


def row_block(rank: int, world: int, N: int):
    """cells [row0, row1) of rank `rank`: contiguous, sizes differing by at most one"""
    base, rem = divmod(N, world)
    row0 = rank * base + min(rank, rem)
    return row0, row0 + base + (1 if rank < rem else 0)


def _gather_rows(shard: Shard, local: np.ndarray, N_global: int) -> np.ndarray:
    """stack the ranks' row blocks (rows = cells) into the full array on every rank"""
    if shard.world == 1:
        return local
    per = (N_global + shard.world - 1) // shard.world
    pad = np.zeros((per,) + local.shape[1:], dtype=np.float64)
    pad[: local.shape[0]] = local
    allp = shard.allgather_small(pad)
    parts = []
    for r in range(shard.world):
        a, b = row_block(r, shard.world, N_global)
        parts.append(allp[r, : b - a])
    return np.concatenate(parts, axis=0).astype(local.dtype)


def sclens_row_sharded(X_local, row0: int, N_global: int, draws: Draws, shard: Shard, th=60, p_step=0.001, n_perturb=20,
                       ctx: Optional[Context] = None, max_search_iters: Optional[int] = None, gather: bool = True,
                       verbose: bool = False, guard_band: float = 4.0, distribute: Optional[bool] = None,
                       nnz_global: Optional[int] = None, return_candidates: bool = False) -> Dict[str, object]:
    """scLENS.sclens (scLENS.jl:649-832) with the cells divided over the ranks of `shard`.

    X_local: this rank's cells (N_local x M, N_global > M). draws: the GLOBAL draws, identical on every rank (X_r = the whole
    null matrix, of which this rank takes its rows, or already its rows; sampler / sample_seed index the global candidate list).
    Candidates, two modes:
      * `draws.z_idx1 / z_idx2` given (global cell indices of the whole list): every rank holds the list, slot -1 for foreign cells
        (the mode the parity tests replay against the unsharded path);
      * `draws.z_idx1 is None` and `draws.cand_seed` set: every rank draws ITS part of the global draw sequence on the device
        (`nnz_global` = stored entries of the whole matrix; default: the sum over the ranks), the global list is the concatenation
        of the ranks' lists -- no rank holds foreign candidates (what the atlas configuration needs: 21 GB -> 2.6 GB per rank).
    `distribute` (default: world > 1 and device-side samples): the evaluations of a search round (`world` sparsities at a time)
    and the ensemble members are decomposed by DIFFERENT ranks -- each partial Gram matrix is summed onto the rank that solves it --
    instead of replicated on all. Returns the reference's result keys; with `gather` the cell-side arrays cover all cells."""
    ctx = ctx or default_context()
    t_all = time.perf_counter()
    X_local = _csc_f32(X_local)
    N_local, M = X_local.shape
    if N_global <= M:
        raise ValueError("row sharding needs cells > genes; use api.sclens(shard=...) otherwise")
    Xr = _resolve(draws.X_r)
    Xr_local = _csc_f32(sp.csc_matrix(Xr.tocsr()[row0: row0 + N_local])) if Xr.shape[0] == N_global else _csc_f32(Xr)
    local_cands = draws.z_idx1 is None and draws.cand_seed is not None
    if distribute is None:
        distribute = shard.world > 1 and draws.sampler is None
    if distribute and draws.sampler is not None:
        raise ValueError("distribute=True needs device-side samples (draws.sampler is None)")
    if local_cands:
        if nnz_global is None:
            nnz_global = int(shard.allgather_small(np.array([float(X_local.nnz)])).sum())
        ses = Session.create_sharded_drawn(ctx, X_local, row0, N_global, nnz_global, draws.cand_seed, shard.reducer(ctx))
        counts = shard.allgather_small(np.array([float(ses.ncand_local)]))[:, 0].astype(np.int64)
        ses.set_candidate_range(int(counts[: shard.rank].sum()), int(counts.sum()))
        n_cand = int(counts.sum())
    else:
        z1, z2 = _resolve(draws.z_idx1), _resolve(draws.z_idx2)
        ses = Session.create_sharded(ctx, X_local, row0, N_global, z1, z2, shard.reducer(ctx))
        n_cand = len(z1)
    if distribute:
        ses.set_reduce_to(shard.reducer_to(ctx))
        ses.set_int("shard_rank", shard.rank)
    # the three first decompositions on ONE rank each (data 0, null 1, binarised 2, modulo the ranks) instead of replicated on all:
    # the partial Gram matrix is summed onto that rank only, it alone runs the eigensolver, and the eigenvalues + the few vectors the
    # others recover their cells from (+ Vr2) are shared inside the library (session option solve_root; round 6)
    first_root = (lambda q: q % shard.world) if distribute else (lambda q: -1)
    try:
        ses.set_int("solve_root", first_root(1)) if distribute else None
        Lr = shard.agree(ses.null_spectrum(Xr_local))  # :704
        ses.set_int("solve_root", first_root(0)) if distribute else None
        L, rec_vals = ses.data_spectrum()
        L = shard.agree(L)
        L_mp, _, _ = _mp_calculation(L, Lr[:-1])
        lambda_c = _tw(L, L_mp)[0]
        # the same guard band as api.sclens: every rank refines (the quotient is summed over the row blocks inside the library)
        L, k, nL, guard = cut_with_guard_band(L, lambda_c, guard_band, lambda lo, hi: shard.agree(ses.refine_eigenvalues(lo, hi)))
        if verbose and shard.rank == 0:
            print(f"(Using hip, {shard.world} row blocks) number of signal ev: {k}")
        nV_local = ses.signal_vectors(k)
        ses.set_int("solve_root", first_root(2)) if distribute else None
        _, r_vr2 = ses.binary_basis()  # :717-721
        ses.set_int("solve_root", -1) if distribute else None
        mpC = mp_check(L_mp)
        p_th = draws.p_th
        n_2 = int(round(r_vr2 / 2))
        p_list = search_schedule(p_step)
        tank = np.zeros((5, 0))
        it, p_ = 0, None
        while p_ is None:  # :725-761
            if distribute:
                # a round of `world` consecutive sparsities: rank r decomposes evaluation it + r (its Gram matrix is summed onto
                # rank r only); the statistics are gathered and consumed in order with the reference's stop rule
                W = shard.world
                ms = [int(round((1 - p_list[it + e]) * M * N_global)) for e in range(W)]
                ok = [n_cand >= mm for mm in ms]
                live = [e for e in range(W) if ok[e]]
                mine = np.full(6, np.nan)
                if live:
                    seeds = [sample_seed_for(draws.sample_seed, "search", it + e) for e in live]
                    my_slot = live.index(shard.rank) if shard.rank in live else -1
                    # the statistic of an evaluation is computed by its root alone, after the reduces: a failure there is local to
                    # that rank, and the gather below is a collective -- agree on the outcome first (Shard.all_ok)
                    try:
                        d5, _ = ses.search_round_seeded(seeds, [ms[e] for e in live], live, my_slot, n_2)
                    except BaseException as e:
                        shard.all_ok(e, "a search round")
                        raise
                    shard.all_ok(None, "a search round")
                    if my_slot >= 0:
                        mine[:5], mine[5] = d5, 1.0
                allr = shard.allgather_small(mine)
                results = [allr[e, :5] if allr[e, 5] == 1.0 else None for e in range(W)]
            else:  # one evaluation at a time, all ranks together
                nnzidx = int(round((1 - p_list[it]) * M * N_global))
                d5 = None
                if n_cand >= nnzidx:
                    if draws.sampler is not None:
                        d5, _ = ses.search_step(draws.sampler("search", it, n_cand, nnzidx), n_2)
                    else:
                        d5, _ = ses.search_step_seeded(sample_seed_for(draws.sample_seed, "search", it), nnzidx, n_2)
                    d5 = shard.agree(d5)
                results = [d5]
            tank, used, stopped, p_fin = consume_search_round(tank, results, p_list, it, p_th, p_step, max_search_iters)
            it += used
            if stopped:
                p_ = p_fin
        trace = [(p_list[q], tank[:, q].copy()) for q in range(tank.shape[1])]
        min_pc = int(math.ceil(k * 1.5))
        m_pert = int(round((1 - p_) * M * N_global))
        nL_set, ncols = [None] * n_perturb, [0] * n_perturb
        res: Dict[str, object] = {"L": L, "L_mp": L_mp, "λ": lambda_c, "lambda_c": lambda_c, "p_": p_, "p_th": p_th,
                                  "n_search": it, "search_trace": trace, "row_block": (row0, row0 + N_local),
                                  "guard_band": guard, "n_cand": n_cand, "distributed": bool(distribute),
                                  "local_candidates": bool(local_cands)}
        if return_candidates and local_cands:  # this rank's part of the global list (global cell indices): tests replay with it
            res["candidates_local"] = ses.local_candidates()
        if k == 0:  # :780-784
            res["partial_eig"] = (0, 0)
            res["wall_s"] = time.perf_counter() - t_all
            return res
        if distribute:  # :767-778, `world` members per round, member t decomposed by rank t mod world
            W = shard.world
            for t0 in range(0, n_perturb, W):
                ts = list(range(t0, min(n_perturb, t0 + W)))
                roots = [t % W for t in ts]
                my_slot = roots.index(shard.rank) if shard.rank in roots else -1
                nl, nc = ses.perturb_round_seeded(ts, [sample_seed_for(draws.sample_seed, "perturb", t) for t in ts],
                                                  [m_pert] * len(ts), roots, my_slot, min_pc)
                for e, t in enumerate(ts):
                    nL_set[t], ncols[t] = nl[e], nc[e]
        else:
            for t in range(n_perturb):  # :767-778
                if draws.sampler is not None:
                    nL_set[t], ncols[t] = ses.perturb(t, draws.sampler("perturb", t, n_cand, m_pert), min_pc)
                else:
                    nL_set[t], ncols[t] = ses.perturb_seeded(t, sample_seed_for(draws.sample_seed, "perturb", t), m_pert, min_pc)
        a_b, b_ = ses.robustness(k, n_perturb)  # :786-807; the small products are summed over the ranks inside
        b_ = shard.agree(b_)
        m_score, sd_score = _robust_scores(b_)
        sig_id = np.flatnonzero(m_score > math.cos(math.radians(th)))
        gmat = ses.gene_basis(nL)
        nV = _gather_rows(shard, nV_local, N_global) if gather else nV_local
        if gather:
            for key in ("TGC", "norm_tgc"):
                rec_vals[key] = _gather_rows(shard, np.ravel(rec_vals[key])[:, None], N_global)[:, 0]
        res.update({"pca": nV * np.sqrt(nL)[None, :].astype(np.float32),
                    "pca_n1": nV[:, sig_id] * np.sqrt(nL[sig_id])[None, :].astype(np.float32), "sig_id": sig_id,
                    "robustness_scores": {"b_": b_, "rob_score": m_score, "m_scores": m_score, "sd_scores": sd_score,
                                          "a_b": a_b},
                    "signal_evec": nV, "signal_ev": nL, "gene_basis": gmat, "pass": mpC["pass"],
                    "ks_static": mpC["ks_static"], "rec_vals": rec_vals, "nL_set": nL_set, "min_pc": min_pc,
                    "partial_eig": (ses.get_int("chefsi_used"), ses.get_int("chefsi_fallback"))})
        res["wall_s"] = time.perf_counter() - t_all
        return res
    finally:
        ses.close()


# ------------------------------------------------------------------------------------------------------------------------------
# The atlas matrices themselves (BASELINE.json configs[4], 1 000 000 x 30 000): a matrix of that size is never held in one piece --
# 3.1e9 stored entries are 37 GB as a scipy CSC (64-bit indices) -- so the data matrix and the null matrix exist as row slabs.
# ------------------------------------------------------------------------------------------------------------------------------
def _slab_cache_path(tag: str, N_total: int, M: int, seed: int, a: int, b: int, cache_dir: Optional[str]):
    import os
    import tempfile

    d = os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir()) if cache_dir is None else cache_dir
    return os.path.join(d, f"sclens_atlas_{tag}_v2_{N_total}x{M}_s{seed}_{a}_{b}.npz") if d not in ("", "0") else None


def _load_slab(path, shape):
    z = np.load(path)
    return sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=shape)


def _save_slab(path, X):
    import os

    tmp = f"{path}.{os.getpid()}.tmp.npz"
    np.savez(tmp, data=X.data, indices=X.indices.astype(np.int32), indptr=X.indptr.astype(np.int64))
    os.replace(tmp, path)


def _synth_one(args):
    from .synth import synth_counts_rows

    N_total, M, seed, a, b, path = args
    X = synth_counts_rows(N_total, M, seed, a, b)
    _save_slab(path, X)
    return int(X.nnz)


class SlabSet:
    """`world` row slabs of one N_total x M matrix, kept as .npz files (`cache_dir`, default SCLENS_BENCH_CACHE or the temp dir) and
    loaded one at a time: `slab(g)` -> scipy CSC with LOCAL row indices (int32), `rows(g)` -> its cell range."""

    def __init__(self, N_total: int, M: int, world: int, paths, nnz):
        self.N_total, self.M, self.world, self.paths, self.nnz = N_total, M, world, list(paths), [int(v) for v in nnz]

    def rows(self, g: int):
        return row_block(g, self.world, self.N_total)

    def slab(self, g: int) -> sp.csc_matrix:
        a, b = self.rows(g)
        return _load_slab(self.paths[g], (b - a, self.M))

    @property
    def nnz_total(self) -> int:
        return int(sum(self.nnz))


def synth_slabs(N_total: int, M: int, seed: int, world: int, cache_dir: Optional[str] = None, workers: Optional[int] = None,
                log=None) -> SlabSet:
    """The seeded synthetic count matrix of `sclens_amd.synth.synth_counts_rows` (a function of (N_total, M, seed) only: the slab
    boundaries do not enter) as `world` slab files, generated by `workers` processes (default: one per 24 GB of available memory -- a
    125 000 x 30 000 slab peaks at ~20 GB while its COO triplets are sorted into CSC --, at
    most the CPUs this process may use; ~80 s per 125 000 x 30 000 slab and process)."""
    import multiprocessing as mp
    import os

    todo, paths = [], []
    for g in range(world):
        a, b = row_block(g, world, N_total)
        p = _slab_cache_path("data", N_total, M, seed, a, b, cache_dir)
        if p is None:
            raise ValueError("the atlas slabs need a cache directory (SCLENS_BENCH_CACHE must not be 0)")
        paths.append(p)
        if not os.path.exists(p):
            todo.append((N_total, M, seed, a, b, p))
    if todo:
        if workers is None:
            try:
                import psutil

                mem = psutil.virtual_memory().available / 1e9
            except Exception:
                mem = 32.0
            workers = max(1, min(len(os.sched_getaffinity(0)), len(todo), int(mem // 24)))
        if log:
            log(f"generating {len(todo)} slab(s) with {workers} process(es)")
        if workers > 1:
            with mp.get_context("fork").Pool(workers) as pool:
                pool.map(_synth_one, todo, chunksize=1)
        else:
            for t in todo:
                _synth_one(t)
    nnz = []
    for g, p in enumerate(paths):
        with np.load(p) as z:
            nnz.append(int(z["indptr"][-1]))
    return SlabSet(N_total, M, world, paths, nnz)


def null_slabs(data: SlabSet, seed: int, cache_dir: Optional[str] = None, log=None) -> SlabSet:
    """R2 (scLENS.jl:701 `random_nz(pre_df, rmix=true)` -> :261-289, :239-248) of the WHOLE matrix: the stored values of all cells
    shuffled globally, every gene keeps its number of entries at distinct cells drawn uniformly from ALL N_total cells -- the library's
    own generator (`sclens_draw_null_matrix`, seed + 1 as in `api.make_draws_native`) on the concatenation of the slabs in global CSC
    order (gene by gene, cells ascending), then cut into the same row slabs. A function of the matrix and the seed only: the slab
    boundaries do not enter. Peak host memory ~ 13 bytes per stored entry (41 GB at 1M x 30k); the data slabs are read from their files
    one at a time."""
    import ctypes as C
    import os

    N_total, M, world = data.N_total, data.M, data.world
    paths = []
    for g in range(world):
        a, b = data.rows(g)
        paths.append(_slab_cache_path("null", N_total, M, seed, a, b, cache_dir))
    if all(os.path.exists(p) for p in paths):
        nnz = []
        for p in paths:
            with np.load(p) as z:
                nnz.append(int(z["indptr"][-1]))
        return SlabSet(N_total, M, world, paths, nnz)
    lib = _lib.load()
    cnt = np.zeros((world, M), dtype=np.int64)
    for g in range(world):
        with np.load(data.paths[g]) as z:
            cnt[g] = np.diff(z["indptr"])
    gcol = np.zeros(M + 1, dtype=np.int64)
    gcol[1:] = np.cumsum(cnt.sum(axis=0))
    nnz = int(gcol[-1])
    nz = np.empty(nnz, dtype=np.float32)
    pref = np.zeros(M, dtype=np.int64)
    for g in range(world):
        s = data.slab(g)
        ip = s.indptr.astype(np.int64)
        dest = np.arange(s.nnz, dtype=np.int64)
        dest += np.repeat(gcol[:-1] + pref - ip[:-1], cnt[g])
        nz[dest] = s.data
        pref += cnt[g]
        del s, dest
    if log:
        log(f"null draw over {nnz} stored entries")
    rrow = np.empty(nnz, dtype=np.int32)
    rval = np.empty(nnz, dtype=np.float32)
    from ._lib import ptr

    rc = lib.sclens_draw_null_matrix(N_total, M, ptr(gcol, C.c_int64), ptr(nz, C.c_float), (int(seed) + 1) & 0xFFFFFFFFFFFFFFFF,
                                     ptr(rrow, C.c_int32), ptr(rval, C.c_float))
    if rc:
        raise RuntimeError(f"sclens_draw_null_matrix: {rc}")
    del nz
    bounds = np.array([data.rows(g)[0] for g in range(world)] + [N_total], dtype=np.int32)
    cuts = np.empty((M, world + 1), dtype=np.int64)  # per gene: first entry of each slab (rows ascend inside a gene)
    for j in range(M):
        cuts[j] = np.searchsorted(rrow[gcol[j]: gcol[j + 1]], bounds)
    out_nnz = []
    for g in range(world):
        a, b = data.rows(g)
        c = cuts[:, g + 1] - cuts[:, g]
        ip = np.zeros(M + 1, dtype=np.int64)
        ip[1:] = np.cumsum(c)
        src = np.arange(int(ip[-1]), dtype=np.int64)
        src += np.repeat(gcol[:-1] + cuts[:, g] - ip[:-1], c)
        rows = rrow[src] - np.int32(a)
        Xg = sp.csc_matrix((rval[src], rows, ip), shape=(b - a, M))
        _save_slab(paths[g], Xg)
        out_nnz.append(int(ip[-1]))
        del src, rows, Xg
    return SlabSet(N_total, M, world, paths, out_nnz)


# ------------------------------------------------------------------------------------------------------------------------------
# scLENS.sclens on ONE GPU for a matrix that only fits in chunks (BASELINE configs[4] on one MI355X): the chunked session
# (include/sclens_hip.h, sclens_hip_session_create_chunked). The serial loop of scLENS.jl:649-832; every decomposition sums the Gram
# contributions of the chunks of cells (:332-361 as a sum over cell blocks).
# ------------------------------------------------------------------------------------------------------------------------------
def chunked_session(ctx: Context, data, cand_seed: int, log=None) -> Session:
    """`data`: a SlabSet, or a list of (row0, csc) chunks of consecutive cells. The host holds one chunk at a time."""
    if isinstance(data, SlabSet):
        N, M, n = data.N_total, data.M, data.world
        nnz = data.nnz_total
        chunks = ((data.rows(g)[0], data.slab(g)) for g in range(n))
    else:
        data = list(data)
        N = sum(int(X.shape[0]) for _, X in data)
        M, n = int(data[0][1].shape[1]), len(data)
        nnz = sum(int(X.nnz) for _, X in data)
        chunks = iter(data)
    ses = Session.create_chunked(ctx, N, M, n, nnz, cand_seed)
    try:
        for g, (row0, X) in enumerate(chunks):
            ses.chunk_add(0, g, row0, _csc_f32(X))
            if log:
                log(f"chunk {g}: cells [{row0}, {row0 + X.shape[0]}), {X.nnz} stored entries")
            del X
        ses.chunk_commit()
    except BaseException:
        ses.close()
        raise
    return ses


def sclens_chunked(data, null, draws: Draws, th=60, p_step=0.001, n_perturb=20, ctx: Optional[Context] = None,
                   max_search_iters: Optional[int] = None, verbose: bool = False, guard_band: float = 4.0, log=None,
                   stop_after: Optional[str] = None) -> Dict[str, object]:
    """scLENS.sclens (scLENS.jl:649-832) on a chunked session. `data` / `null`: SlabSets (or lists of (row0, csc)) of the count matrix
    and of the null matrix X_r cut into the same blocks of cells; `draws`: cand_seed (R1 drawn on the device, chunk by chunk),
    sample_seed (R4 / R5: keyed permutations of the global candidate list), p_th. Returns the reference's result keys plus per-phase
    wall times. stop_after = "spectra": stop after the data / null spectra, lambda_c and the signal count (what the float64 fixture pins)."""
    ctx = ctx or default_context()
    t_all = time.perf_counter()
    phases: Dict[str, float] = {}

    def lap(name, t0):
        ctx.sync()
        phases[name] = round(time.perf_counter() - t0, 3)
        if log:
            log(f"{name}: {phases[name]} s")

    if draws.cand_seed is None:
        raise ValueError("sclens_chunked draws the zero candidates on the device: draws.cand_seed must be set")
    t0 = time.perf_counter()
    ses = chunked_session(ctx, data, draws.cand_seed, log=log if verbose else None)
    lap("session_create", t0)
    N, M = ses.N, ses.M
    try:
        t0 = time.perf_counter()
        nullc = ((null.rows(g)[0], null.slab(g)) for g in range(null.world)) if isinstance(null, SlabSet) else iter(null)
        for g, (row0, X) in enumerate(nullc):
            ses.chunk_add(1, g, row0, _csc_f32(X))
            del X
        Lr = ses.null_spectrum_chunked()  # :704
        lap("null_spectrum", t0)
        t0 = time.perf_counter()
        L, rec_vals = ses.data_spectrum()
        lap("data_spectrum", t0)
        L_mp, _, _ = _mp_calculation(L, Lr[:-1])
        lambda_c = _tw(L, L_mp)[0]
        t0 = time.perf_counter()
        L, k, nL, guard = cut_with_guard_band(L, lambda_c, guard_band, ses.refine_eigenvalues)
        lap("guard_band", t0)
        if verbose:
            print(f"(Using hip, {ses.get_int('chunks')} chunks of cells) number of signal ev: {k}")
        res: Dict[str, object] = {"L": L, "Lr": Lr, "L_mp": L_mp, "λ": lambda_c, "lambda_c": lambda_c, "k": k, "guard_band": guard,
                                  "rec_vals": rec_vals, "phase_s": phases, "chunks": ses.get_int("chunks")}
        if stop_after == "spectra":
            res["chunk_builds"], res["chunk_visits"] = ses.get_int("chunk_builds"), ses.get_int("chunk_visits")
            res["wall_s"] = time.perf_counter() - t_all
            return res
        t0 = time.perf_counter()
        nV = ses.signal_vectors(k)
        lap("signal_vectors", t0)
        t0 = time.perf_counter()
        _, r_vr2 = ses.binary_basis()  # :717-721
        lap("binary_basis", t0)
        mpC = mp_check(L_mp)
        p_th = draws.p_th
        n_2 = int(round(r_vr2 / 2))
        p_list = search_schedule(p_step)
        t0 = time.perf_counter()
        n_cand = ses.get_int("n_cand")
        lap("candidate_count", t0)
        tank = np.zeros((5, 0))
        it, p_ = 0, None
        t0 = time.perf_counter()
        while p_ is None:  # :725-761
            nnzidx = int(round((1 - p_list[it]) * M * N))
            d5 = None
            if n_cand >= nnzidx:
                if draws.sampler is not None:
                    d5, _ = ses.search_step(draws.sampler("search", it, n_cand, nnzidx), n_2)
                else:
                    d5, _ = ses.search_step_seeded(sample_seed_for(draws.sample_seed, "search", it), nnzidx, n_2)
            tank, used, stopped, p_fin = consume_search_round(tank, [d5], p_list, it, p_th, p_step, max_search_iters)
            it += used
            if log:
                log(f"search evaluation {it}: d5[1] = {None if d5 is None else round(float(d5[1]), 6)} (p_th {p_th:.6f})")
            if stopped:
                p_ = p_fin
        lap("sparsity_search", t0)
        trace = [(p_list[q], tank[:, q].copy()) for q in range(tank.shape[1])]
        min_pc = int(math.ceil(k * 1.5))
        m_pert = int(round((1 - p_) * M * N))
        res.update({"p_": p_, "p_th": p_th, "n_search": it, "search_trace": trace, "n_cand": n_cand})
        if k == 0:  # :780-784
            res["wall_s"] = time.perf_counter() - t_all
            return res
        nL_set, ncols = [None] * n_perturb, [0] * n_perturb
        t0 = time.perf_counter()
        for t in range(n_perturb):  # :767-778
            if draws.sampler is not None:
                nL_set[t], ncols[t] = ses.perturb(t, draws.sampler("perturb", t, n_cand, m_pert), min_pc)
            else:
                nL_set[t], ncols[t] = ses.perturb_seeded(t, sample_seed_for(draws.sample_seed, "perturb", t), m_pert, min_pc)
        lap("perturbation_ensemble", t0)
        t0 = time.perf_counter()
        a_b, b_ = ses.robustness(k, n_perturb)  # :786-807
        m_score, sd_score = _robust_scores(b_)
        sig_id = np.flatnonzero(m_score > math.cos(math.radians(th)))
        gmat = ses.gene_basis(nL)
        lap("robustness_gene_basis", t0)
        res.update({"pca": nV * np.sqrt(nL)[None, :].astype(np.float32),
                    "pca_n1": nV[:, sig_id] * np.sqrt(nL[sig_id])[None, :].astype(np.float32), "sig_id": sig_id,
                    "robustness_scores": {"b_": b_, "rob_score": m_score, "m_scores": m_score, "sd_scores": sd_score, "a_b": a_b},
                    "signal_evec": nV, "signal_ev": nL, "gene_basis": gmat, "pass": mpC["pass"], "ks_static": mpC["ks_static"],
                    "nL_set": nL_set, "min_pc": min_pc,
                    "partial_eig": (ses.get_int("chefsi_used"), ses.get_int("chefsi_fallback")),
                    "chunk_builds": ses.get_int("chunk_builds"), "chunk_visits": ses.get_int("chunk_visits")})
        res["wall_s"] = time.perf_counter() - t_all
        return res
    finally:
        ses.close()
