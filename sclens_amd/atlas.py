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
binarised) are still replicated. Memory scales with the ranks either way: the sparse pattern, the candidates (drawn locally:
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
    try:
        Lr = shard.agree(ses.null_spectrum(Xr_local))  # :704
        L, rec_vals = ses.data_spectrum()
        L = shard.agree(L)
        L_mp, _, _ = _mp_calculation(L, Lr[:-1])
        lambda_c = _tw(L, L_mp)[0]
        # the same guard band as api.sclens: every rank refines (the quotient is summed over the row blocks inside the library)
        L, k, nL, guard = cut_with_guard_band(L, lambda_c, guard_band, lambda lo, hi: shard.agree(ses.refine_eigenvalues(lo, hi)))
        if verbose and shard.rank == 0:
            print(f"(Using hip, {shard.world} row blocks) number of signal ev: {k}")
        nV_local = ses.signal_vectors(k)
        _, r_vr2 = ses.binary_basis()  # :717-721
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
