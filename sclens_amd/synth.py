"""Seeded synthetic count matrices for parity tests and `bench.py`.

Poisson-lognormal counts with `C` planted clusters, tuned to a target sparsity, with the
`preprocess` invariants of the reference enforced (scLENS.jl:160-162: every cell expresses
>= 200 genes (scaled down for tiny matrices), every gene is seen in >= 15 cells), so that
`TGC > 0` (scLENS.jl:678) and every per-gene std is > 0 (scLENS.jl:683).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _rates(M, C, rng, marker_frac=0.05, marker_sd=1.0):
    beta0 = np.clip(rng.normal(-1.5, 1.5, size=M), -6.0, 3.0)
    delta = np.zeros((C, M))
    for c in range(C):
        mk = rng.random(M) < marker_frac
        delta[c, mk] = rng.normal(0.0, marker_sd, size=int(mk.sum()))
    return beta0, delta


def _sparsity_for_offset(beta0, off, lib_sigma=0.3):
    # E[P(x = 0)] ignoring the marker shifts and library-size spread (good enough for tuning)
    return float(np.mean(np.exp(-np.exp(beta0 + off))))


def synth_counts(N: int, M: int, seed: int, C: int = 8, sparsity: float = 0.90,
                 min_genes_per_cell: int | None = None, min_cells_per_gene: int | None = None,
                 chunk_rows: int = 4096, marker_frac: float = 0.05, marker_sd: float = 1.0) -> sp.csc_matrix:
    """Return an N x M CSC float32 matrix of integer counts (cells x genes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    beta0, delta = _rates(M, C, rng, marker_frac, marker_sd)
    lo, hi = -8.0, 8.0
    for _ in range(60):  # bisection on a global log-rate offset
        mid = 0.5 * (lo + hi)
        if _sparsity_for_offset(beta0, mid) > sparsity:
            lo = mid
        else:
            hi = mid
    beta0 = beta0 + 0.5 * (lo + hi)
    labels = np.arange(N) % C
    rng.shuffle(labels)
    lib = rng.lognormal(0.0, 0.3, size=N)
    rows, cols, vals = [], [], []
    for r0 in range(0, N, chunk_rows):
        r1 = min(N, r0 + chunk_rows)
        lam = lib[r0:r1, None] * np.exp(beta0[None, :] + delta[labels[r0:r1]])
        x = rng.poisson(lam).astype(np.float32)
        i, j = np.nonzero(x)
        rows.append((i + r0).astype(np.int64))
        cols.append(j.astype(np.int64))
        vals.append(x[i, j])
    X = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(N, M), dtype=np.float32)
    # enforce the QC invariants by planting single counts where needed (deterministic given seed)
    mg = min(200, max(2, M // 20)) if min_genes_per_cell is None else min_genes_per_cell
    mc = min(15, max(2, N // 20)) if min_cells_per_gene is None else min_cells_per_gene
    # (planted entries are collected as coordinates and added in one sparse sum: a LIL round trip of a 3e8-entry matrix
    # cost a third of the generation time; the draws and the result are the same)
    row_nnz = np.bincount(X.indices, minlength=N)
    short_rows = np.flatnonzero(row_nnz < mg)
    if len(short_rows):
        Xr = X.tocsr()
        pi, pj = [], []
        for i in short_rows:
            need = mg - row_nnz[i]
            zero_cols = np.setdiff1d(np.arange(M), Xr.indices[Xr.indptr[i]:Xr.indptr[i + 1]])
            pj.append(rng.choice(zero_cols, size=need, replace=False))
            pi.append(np.full(need, i, dtype=np.int64))
        pi, pj = np.concatenate(pi), np.concatenate(pj)
        X = (X + sp.csc_matrix((np.ones(len(pi), np.float32), (pi, pj)), shape=(N, M))).tocsc()
        X.sort_indices()
    col_nnz = np.diff(X.indptr)
    short_cols = np.flatnonzero(col_nnz < mc)
    if len(short_cols):
        pi, pj = [], []
        for j in short_cols:
            present = X.indices[X.indptr[j]:X.indptr[j + 1]]
            zero_rows = np.setdiff1d(np.arange(N), present)
            pi.append(rng.choice(zero_rows, size=mc - len(present), replace=False))
            pj.append(np.full(mc - len(present), j, dtype=np.int64))
        pi, pj = np.concatenate(pi), np.concatenate(pj)
        X = (X + sp.csc_matrix((np.ones(len(pi), np.float32), (pi, pj)), shape=(N, M))).tocsc()
    X.sort_indices()
    return X.astype(np.float32)


def synth_counts_rows(N_total: int, M: int, seed: int, row0: int, row1: int, C: int = 8, sparsity: float = 0.90,
                      chunk_rows: int = 4096, marker_frac: float = 0.05, marker_sd: float = 1.0,
                      min_genes_per_cell: int = 200) -> sp.csc_matrix:
    """Rows [row0, row1) of an N_total x M matrix of the same model as `synth_counts`, for matrices that are never held
    in one piece (the 1M x 30k atlas configuration is generated chunk-wise). The gene parameters, cluster
    labels and library sizes come from the generator seeded with `seed`; the Poisson draws of chunk c (rows
    [c * chunk_rows, (c + 1) * chunk_rows)) from a generator seeded with (seed, c), so any range of rows is reproducible on
    its own and only one chunk is ever dense on the host. Cells with fewer than `min_genes_per_cell` expressed genes get
    single counts planted (scLENS.jl:160-162); the per-gene minimum is a property of the whole matrix and is not enforced
    on a slab."""
    rng = np.random.Generator(np.random.PCG64(seed))
    beta0, delta = _rates(M, C, rng, marker_frac, marker_sd)
    lo, hi = -8.0, 8.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if _sparsity_for_offset(beta0, mid) > sparsity:
            lo = mid
        else:
            hi = mid
    beta0 = beta0 + 0.5 * (lo + hi)
    labels = np.arange(N_total) % C
    rng.shuffle(labels)
    lib = rng.lognormal(0.0, 0.3, size=N_total)
    mg = min(min_genes_per_cell, max(2, M // 20))
    rows, cols, vals = [], [], []
    c0 = row0 // chunk_rows
    for c in range(c0, (row1 + chunk_rows - 1) // chunk_rows):
        a, b = c * chunk_rows, min(N_total, (c + 1) * chunk_rows)
        crng = np.random.Generator(np.random.PCG64([seed, c]))
        lam = lib[a:b, None] * np.exp(beta0[None, :] + delta[labels[a:b]])
        x = crng.poisson(lam).astype(np.float32)
        short = np.flatnonzero((x > 0).sum(axis=1) < mg)
        for i in short:  # plant single counts (deterministic given seed and chunk)
            zero_cols = np.flatnonzero(x[i] == 0)
            x[i, crng.choice(zero_cols, size=mg - int((x[i] > 0).sum()), replace=False)] = 1.0
        s0, s1 = max(a, row0) - a, min(b, row1) - a
        i, j = np.nonzero(x[s0:s1])
        rows.append((i + (a + s0 - row0)).astype(np.int32))  # (32-bit triplets: a 125 000 x 30 000 slab peaked at 23 GB with 64-bit ones)
        cols.append(j.astype(np.int32))
        vals.append(x[s0:s1][i, j])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    X = sp.csc_matrix((vals, (rows, cols)), shape=(row1 - row0, M), dtype=np.float32)
    del rows, cols, vals
    X.sort_indices()
    return X
