"""Host-side mirror of the reference's operator interface for the sclens() hot path, over the C ABI.

Julia is not available in the build image, so this Python module plays the role of the thin Julia shim
(`julia/scLENS_hip.jl`): same function names, argument meaning and error behaviour as
`/root/reference/src/scLENS.jl`, every heavy operation forwarded to libsclens_hip.so. There is no CPU
fallback here and nothing in this file imports `oracle/`.

  _wishart_matrix, corr_mat, _get_eigen, get_eigvec      scLENS.jl:332-387, :489-524 (per-call drop-ins)
  _mp_calculation, _tw, mp_check                          scLENS.jl:424-487 (host statistics, C++)
  sclens                                                  scLENS.jl:649-832 (device-resident session)
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import time
from concurrent.futures import Future, ThreadPoolExecutor
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import Context, SclensHipError, ptr
from .shard import Shard, consume_search_round, owned_perturbations, search_schedule

_default_ctx: Optional[Context] = None


def default_context(device: int = 0) -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(device)
    return _default_ctx


def _f32_colmajor(X) -> np.ndarray:
    return np.asfortranarray(np.asarray(X, dtype=np.float32))


# ----------------------------------------------------------------------------- per-call drop-ins
def _wishart_matrix(X, device="gpu", dims=1, ctx: Optional[Context] = None) -> np.ndarray:
    """scLENS.jl:332-361. Unknown device strings return None like the reference (SURVEY 8a defect 5)."""
    if device != "gpu":
        return None
    ctx = ctx or default_context()
    Xf = _f32_colmajor(X)
    N, M = Xf.shape
    n = M if dims == 2 else N
    Y = np.empty((n, n), dtype=np.float32, order="F")
    ctx.check(ctx.lib.sclens_hip_wishart_matrix_f32(ctx.h, ptr(Xf, C.c_float), N, M, int(dims), ptr(Y, C.c_float)))
    return Y


def corr_mat(X, Y, device="gpu", ctx: Optional[Context] = None) -> np.ndarray:
    """scLENS.jl:363-373: X' * Y."""
    if device != "gpu":
        return None
    ctx = ctx or default_context()
    Xf, Yf = _f32_colmajor(X), _f32_colmajor(Y)
    n, p = Xf.shape
    q = Yf.shape[1]
    out = np.empty((p, q), dtype=np.float32, order="F")
    ctx.check(ctx.lib.sclens_hip_corr_mat_f32(ctx.h, ptr(Xf, C.c_float), n, p, ptr(Yf, C.c_float), q, ptr(out, C.c_float)))
    return out


def _get_eigen(Y, device="gpu", ctx: Optional[Context] = None):
    """scLENS.jl:375-387: (values ascending, vectors as columns)."""
    if device != "gpu":
        return None
    ctx = ctx or default_context()
    Yf = _f32_colmajor(Y)
    n = Yf.shape[0]
    L = np.empty(n, dtype=np.float32)
    V = np.empty((n, n), dtype=np.float32, order="F")
    ctx.check(ctx.lib.sclens_hip_get_eigen_f32(ctx.h, ptr(Yf, C.c_float), n, ptr(L, C.c_float), ptr(V, C.c_float)))
    return L, V


def get_eigvec(X, device="gpu", keep_top: int = 0, ctx: Optional[Context] = None):
    """scLENS.jl:489-524: (nL descending, N x r cell-side unit eigenvectors)."""
    if device != "gpu":
        return None
    ctx = ctx or default_context()
    Xf = _f32_colmajor(X)
    N, M = Xf.shape
    n = min(N, M)
    nL = np.empty(n, dtype=np.float32)
    ncol = n if keep_top <= 0 else min(keep_top, n)
    nV = np.empty((N, ncol), dtype=np.float32, order="F")
    r = C.c_int64(n)
    ctx.check(ctx.lib.sclens_hip_get_eigvec_f32(ctx.h, ptr(Xf, C.c_float), N, M, int(keep_top), ptr(nL, C.c_float),
                                                ptr(nV, C.c_float), C.byref(r)))
    rr = r.value
    return nL[:rr], nV[:, : min(rr, ncol)]


def preprocess(X, gene_names, cell_names=None, min_tp_c=0, min_tp_g=0, max_tp_c=np.inf, max_tp_g=np.inf,
               min_genes_per_cell=200, max_genes_per_cell=0, min_cells_per_gene=15, mito_percent=5.0, ribo_percent=0.0,
               ctx: Optional[Context] = None, keep_on_device: bool = False):
    """scLENS.preprocess (scLENS.jl:160-236) on the device: QC-filter a raw cells x genes count matrix. Same keyword
    arguments and defaults as the reference. Returns (filtered CSC float32 with genes sorted by mean count, gene names,
    cell names or indices) or None when no cell or gene passes (the reference prints a message and returns nothing).
    `keep_on_device=True`: the first item is a `DeviceCounts` handle (the matrix stays in HBM) that `sclens()` accepts directly."""
    import re

    ctx = ctx or default_context()
    Xc = _csc_f32(X)
    N, M = Xc.shape
    names = np.asarray(gene_names)
    if len(names) != M:
        raise ValueError("gene_names must have one entry per column")
    is_mito = np.array([bool(re.match(r"(?i)^mt-.", str(g))) for g in names], dtype=np.uint8)  # :194
    is_ribo = np.array([bool(re.match(r"(?i)^RP[SL].", str(g))) for g in names], dtype=np.uint8)  # :195
    colptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(Xc.indices, dtype=np.int32)
    nzval = np.ascontiguousarray(Xc.data, dtype=np.float32)
    keep_cell = np.zeros(N, dtype=np.uint8)
    order = np.zeros(M, dtype=np.int64)
    nc, ng, nnz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    big = 1.7976931348623157e308  # Inf thresholds cross the C ABI as finite doubles compare the same way
    ctx.check(ctx.lib.sclens_hip_preprocess_csc(
        ctx.h, N, M, ptr(colptr, C.c_int64), ptr(rowval, C.c_int32), ptr(nzval, C.c_float), ptr(is_mito, C.c_uint8),
        ptr(is_ribo, C.c_uint8), float(min_tp_c), float(min_tp_g), float(min(max_tp_c, big)), float(min(max_tp_g, big)),
        int(min_genes_per_cell), int(max_genes_per_cell), int(min_cells_per_gene), float(mito_percent), float(ribo_percent),
        ptr(keep_cell, C.c_uint8), ptr(order, C.c_int64), C.byref(nc), C.byref(ng), C.byref(nnz)))
    if nc.value == 0 or ng.value == 0:
        return None
    cells = np.flatnonzero(keep_cell)
    if keep_on_device:  # the filtered matrix stays in HBM; sclens() takes the handle as it takes a matrix
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_preprocess_keep(ctx.h, C.byref(h)))
        return DeviceCounts(ctx, h), names[order[: ng.value]], (np.asarray(cell_names)[cells] if cell_names is not None else cells)
    out_colptr = np.empty(ng.value + 1, dtype=np.int64)
    out_row = np.empty(nnz.value, dtype=np.int32)
    out_val = np.empty(nnz.value, dtype=np.float32)
    ctx.check(ctx.lib.sclens_hip_preprocess_gather(ctx.h, ptr(out_colptr, C.c_int64), ptr(out_row, C.c_int32),
                                                   ptr(out_val, C.c_float)))
    Xo = sp.csc_matrix((out_val, out_row, out_colptr), shape=(nc.value, ng.value))
    Xo._sclens_canonical = True  # sorted rows, no explicit zeros, float32
    return Xo, names[order[: ng.value]], (np.asarray(cell_names)[cells] if cell_names is not None else cells)


def logn_scale(X, centering: str = "mean", inline_f64: bool = False, device="gpu", ctx: Optional[Context] = None):
    """`logn_scale(pre_scale(x))` of scLENS.jl:650-654 (centering "mean" or "median") on the device: counts (N x M, any
    scipy-sparse / dense) -> dense scaled N x M float32. `inline_f64=True` is the inline Float64 twin of the data matrix
    (:676-696) and also returns rec_vals."""
    if device != "gpu":
        return None
    if centering not in ("mean", "median"):
        raise NotImplementedError(f"centering={centering!r}")
    if inline_f64 and centering != "mean":
        raise ValueError("the inline Float64 path with rec_vals exists for centering='mean' only (scLENS.jl:676-698)")
    ctx = ctx or default_context()
    Xc = _csc_f32(X)
    N, M = Xc.shape
    colptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(Xc.indices, dtype=np.int32)
    nzval = np.ascontiguousarray(Xc.data, dtype=np.float32)
    out = np.empty((N, M), dtype=np.float32, order="F")
    if inline_f64:
        rec = {"TGC": np.empty(N), "mat2_mean": np.empty(M), "mat2_std": np.empty(M), "norm_tgc": np.empty(N),
               "cent_": np.empty(M)}
        recp = [ptr(rec[k], C.c_double) for k in ("TGC", "mat2_mean", "mat2_std", "norm_tgc", "cent_")]
    else:
        rec, recp = None, [None] * 5
    ctx.check(ctx.lib.sclens_hip_scale_csc_f32(ctx.h, N, M, ptr(colptr, C.c_int64), ptr(rowval, C.c_int32),
                                               ptr(nzval, C.c_float), 1 if centering == "median" else 0,
                                               0 if inline_f64 else 1, ptr(out, C.c_float), *recp))
    return (out, rec) if inline_f64 else out


def _corr_colmax(X, Y, use_split: bool, ctx: Optional[Context] = None) -> np.ndarray:
    """Unit-test piece (sclens_hip_corr_colmax_f32): max over the columns of X of |X' Y| per column of Y."""
    ctx = ctx or default_context()
    Xf = np.asfortranarray(X, dtype=np.float32)
    Yf = np.asfortranarray(Y, dtype=np.float32)
    n, p = Xf.shape
    q = Yf.shape[1]
    out = np.empty(q, dtype=np.float32)
    ctx.check(ctx.lib.sclens_hip_corr_colmax_f32(ctx.h, ptr(Xf, C.c_float), n, p, ptr(Yf, C.c_float), q, 1 if use_split else 0,
                                                 ptr(out, C.c_float)))
    return out


def _gram_binary(X, use_bits: bool, divisor: Optional[float] = None, ctx: Optional[Context] = None) -> np.ndarray:
    """Unit-test piece (sclens_hip_gram_binary_f32): M x M Gram matrix of logn_scale(pre_scale(P)) / divisor for the
    binarised counts P of an N > M matrix; `use_bits` picks the fp16-MFMA co-occurrence product of the sparsity search
    (csrc/gram_bits.hip) instead of the scaled matrix + fp32 product."""
    ctx = ctx or default_context()
    Xc = _csc_f32(X)
    N, M = Xc.shape
    colptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(Xc.indices, dtype=np.int32)
    nzval = np.ascontiguousarray(Xc.data, dtype=np.float32)
    out = np.empty((M, M), dtype=np.float32)
    ctx.check(ctx.lib.sclens_hip_gram_binary_f32(ctx.h, N, M, ptr(colptr, C.c_int64), ptr(rowval, C.c_int32), ptr(nzval, C.c_float),
                                                 1 if use_bits else 0, float(N if divisor is None else divisor), ptr(out, C.c_float)))
    return out


def _gram_counts(X, mode: int, f32path: bool = True, binary: bool = False, divisor: Optional[float] = None, ctx: Optional[Context] = None) -> np.ndarray:
    """Unit-test / A-B piece (sclens_hip_gram_counts_f32): M x M Gram matrix of the scaled count matrix, mode 0 = dense product, 1 = from
    the sparse structure."""
    ctx = ctx or default_context()
    Xc = _csc_f32(X)
    N, M = Xc.shape
    colptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
    rowval = np.ascontiguousarray(Xc.indices, dtype=np.int32)
    nzval = np.ascontiguousarray(Xc.data, dtype=np.float32)
    out = np.empty((M, M), dtype=np.float32)
    ctx.check(ctx.lib.sclens_hip_gram_counts_f32(ctx.h, N, M, ptr(colptr, C.c_int64), ptr(rowval, C.c_int32), ptr(nzval, C.c_float), int(mode),
                                                 1 if f32path else 0, 1 if binary else 0, float(divisor if divisor is not None else M),
                                                 ptr(out, C.c_float)))
    return out


def get_denoised_df(inp_obj: Dict[str, object], device_="gpu", ctx: Optional[Context] = None) -> np.ndarray:
    """scLENS.jl:889-931: denoised count means from the robust signals of an sclens() result (N x M array; the
    reference wraps it in a DataFrame with `gene_id` columns and a `cell` column)."""
    if device_ != "gpu":
        raise NotImplementedError("sclens_amd implements the device path only")
    ctx = ctx or default_context()
    sig = np.asarray(inp_obj["sig_id"], dtype=np.int64)
    if sig.size == 0:
        raise ValueError("no robust signal to reconstruct from")
    g_mat = np.ascontiguousarray(np.asarray(inp_obj["gene_basis"], dtype=np.float32)[sig, :])
    Xout0 = np.asfortranarray(np.asarray(inp_obj["pca_n1"], dtype=np.float32))
    N, s_ = Xout0.shape
    M = g_mat.shape[1]
    rv = inp_obj["rec_vals"]
    vec = {k: np.ascontiguousarray(np.ravel(rv[k]), dtype=np.float64) for k in ("TGC", "mat2_mean", "mat2_std", "norm_tgc", "cent_")}
    out = np.empty((N, M), dtype=np.float32, order="F")
    ctx.check(ctx.lib.sclens_hip_get_denoised_f32(ctx.h, ptr(Xout0, C.c_float), N, s_, ptr(g_mat, C.c_float), M,
                                                  ptr(vec["TGC"], C.c_double), ptr(vec["mat2_mean"], C.c_double),
                                                  ptr(vec["mat2_std"], C.c_double), ptr(vec["norm_tgc"], C.c_double),
                                                  ptr(vec["cent_"], C.c_double), ptr(out, C.c_float)))
    return out


# ----------------------------------------------------------------------------- host statistics
def _mp_calculation(L, Lr):
    """scLENS.jl:424-459 -> (L_mp, b_plus, b_minus)."""
    lib = _lib.load()
    L = np.ascontiguousarray(L, dtype=np.float64)
    Lr = np.ascontiguousarray(Lr, dtype=np.float64)
    mask = np.zeros(L.size, dtype=np.uint8)
    bp, bm = C.c_double(0), C.c_double(0)
    rc = lib.sclens_mp_calculation(ptr(L, C.c_double), L.size, ptr(Lr, C.c_double), Lr.size, C.byref(bp), C.byref(bm),
                                   ptr(mask, C.c_uint8))
    if rc:
        raise SclensHipError(rc, "sclens_mp_calculation")
    return L[mask.astype(bool)], bp.value, bm.value


def _tw(L, L_mp):
    """scLENS.jl:461-467 -> (lambda_c, gamma, p, sigma)."""
    lib = _lib.load()
    L_mp = np.ascontiguousarray(L_mp, dtype=np.float64)
    out = [C.c_double(0) for _ in range(4)]
    rc = lib.sclens_tw(len(L), ptr(L_mp, C.c_double), L_mp.size, *[C.byref(o) for o in out])
    if rc:
        raise SclensHipError(rc, "sclens_tw")
    return tuple(o.value for o in out)


def mp_check(test_L, p_val=0.05):
    """scLENS.jl:469-487."""
    lib = _lib.load()
    t = np.ascontiguousarray(test_L, dtype=np.float64)
    ks, ok = C.c_double(0), C.c_int(0)
    rc = lib.sclens_mp_check(ptr(t, C.c_double), t.size, float(p_val), C.byref(ks), C.byref(ok))
    if rc:
        raise SclensHipError(rc, "sclens_mp_check")
    return {"ks_static": ks.value, "pass": bool(ok.value)}


def _robust_scores(b_):
    lib = _lib.load()
    b = np.ascontiguousarray(b_, dtype=np.float64)
    k, npairs = b.shape
    m = np.empty(k)
    sd = np.empty(k)
    rc = lib.sclens_robust_scores(ptr(b, C.c_double), k, npairs, ptr(m, C.c_double), ptr(sd, C.c_double))
    if rc:
        raise SclensHipError(rc, "sclens_robust_scores")
    return m, sd


# ----------------------------------------------------------------------------- random draws (host)
@dataclass
class Draws:
    """The five random draws of one sclens() call (scLENS.jl:669, :701, :711, :731, :772), 0-based.

    Julia's global RNG stream cannot be reproduced outside Julia; results are statistically equivalent.
    Tests inject the same Draws into the oracle and into this path.
    """

    z_idx1: np.ndarray
    z_idx2: np.ndarray
    X_r: sp.csc_matrix
    p_th: float
    # R4/R5: (kind, iteration, population, m) -> index vector; None = draw on the device from `sample_seed`
    sampler: Optional[Callable[[str, int, int, int], np.ndarray]] = None
    sample_seed: int = 0
    # R1 on the device: z_idx1 / z_idx2 are None and the candidate list is drawn from this seed inside the library
    # (sclens_hip_pattern_create_drawn; sclens_draw_zero_candidates(seed) gives the identical list on the host)
    cand_seed: Optional[int] = None


_M64 = (1 << 64) - 1


def sample_seed_for(base_seed: int, kind: str, it: int) -> int:
    """Seed of the keyed permutation used for the sample of (kind, iteration)."""
    k = 1 if kind == "search" else 2
    return (base_seed * 0x9E3779B97F4A7C15 + k * 0x632BE59BD9B4E019 + (it + 1) * 0xD1B54A32D192ED03) & _M64


def sample_indices(population: int, m: int, seed: int) -> np.ndarray:
    """m distinct indices of [0, population): the host evaluation of the permutation the device uses
    (sclens_sample_without_replacement)."""
    lib = _lib.load()
    out = np.empty(m, dtype=np.uint32)
    rc = lib.sclens_sample_without_replacement(int(population), int(m), int(seed) & _M64, ptr(out, C.c_uint32))
    if rc:
        raise SclensHipError(rc, "sclens_sample_without_replacement")
    return out


def _csc_f32(X) -> sp.csc_matrix:
    if getattr(X, "_sclens_canonical", False):  # already float32 CSC, sorted, no explicit zeros (our own generators)
        return X
    X = sp.csc_matrix(X, dtype=np.float32)
    X.eliminate_zeros()
    X.sort_indices()
    X._sclens_canonical = True
    return X


def draw_zero_candidates(X: sp.csc_matrix, rng: np.random.Generator):
    """scLENS.jl:668-673: nnz uniform (i,j) pairs minus the stored set, first-occurrence order."""
    N, M = X.shape
    nnz = X.nnz
    key = rng.integers(0, N, size=nnz, dtype=np.int64) + rng.integers(0, M, size=nnz, dtype=np.int64) * N
    _, first = np.unique(key, return_index=True)
    first.sort()
    key = key[first]
    nzkey = X.indices.astype(np.int64) + np.repeat(np.arange(M, dtype=np.int64), np.diff(X.indptr)) * N
    key = key[~np.isin(key, nzkey, assume_unique=False)]
    return (key % N).astype(np.uint32), (key // N).astype(np.uint32)


def draw_null_matrix(X: sp.csc_matrix, rng: np.random.Generator) -> sp.csc_matrix:
    """random_nz(pre_df, rmix=true) (scLENS.jl:261-289, :239-248), intent as in SURVEY 8a defect 4:
    stored values shuffled globally; every gene keeps its number of entries at uniformly drawn distinct cells."""
    N, M = X.shape
    vals = rng.permutation(X.data)
    rows = np.empty_like(X.indices)
    ip = X.indptr
    for j in range(M):
        c = ip[j + 1] - ip[j]
        if c:
            rows[ip[j]: ip[j + 1]] = np.sort(rng.choice(N, size=c, replace=False))
    return sp.csc_matrix((vals, rows, ip.copy()), shape=(N, M), dtype=np.float32)


def draw_noise_baseline(n: int, rng: np.random.Generator, trials: int = 5000) -> float:
    """scLENS.jl:709-712."""
    acc = 0.0
    sd = math.sqrt(1.0 / n)
    chunk = max(1, int(2e7 // n))
    done = 0
    while done < trials:
        t = min(chunk, trials - done)
        acc += float(np.abs(rng.standard_normal((t, n))).max(axis=1).sum()) * sd
        done += t
    return acc / trials


def _resolve(x):
    """X_r may be a concurrent.futures.Future (make_draws_native(async_null=True))."""
    return x.result() if hasattr(x, "result") else x


_draw_pool = ThreadPoolExecutor(max_workers=2)


class _PinnedBlock:
    """one page-locked host block (sclens_hip_host_alloc); goes back to the pool's free list when the last array built on it dies"""

    def __init__(self, pool, addr, nbytes):
        self.pool, self.addr, self.nbytes = pool, addr, nbytes

    def __del__(self):
        try:
            self.pool._give_back(self.addr, self.nbytes)
        except Exception:  # interpreter shutdown
            pass


class _PinnedPool:
    """Arrays in page-locked host memory for what the host draws and the library uploads (the null matrix: row indices + values,
    1.1 GB at 100 000 x 30 000 -- from pageable memory the upload took 0.3 s of the 0.6 s the null decomposition waited for its
    pattern). A block is reused by the next call once every array on it has been garbage collected; without a HIP device
    `array()` returns None and the caller takes ordinary memory."""
    CAP_BYTES = 8 << 30  # idle blocks beyond this are freed

    def __init__(self):
        # re-entrant: a garbage collection triggered inside array() / _give_back() may finalise another _PinnedBlock on the same thread,
        # whose __del__ comes back here
        self.lock = threading.RLock()
        self.free = {}  # nbytes -> [address]
        self.idle = 0

    def array(self, count: int, dtype):
        dtype = np.dtype(dtype)
        nbytes = max(1, -(-int(count) * dtype.itemsize // (2 << 20))) * (2 << 20)
        with self.lock:
            lst = self.free.get(nbytes)
            addr = lst.pop() if lst else None
            if addr is not None:
                self.idle -= nbytes
        if addr is None:
            try:
                lib = _lib.load()
                out = C.c_void_p()
                if lib.sclens_hip_host_alloc(nbytes, C.byref(out)) != 0 or not out.value:
                    return None
                addr = out.value
            except Exception:
                return None
        buf = (C.c_char * nbytes).from_address(addr)
        buf._sclens_block = _PinnedBlock(self, addr, nbytes)  # lives as long as `buf`, i.e. as long as any array built on it
        return np.frombuffer(buf, dtype=dtype, count=int(count))

    def _give_back(self, addr, nbytes):
        with self.lock:
            if self.idle + nbytes <= self.CAP_BYTES:
                self.free.setdefault(nbytes, []).append(addr)
                self.idle += nbytes
                return
        _lib.load().sclens_hip_host_free(C.c_void_p(addr))

    def trim(self):
        with self.lock:
            blocks, self.free, self.idle = self.free, {}, 0
        for lst in blocks.values():
            for addr in lst:
                _lib.load().sclens_hip_host_free(C.c_void_p(addr))


_pinned = _PinnedPool()


class _FutureItem:
    """item `i` of a future's tuple result, resolved by `_resolve`"""

    def __init__(self, fut, i):
        self.fut, self.i = fut, i

    def result(self):
        return self.fut.result()[self.i]


def make_draws_native(X, seed: int, host_sampler: bool = False, async_null: bool = False,
                      async_candidates: bool = False, device_candidates: bool = False) -> Draws:
    """All draws from the library's own generators (C++ on the host for R1/R2, the exact expectation for R3 --
    the quantity scLENS.jl:709-712 estimates with 5000 Monte-Carlo trials -- and the device-side keyed permutation
    for R4/R5). `host_sampler=True` materialises the identical R4/R5 index vectors on the host instead."""
    lib = _lib.load()
    if isinstance(X, DeviceCounts):
        # the matrix is in HBM: R1 is drawn there; the host null-matrix generator (R2) needs the gene counts and the values only
        if not device_candidates and not async_candidates:
            device_candidates = True
        N, M = X.shape
        cp, rv, nz = X.download(rowval=not device_candidates)
        nnz_total = X.nnz
    else:
        X = _csc_f32(X)
        N, M = X.shape
        cp = np.ascontiguousarray(X.indptr, dtype=np.int64)
        rv = np.ascontiguousarray(X.indices, dtype=np.int32)
        nz = np.ascontiguousarray(X.data, dtype=np.float32)
        nnz_total = X.nnz

    def candidates():
        z1 = np.empty(nnz_total, dtype=np.uint32)
        z2 = np.empty(nnz_total, dtype=np.uint32)
        cnt = C.c_int64(0)
        rc = lib.sclens_draw_zero_candidates(N, M, ptr(cp, C.c_int64), ptr(rv, C.c_int32), int(seed) & _M64,
                                             ptr(z1, C.c_uint32), ptr(z2, C.c_uint32), C.byref(cnt))
        if rc:
            raise SclensHipError(rc, "sclens_draw_zero_candidates")
        return z1[: cnt.value].copy(), z2[: cnt.value].copy()

    # async_candidates: the candidate list is only needed once the sparsity search starts (sclens() attaches it to the
    # session after the first three decompositions), so it can be drawn on a host thread meanwhile
    if device_candidates:  # R1 is drawn on the device when sclens() builds the union pattern (nothing to do on the host)
        z1 = z2 = None
    elif async_candidates:
        zf = _draw_pool.submit(candidates)
        z1, z2 = _FutureItem(zf, 0), _FutureItem(zf, 1)
    else:
        z1, z2 = candidates()

    def null_matrix():
        rrow = rval = None
        # page-locked blocks for arrays that go to the device next: opt-in (SCLENS_PINNED_DRAWS=1). Measured at 100 000 x 30 000
        # (profiles/r04_pinned_null_matrix.log): the null pattern is on the device 0.8 s earlier (0.40 instead of 1.23 s into the call),
        # but the data | null pair then runs fully concurrently and ends no earlier, and the call was 1.2 s SLOWER on the box it was
        # timed on (the pair started 1.2 s late; not understood, one box) -- off by default.
        if nnz_total >= (1 << 22) and os.environ.get("SCLENS_PINNED_DRAWS", "0") == "1":
            rrow, rval = _pinned.array(nnz_total, np.int32), _pinned.array(nnz_total, np.float32)
        if rrow is None or rval is None:
            rrow = np.empty(nnz_total, dtype=np.int32)
            rval = np.empty(nnz_total, dtype=np.float32)
        rc2 = lib.sclens_draw_null_matrix(N, M, ptr(cp, C.c_int64), ptr(nz, C.c_float), (int(seed) + 1) & _M64,
                                          ptr(rrow, C.c_int32), ptr(rval, C.c_float))
        if rc2:
            raise SclensHipError(rc2, "sclens_draw_null_matrix")
        out = sp.csc_matrix((rval, rrow, cp.copy()), shape=(N, M), dtype=np.float32)
        out._sclens_canonical = True  # rows ascending per gene, values are the (non-zero) stored counts
        return out

    # async_null: X_r is generated on a host thread (the C++ generator releases the GIL) while the caller already
    # builds the session and decomposes the data matrix; sclens() resolves the future when the null spectrum is due
    Xr = _draw_pool.submit(null_matrix) if async_null else null_matrix()
    p_th = float(lib.sclens_noise_baseline_exact(min(N, M)))
    d = Draws(z1, z2, Xr, p_th, None, int(seed))
    if device_candidates:
        d.cand_seed = int(seed) & _M64
    if host_sampler:
        d.sampler = lambda kind, it, population, m: sample_indices(population, m, sample_seed_for(int(seed), kind, it))
    return d


def make_draws(X, seed: int, p_th_trials: int = 5000) -> Draws:
    X = _csc_f32(X)
    rng = np.random.default_rng(seed)
    z1, z2 = draw_zero_candidates(X, rng)
    Xr = draw_null_matrix(X, rng)
    p_th = draw_noise_baseline(min(X.shape), rng, p_th_trials)

    def sampler(kind, it, population, m):
        r = np.random.default_rng([seed, 1 if kind == "search" else 2, it])
        return r.choice(population, size=m, replace=False)

    return Draws(z1, z2, Xr, p_th, sampler)


# ----------------------------------------------------------------------------- session wrapper
class DeviceCounts:
    """sclens_hip_counts: a cells x genes count matrix (CSC) that lives in HBM -- what `preprocess(..., keep_on_device=True)`
    returns instead of a scipy matrix, or `DeviceCounts.upload(ctx, X)`. `sclens()` builds its session and its union pattern
    from it in place (no host round trip between the QC filter and the session)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.h = ctx, handle
        ctx._adopt()
        n, m, z = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        ctx.check(ctx.lib.sclens_hip_counts_info(handle, C.byref(n), C.byref(m), C.byref(z)))
        self.shape, self.nnz = (n.value, m.value), z.value
        self._host = {}

    @classmethod
    def upload(cls, ctx: Context, X) -> "DeviceCounts":
        X = _csc_f32(X)
        cp = np.ascontiguousarray(X.indptr, dtype=np.int64)
        rv = np.ascontiguousarray(X.indices, dtype=np.int32)
        nz = np.ascontiguousarray(X.data, dtype=np.float32)
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_counts_upload(ctx.h, X.shape[0], X.shape[1], ptr(cp, C.c_int64), ptr(rv, C.c_int32),
                                                   ptr(nz, C.c_float), C.byref(h)))
        return cls(ctx, h)

    def download(self, colptr=True, rowval=True, nzval=True):
        """host copies of the requested arrays (cached): (colptr, rowval, nzval), None where not requested"""
        want = {"colptr": colptr, "rowval": rowval, "nzval": nzval}
        need = [k for k, w in want.items() if w and k not in self._host]
        if need:
            bufs = {"colptr": np.empty(self.shape[1] + 1, dtype=np.int64) if "colptr" in need else None,
                    "rowval": np.empty(self.nnz, dtype=np.int32) if "rowval" in need else None,
                    "nzval": np.empty(self.nnz, dtype=np.float32) if "nzval" in need else None}
            self.ctx.check(self.ctx.lib.sclens_hip_counts_download(
                self.ctx.h, self.h, ptr(bufs["colptr"], C.c_int64) if bufs["colptr"] is not None else None,
                ptr(bufs["rowval"], C.c_int32) if bufs["rowval"] is not None else None,
                ptr(bufs["nzval"], C.c_float) if bufs["nzval"] is not None else None))
            self._host.update({k: v for k, v in bufs.items() if v is not None})
        return tuple(self._host.get(k) if want[k] else None for k in ("colptr", "rowval", "nzval"))

    def to_scipy(self) -> sp.csc_matrix:
        cp, rv, nz = self.download()
        X = sp.csc_matrix((nz, rv, cp), shape=self.shape)
        X._sclens_canonical = True
        return X

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.sclens_hip_counts_destroy(self.h)
            self.h = None
            self.ctx._release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pattern:
    """sclens_hip_pattern: the sparse pattern counts + zero candidates, built (host) and uploaded on `ctx`'s stream."""

    def __init__(self, ctx: Context, X: sp.csc_matrix, z1: np.ndarray, z2: np.ndarray):
        self.ctx = ctx
        colptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X.data, dtype=np.float32)
        z1 = np.ascontiguousarray(z1, dtype=np.uint32)
        z2 = np.ascontiguousarray(z2, dtype=np.uint32)
        self.ncand = int(z1.size)
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_pattern_create(ctx.h, X.shape[0], X.shape[1], ptr(colptr, C.c_int64),
                                                    ptr(rowval, C.c_int32), ptr(nzval, C.c_float), self.ncand,
                                                    ptr(z1, C.c_uint32), ptr(z2, C.c_uint32), C.byref(h)))
        self.h = h
        self.ctx._adopt()

    @classmethod
    def drawn(cls, ctx: Context, X: sp.csc_matrix, seed: int) -> "Pattern":
        """counts' CSC in; the zero candidates (R1) are drawn and merged into the union pattern on the device"""
        p = cls.__new__(cls)
        p.ctx = ctx
        if isinstance(X, DeviceCounts):  # in place from the device-resident matrix
            h, nc = C.c_void_p(), C.c_int64(0)
            ctx.check(ctx.lib.sclens_hip_pattern_create_drawn_from_counts(ctx.h, X.h, int(seed) & _M64, C.byref(h), C.byref(nc)))
            p.h, p.ncand = h, int(nc.value)
            ctx._adopt()
            return p
        colptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X.data, dtype=np.float32)
        h, nc = C.c_void_p(), C.c_int64(0)
        ctx.check(ctx.lib.sclens_hip_pattern_create_drawn(ctx.h, X.shape[0], X.shape[1], ptr(colptr, C.c_int64), ptr(rowval, C.c_int32),
                                                          ptr(nzval, C.c_float), int(seed) & _M64, C.byref(h), C.byref(nc)))
        p.h, p.ncand = h, int(nc.value)
        ctx._adopt()
        return p

    def candidates(self):
        z1, z2 = np.empty(self.ncand, dtype=np.uint32), np.empty(self.ncand, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.sclens_hip_pattern_candidates(self.ctx.h, self.h, ptr(z1, C.c_uint32), ptr(z2, C.c_uint32)))
        return z1, z2

    def download(self, which: int, count: int) -> np.ndarray:
        out = np.empty(count, dtype=[np.int64, np.int32, np.int64, np.int64, np.int32, np.int64, np.float32][which])
        self.ctx.check(self.ctx.lib.sclens_hip_pattern_download(self.ctx.h, self.h, int(which), out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.sclens_hip_pattern_destroy(self.h)
            self.h = None
            self.ctx._release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """Device-resident state of one sclens() call (include/sclens_hip.h, part B)."""

    def __init__(self, ctx: Context, X: sp.csc_matrix, z1: Optional[np.ndarray] = None, z2: Optional[np.ndarray] = None):
        """z1/z2 = None: counts only; the candidates are attached later with set_pattern (see Pattern)."""
        self.ctx = ctx
        self.N, self.M = X.shape
        self.n = min(X.shape)
        if isinstance(X, DeviceCounts):  # counts-only session built in place from the device-resident matrix
            if z1 is not None and len(z1):
                raise ValueError("a session from DeviceCounts starts without candidates: attach them with set_pattern")
            self.ncand = 0
            h = C.c_void_p()
            ctx.check(ctx.lib.sclens_hip_session_create_from_counts(ctx.h, X.h, C.byref(h)))
            self.h = h
            self.ctx._adopt()
            return
        colptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X.data, dtype=np.float32)
        z1 = np.ascontiguousarray(z1 if z1 is not None else [], dtype=np.uint32)
        z2 = np.ascontiguousarray(z2 if z2 is not None else [], dtype=np.uint32)
        self.ncand = int(z1.size)
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_session_create(ctx.h, self.N, self.M, ptr(colptr, C.c_int64), ptr(rowval, C.c_int32),
                                                    ptr(nzval, C.c_float), self.ncand, ptr(z1, C.c_uint32),
                                                    ptr(z2, C.c_uint32), C.byref(h)))
        self.h = h
        self.ctx._adopt()

    @classmethod
    def create_sharded(cls, ctx: Context, X_local: sp.csc_matrix, row0: int, N_global: int, z1: np.ndarray, z2: np.ndarray,
                       reducer) -> "Session":
        """Row-sharded session (sclens_hip_session_create_sharded): `X_local` = the cells [row0, row0 + N_local) of an
        N_global x M matrix, z1 = GLOBAL cell indices of the whole candidate list, `reducer` = (function, user pointer) from
        `Shard.reducer` (kept alive by the session)."""
        s = cls.__new__(cls)
        s.ctx = ctx
        s.N, s.M = X_local.shape
        s.n = s.M
        fn, user = reducer if isinstance(reducer, tuple) else (reducer, None)
        s._reducer = (fn, user)
        colptr = np.ascontiguousarray(X_local.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X_local.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X_local.data, dtype=np.float32)
        z1 = np.ascontiguousarray(z1, dtype=np.uint32)
        z2 = np.ascontiguousarray(z2, dtype=np.uint32)
        s.ncand = int(z1.size)
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_session_create_sharded(ctx.h, int(N_global), int(row0), s.N, s.M, ptr(colptr, C.c_int64),
                                                            ptr(rowval, C.c_int32), ptr(nzval, C.c_float), s.ncand,
                                                            ptr(z1, C.c_uint32), ptr(z2, C.c_uint32), fn, user,
                                                            C.byref(h)))
        s.h = h
        s.ctx._adopt()
        return s

    @classmethod
    def create_sharded_drawn(cls, ctx: Context, X_local: sp.csc_matrix, row0: int, N_global: int, nnz_global: int, seed: int,
                             reducer) -> "Session":
        """Row-sharded session whose zero candidates are drawn on the device for the local cells only
        (sclens_hip_session_create_sharded_drawn); `s.ncand` = the local count until `set_candidate_range` is called."""
        s = cls.__new__(cls)
        s.ctx = ctx
        s.N, s.M = X_local.shape
        s.n = s.M
        fn, user = reducer if isinstance(reducer, tuple) else (reducer, None)
        s._reducer = (fn, user)
        colptr = np.ascontiguousarray(X_local.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X_local.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X_local.data, dtype=np.float32)
        h, nc = C.c_void_p(), C.c_int64(0)
        ctx.check(ctx.lib.sclens_hip_session_create_sharded_drawn(ctx.h, int(N_global), int(row0), s.N, s.M, ptr(colptr, C.c_int64),
                                                                  ptr(rowval, C.c_int32), ptr(nzval, C.c_float), int(nnz_global),
                                                                  int(seed) & _M64, fn, user, C.byref(h), C.byref(nc)))
        s.h = h
        s.ctx._adopt()
        s.ncand_local = s.ncand = int(nc.value)
        return s

    @classmethod
    def create_chunked(cls, ctx: Context, N_global: int, M: int, n_chunks: int, nnz_global: int, seed: int) -> "Session":
        """Chunked session (sclens_hip_session_create_chunked): all cells of an N_global x M matrix on this device as `n_chunks` CSC chunks of
        consecutive cells, added one at a time with `chunk_add` and sealed with `chunk_commit`; `seed` = the candidate seed (R1)."""
        s = cls.__new__(cls)
        s.ctx = ctx
        s.N, s.M = int(N_global), int(M)
        s.n = s.M
        s.ncand = 0
        h = C.c_void_p()
        ctx.check(ctx.lib.sclens_hip_session_create_chunked(ctx.h, s.N, s.M, int(n_chunks), int(nnz_global), int(seed) & _M64, C.byref(h)))
        s.h = h
        s.ctx._adopt()
        return s

    def chunk_add(self, which: int, g: int, row0: int, X_chunk: sp.csc_matrix):
        """cells [row0, row0 + rows) of the count matrix (which = 0) or of the null matrix X_r (which = 1), LOCAL row indices"""
        colptr = np.ascontiguousarray(X_chunk.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(X_chunk.indices, dtype=np.int32)
        nzval = np.ascontiguousarray(X_chunk.data, dtype=np.float32)
        self.ctx.check(self.ctx.lib.sclens_hip_session_chunk_add(self.h, int(which), int(g), int(row0), int(X_chunk.shape[0]),
                                                                 ptr(colptr, C.c_int64), ptr(rowval, C.c_int32), ptr(nzval, C.c_float)))

    def chunk_commit(self):
        self.ctx.check(self.ctx.lib.sclens_hip_session_chunk_commit(self.h))

    def null_spectrum_chunked(self) -> np.ndarray:
        Lr = np.empty(self.n, dtype=np.float64)
        self.ctx.check(self.ctx.lib.sclens_hip_session_null_spectrum_chunked(self.h, ptr(Lr, C.c_double)))
        return Lr

    def set_candidate_range(self, cand_off: int, ncand_global: int):
        self.ctx.check(self.ctx.lib.sclens_hip_session_set_candidate_range(self.h, int(cand_off), int(ncand_global)))
        self.ncand = int(ncand_global)

    def set_reduce_to(self, reducer_to):
        fn, user = reducer_to
        self._reducer_to = (fn, user)
        self.ctx.check(self.ctx.lib.sclens_hip_session_set_reduce_to(self.h, fn, user))

    def local_candidates(self):
        n = int(getattr(self, "ncand_local", 0))
        z1, z2 = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.sclens_hip_session_local_candidates(self.h, ptr(z1, C.c_uint32), ptr(z2, C.c_uint32)))
        return z1, z2

    def search_round_seeded(self, seeds, ms, roots, my_slot: int, n_2: int):
        """one round of a row-sharded search: returns (d5, r) of evaluation `my_slot` (None when my_slot < 0)"""
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        mm = np.ascontiguousarray(ms, dtype=np.int64)
        rr = np.ascontiguousarray(roots, dtype=np.int32)
        d5, r = np.empty(5), C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_search_round_seeded(self.h, sd.ctypes.data_as(C.POINTER(C.c_uint64)), ptr(mm, C.c_int64),
                                                                         ptr(rr, C.c_int32), len(sd), int(my_slot), int(n_2),
                                                                         ptr(d5, C.c_double), C.byref(r)))
        return (d5, r.value) if my_slot >= 0 else (None, 0)

    def perturb_round_seeded(self, ts, seeds, ms, roots, my_slot: int, min_pc: int):
        tt = np.ascontiguousarray(ts, dtype=np.int64)
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        mm = np.ascontiguousarray(ms, dtype=np.int64)
        rr = np.ascontiguousarray(roots, dtype=np.int32)
        nL = np.zeros((len(tt), int(min_pc)))
        nc = np.zeros(len(tt), dtype=np.int64)
        self.ctx.check(self.ctx.lib.sclens_hip_session_perturb_round_seeded(self.h, ptr(tt, C.c_int64), sd.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                                          ptr(mm, C.c_int64), ptr(rr, C.c_int32), len(tt), int(my_slot),
                                                                          int(min_pc), ptr(nL, C.c_double), ptr(nc, C.c_int64)))
        return [nL[e, : nc[e]].copy() for e in range(len(tt))], [int(c) for c in nc]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.sclens_hip_session_destroy(self.h)
            self.h = None
            self.ctx._release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shared_buffer(self, what: int, rows: int = 0, k: int = 0, theta0: Optional[np.ndarray] = None):
        """(device pointer, rows, theta0) of Vr2 (what = 1) or of the seed block (what = 2); rows > 0 allocates on a receiver."""
        p, r_out, k_out, ld = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        th = np.zeros(max(int(rows), 128))
        if theta0 is not None:
            th[: len(theta0)] = theta0
        self.ctx.check(self.ctx.lib.sclens_hip_session_shared_buffer(self.h, int(what), int(rows), int(k), ptr(th, C.c_double),
                                                                     C.byref(p), C.byref(r_out), C.byref(k_out), C.byref(ld)))
        self._ldz = ld.value
        return p.value, r_out.value, th[: r_out.value].copy()

    def ldz(self) -> int:
        return (self.n + 31) // 32 * 32

    def set_pattern(self, pat: "Pattern"):
        """Attach the union pattern (counts + zero candidates) built by `Pattern`; the session must be idle."""
        self.ctx.check(self.ctx.lib.sclens_hip_session_set_pattern(self.h, pat.h))
        self.ncand = pat.ncand

    def clone(self, ctx2: Context) -> "Session":
        """Worker session on another context (stream) of the same GPU, sharing the read-only device data."""
        w = Session.__new__(Session)
        w.ctx, w.N, w.M, w.n, w.ncand = ctx2, self.N, self.M, self.n, self.ncand
        h = C.c_void_p()
        ctx2.check(ctx2.lib.sclens_hip_session_clone(ctx2.h, self.h, C.byref(h)))
        w.h = h
        w.ctx._adopt()
        return w

    def spectrum(self, X_r: sp.csc_matrix):
        X_r = _csc_f32(X_r)
        cp = np.ascontiguousarray(X_r.indptr, dtype=np.int64)
        rv = np.ascontiguousarray(X_r.indices, dtype=np.int32)
        nz = np.ascontiguousarray(X_r.data, dtype=np.float32)
        L = np.empty(self.n)
        Lr = np.empty(self.n)
        rec = {"TGC": np.empty(self.N), "mat2_mean": np.empty(self.M), "mat2_std": np.empty(self.M),
               "norm_tgc": np.empty(self.N), "cent_": np.empty(self.M)}
        self.ctx.check(self.ctx.lib.sclens_hip_session_spectrum(
            self.h, ptr(cp, C.c_int64), ptr(rv, C.c_int32), ptr(nz, C.c_float), ptr(L, C.c_double), ptr(Lr, C.c_double),
            ptr(rec["TGC"], C.c_double), ptr(rec["mat2_mean"], C.c_double), ptr(rec["mat2_std"], C.c_double),
            ptr(rec["norm_tgc"], C.c_double), ptr(rec["cent_"], C.c_double)))
        return L, Lr, rec

    def null_spectrum(self, X_r: sp.csc_matrix) -> np.ndarray:
        X_r = _csc_f32(X_r)
        cp = np.ascontiguousarray(X_r.indptr, dtype=np.int64)
        rv = np.ascontiguousarray(X_r.indices, dtype=np.int32)
        nz = np.ascontiguousarray(X_r.data, dtype=np.float32)
        Lr = np.empty(self.n)
        self.ctx.check(self.ctx.lib.sclens_hip_session_null_spectrum(self.h, ptr(cp, C.c_int64), ptr(rv, C.c_int32),
                                                                     ptr(nz, C.c_float), ptr(Lr, C.c_double)))
        return Lr

    def null_spectrum_pattern(self, pat: "Pattern") -> np.ndarray:
        Lr = np.empty(self.n)
        self.ctx.check(self.ctx.lib.sclens_hip_session_null_spectrum_pattern(self.h, pat.h, ptr(Lr, C.c_double)))
        return Lr

    def data_spectrum(self, with_rec_vals: bool = True):
        L = np.empty(self.n)
        if not with_rec_vals:  # centering="median": the reference's rec_vals Dict stays empty (scLENS.jl:697-698)
            self.ctx.check(self.ctx.lib.sclens_hip_session_data_spectrum(self.h, ptr(L, C.c_double), None, None, None, None, None))
            return L, {}
        rec = {"TGC": np.empty(self.N), "mat2_mean": np.empty(self.M), "mat2_std": np.empty(self.M),
               "norm_tgc": np.empty(self.N), "cent_": np.empty(self.M)}
        self.ctx.check(self.ctx.lib.sclens_hip_session_data_spectrum(
            self.h, ptr(L, C.c_double), ptr(rec["TGC"], C.c_double), ptr(rec["mat2_mean"], C.c_double),
            ptr(rec["mat2_std"], C.c_double), ptr(rec["norm_tgc"], C.c_double), ptr(rec["cent_"], C.c_double)))
        return L, rec

    def adopt(self, src: "Session", what: int):
        self.ctx.check(self.ctx.lib.sclens_hip_session_adopt(self.h, src.h, int(what)))

    def refine_eigenvalues(self, idx_lo: int, idx_hi: int) -> np.ndarray:
        rho = np.empty(max(0, idx_hi - idx_lo))
        if rho.size:
            self.ctx.check(self.ctx.lib.sclens_hip_session_refine_eigenvalues(self.h, int(idx_lo), int(idx_hi), ptr(rho, C.c_double)))
        return rho

    def signal_vectors(self, k: int) -> np.ndarray:
        nV = np.empty((self.N, k), dtype=np.float32, order="F")
        self.ctx.check(self.ctx.lib.sclens_hip_session_signal_vectors(self.h, int(k), ptr(nV, C.c_float)))
        return nV

    def binary_basis(self):
        L = np.empty(self.n)
        r = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_binary_basis(self.h, ptr(L, C.c_double), C.byref(r)))
        return L, r.value

    def search_step(self, sample: np.ndarray, n_2: int):
        s = np.ascontiguousarray(sample, dtype=np.uint32)
        d5 = np.empty(5)
        r = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_search_step(self.h, ptr(s, C.c_uint32), s.size, int(n_2),
                                                                   ptr(d5, C.c_double), C.byref(r)))
        return d5, r.value

    def search_step_seeded(self, seed: int, m: int, n_2: int):
        d5 = np.empty(5)
        r = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_search_step_seeded(self.h, int(seed) & _M64, int(m), int(n_2),
                                                                          ptr(d5, C.c_double), C.byref(r)))
        return d5, r.value

    def perturb_seeded(self, t: int, seed: int, m: int, min_pc: int):
        nL = np.empty(min_pc)
        c = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_perturb_seeded(self.h, int(t), int(seed) & _M64, int(m), int(min_pc),
                                                                      ptr(nL, C.c_double), C.byref(c)))
        return nL[: c.value], c.value

    def perturb(self, t: int, sample: np.ndarray, min_pc: int):
        s = np.ascontiguousarray(sample, dtype=np.uint32)
        nL = np.empty(min_pc)
        c = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_perturb(self.h, int(t), ptr(s, C.c_uint32), s.size, int(min_pc),
                                                               ptr(nL, C.c_double), C.byref(c)))
        return nL[: c.value], c.value

    def get_perturbed(self, t: int, ncols: int) -> np.ndarray:
        out = np.empty((self.N, ncols), dtype=np.float32, order="F")
        self.ctx.check(self.ctx.lib.sclens_hip_session_get_perturbed(self.h, int(t), ptr(out, C.c_float)))
        return out

    def set_int(self, name: str, value: int):
        self.ctx.check(self.ctx.lib.sclens_hip_session_set_int(self.h, name.encode(), int(value)))

    def get_int(self, name: str) -> int:
        v = C.c_int64(0)
        self.ctx.check(self.ctx.lib.sclens_hip_session_get_int(self.h, name.encode(), C.byref(v)))
        return v.value

    def slot_ld(self) -> int:
        return int(self.ctx.lib.sclens_hip_session_slot_ld(self.h))

    def export_slot(self, t: int, min_pc: int, dst_dev_ptr: int):
        self.ctx.check(self.ctx.lib.sclens_hip_session_export_slot(self.h, int(t), int(min_pc), C.c_void_p(dst_dev_ptr)))

    def import_slot(self, t: int, min_pc: int, ncols: int, src_dev_ptr: int):
        self.ctx.check(self.ctx.lib.sclens_hip_session_import_slot(self.h, int(t), int(min_pc), int(ncols),
                                                                   C.c_void_p(src_dev_ptr)))

    def robustness(self, k: int, P: int):
        a_b = np.empty((k, P), dtype=np.int32, order="F")
        b = np.empty((k, P * (P - 1) // 2), dtype=np.float64)
        self.ctx.check(self.ctx.lib.sclens_hip_session_robustness(self.h, int(P), ptr(a_b, C.c_int32), ptr(b, C.c_double)))
        return a_b, b

    def gene_basis(self, nL: np.ndarray) -> np.ndarray:
        nL = np.ascontiguousarray(nL, dtype=np.float64)
        out = np.empty((nL.size, self.M), dtype=np.float32)
        self.ctx.check(self.ctx.lib.sclens_hip_session_gene_basis(self.h, ptr(nL, C.c_double), ptr(out, C.c_float)))
        return out


# ----------------------------------------------------------------------------- driver
def _needs_chunks(ctx: Context, X_, streams) -> bool:
    """does the plain session's resident state exceed the device? (cells > genes only: chunks are chunks of cells)"""
    if isinstance(X_, DeviceCounts):
        return False
    N, M = X_.shape
    if N <= M:
        return False
    try:  # (the ABI has no memory-info entry point: the device's size through torch when it is there, else an MI355X's)
        import torch

        total = torch.cuda.get_device_properties(ctx.device).total_memory
    except Exception:
        total = 288 << 30
    n_streams = streams if streams is not None else (3 if min(N, M) < 16000 else 2)
    need = 56 * 2 * int(X_.nnz) + (1 + n_streams) * 4 * N * M + 8 * 2 * int(X_.nnz) + n_streams * (45 << 30)
    return need > 0.85 * total


def cut_with_guard_band(L: np.ndarray, lambda_c: float, guard_band: float, refine: Callable[[int, int], np.ndarray]):
    """`sum(L .> lambda_c)` (scLENS.jl:539, :541, :580) with the fp32 eigenvalues near the cut re-evaluated in float64.

    L ascending (fp32 solver). Every eigenvalue within `guard_band` sqrt(n) eps32 lambda_max of lambda_c is replaced by the
    float64 Rayleigh quotient of ITS eigenvector (`refine(lo, hi)` -> rho for the ascending indices lo..hi-1). The refined
    values of a near-degenerate pair may come out in the other order, so the cut is taken from the top and stays contiguous:
    k = the number of leading indices n-1, n-2, ... whose value exceeds lambda_c, which is what `signal_vectors(k)` returns
    vectors for; eigenvalue q of `nL` is the quotient of vector q (descending index order). Returns (L with the refined
    values, sorted ascending; k; nL; report)."""
    L = np.asarray(L, dtype=np.float64)
    n = len(L)
    band = float(guard_band) * math.sqrt(n) * 5.96e-8 * float(L[-1]) if n else 0.0
    near = np.flatnonzero(np.abs(L - lambda_c) <= band)
    guard = {"band": band, "refined": [], "monotone": True}
    Lw = L
    if guard_band > 0 and near.size and near.size <= 64:
        lo_i, hi_i = int(near[0]), int(near[-1]) + 1
        rho = np.asarray(refine(lo_i, hi_i), dtype=np.float64)
        guard["refined"] = [(int(i), float(L[i]), float(rho[i - lo_i])) for i in range(lo_i, hi_i)]
        Lw = L.copy()
        Lw[lo_i:hi_i] = rho
        guard["monotone"] = bool(np.all(np.diff(Lw[max(lo_i - 1, 0): min(hi_i + 1, n)]) >= 0))
    below = np.flatnonzero(~(Lw > lambda_c))  # strict, as in the reference (Appendix A11)
    k = n - (int(below[-1]) + 1) if below.size else n
    nL = Lw[n - k:][::-1].copy()  # descending index order: value q belongs to eigenvector q of signal_vectors(k)
    return (Lw if guard["monotone"] else np.sort(Lw)), k, nL, guard


def _exchange_ensemble(ses: "Session", shard: Shard, n_perturb: int, min_pc: int, nL_set, ncols):
    """The single gather of the ensemble: all-gather the owned N x min_pc blocks (+ their
    eigenvalues and column counts) over RCCL; every rank then holds all slots (rank 0 scores them). The blocks
    travel between library-owned buffers (`Shard.allgather_dev`)."""
    ctx = ses.ctx
    ld = ses.slot_ld()
    per = (n_perturb + shard.world - 1) // shard.world
    blk = min_pc * ld  # floats per slot
    meta = np.zeros((per, min_pc + 1))
    mine = owned_perturbations(shard.rank, shard.world, n_perturb)
    send = ctx.malloc(4 * per * blk)
    recv = ctx.malloc(4 * shard.world * per * blk)
    try:
        ctx.memset(send, 0, 4 * per * blk)
        for q, t in enumerate(mine):
            ses.export_slot(t, min_pc, send + 4 * q * blk)
            meta[q, 0] = ncols[t]
            meta[q, 1: 1 + len(nL_set[t])] = nL_set[t]
        ctx.sync()
        shard.allgather_dev(ctx, send, recv, per * blk)
        allm = shard.allgather_small(meta)
        for r in range(shard.world):
            for q, t in enumerate(owned_perturbations(r, shard.world, n_perturb)):
                c = int(allm[r, q, 0])
                ncols[t] = c
                nL_set[t] = allm[r, q, 1: 1 + c].copy()
                if r != shard.rank:
                    ses.import_slot(t, min_pc, c, recv + 4 * (r * per + q) * blk)
        ctx.sync()
    finally:
        ctx.free(send)
        ctx.free(recv)


def _extract(inp):
    """df2sparr(inp_df) (scLENS.jl:662, :90-120) for a DataFrame with a leading `cell` column, a scipy sparse
    matrix or a dense array (cells x genes)."""
    cell_id = gene_id = None
    if isinstance(inp, DeviceCounts):
        return inp, np.arange(inp.shape[0]).astype(str), np.arange(inp.shape[1]).astype(str)
    try:
        import pandas as pd

        if isinstance(inp, pd.DataFrame):
            cell_id = inp.iloc[:, 0].astype(str).to_numpy()
            gene_id = np.asarray(inp.columns[1:])
            inp = inp.iloc[:, 1:].to_numpy(dtype=np.float32)
    except ImportError:
        pass
    X = _csc_f32(inp)
    if cell_id is None:
        cell_id = np.arange(X.shape[0]).astype(str)
        gene_id = np.arange(X.shape[1]).astype(str)
    return X, cell_id, gene_id


_LAST_SHAPE = None  # (device, N, M) of the previous sclens() call of this process


def sclens(inp_df, device_="gpu", th=60, p_step=0.001, n_perturb=20, centering="mean", draws: Optional[Draws] = None,
           seed: Optional[int] = None, ctx: Optional[Context] = None, max_search_iters: Optional[int] = None,
           keep_intermediates: bool = False, verbose: bool = False, shard: Optional[Shard] = None,
           partial_eig: bool = True, streams: Optional[int] = None, spread_initial: bool = True,
           guard_band: float = 4.0, keep_warm: bool = False, ensemble_tail: str = "auto",
           chunk_rows: Optional[int] = None) -> Dict[str, object]:
    """scLENS.sclens (scLENS.jl:649-832) on one MI355X.

    Same keyword arguments as the reference. `draws`/`seed` expose the randomness the reference takes from
    Julia's global RNG; `max_search_iters` is a test-only cap of the sparsity search; `guard_band` is the half-width of the
    band around the signal threshold, in units of sqrt(n) eps32 lambda_max, inside which eigenvalues are re-evaluated in
    float64 before the cut (0 switches the refinement off). The arithmetic (the reference's `device_` picks Float32 GPU or Float64
    CPU, scLENS.jl:649) is the context's "precision" option: `ctx.set_option("precision", 0)` = every product on the fp32 matrix cores,
    1 (default) = large products from two fp16 pieces; worker contexts inherit the caller's options. `keep_warm=True` leaves the call's
    device blocks in the library's pool for a next call of the same shape (a loop of calls, bench.py); the default gives them back to
    the driver when the call returns (sclens_hip_trim), so that whatever runs next on the GPU finds the memory free.
    `ensemble_tail`: how the partial eigensolver treats the eigenpairs k .. ceil(1.5 k)-1 of an ensemble member, which the reference
    computes (:776) and consumes only if the matching (:788) picks one: "converged" = to 5e-3 of their eigenvalue; "certified" = not
    at all, and the matching is accepted per member only with a proof that no vector outside the first k could have been picked
    (else that member is solved again, tail included); "auto" = certified from order 16 000. The outputs are the same either way;
    `nL_set[t][k:]` (not a reference output) is NaN in the certified mode for members that were not solved again; the unconverged
    block's Ritz values are returned as `nL_tail_ritz_estimates[t]`.
    """
    if device_ != "gpu":
        raise NotImplementedError("sclens_amd implements the device path only; use the reference for device_='cpu'")
    fallback_centering = False
    if centering not in ("mean", "median"):
        # scLENS.jl:655-657: the reference warns and runs scaled_gdata(norm_l(scaled_gdata(x, "mean")), "cent") on a dense Float32 copy.
        # That is the SAME function of x as the mean branch -- (x - mean) / std per gene, rows scaled to the mean row norm, columns
        # centred -- evaluated in Float32 instead of through the sparse Float64 identities of zscore_with_l2 (oracle.logn_scale_other
        # restates it; tests/test_oracle.py pins the two against each other). The device path is fp32 either way: same warning, mean path.
        print("Warning: The specified centering method is not supported in the current algorithm. scLENS will automatically use mean centering.")
        centering, fallback_centering = "mean", True
    median = centering == "median"
    want_rec = not median and not fallback_centering  # rec_vals is filled by the "mean" branch only (scLENS.jl:677-698)
    ctx = ctx or default_context()
    shard = shard or Shard()
    t_all = time.perf_counter()
    X_, cell_id, gene_id = _extract(inp_df)  # :662
    N, M = X_.shape
    global _LAST_SHAPE
    if _LAST_SHAPE is not None and _LAST_SHAPE != (ctx.device, N, M):
        # the library's memory pool keeps the blocks of the previous call for the next call of the SAME shape; a different shape
        # asks for other size classes, so the idle blocks go back to the driver first instead of piling up beside the new ones
        ctx.trim_pool()
    _LAST_SHAPE = (ctx.device, N, M)
    if streams is None:
        # concurrent decompositions pay while one decomposition cannot fill the GPU (latency-bound column steps / panels /
        # bulge chasing: 4.8 s instead of 7.9 s at 10 000 x 20 000 with three streams). At n >= 16 000 a search step used to be
        # dominated by MFMA-saturating fp32 products (three streams finished 21 evaluations in the time of 21); with the Gram
        # matrix and the search statistic on the fp16 MFMA what remains is the two-stage eigensolver, and TWO streams overlap
        # again: 100 000 x 30 000 in 52.7 s instead of 55.3 s (search 38.3 vs 40.8 s, ensemble 4.0 vs 5.0 s, 2.2 s of worker
        # teardown included; profiles/r02_bench_cfg4_streams2_step1.json)
        streams = 3 if min(N, M) < 16000 else 2
    if draws is None:
        if seed is None:  # every rank must draw the same candidates, null matrix and sample seeds: rank 0's clock decides
            seed = int(shard.bcast_host(np.array([float(time.time_ns() % (2**31))]), 0)[0])
        draws = make_draws_native(X_, int(seed), device_candidates=chunk_rows is not None or _needs_chunks(ctx, X_, streams))
    # A matrix whose resident forms do not fit the device (union pattern 56 B per stored entry, three scaled matrices of 4 N M bytes, the
    # eigensolver's scratch) goes through the CHUNKED session when it has more cells than genes: the same call sequence over chunks of cells
    # (atlas.sclens_chunked; BASELINE configs[4], 1 000 000 x 30 000, on one MI355X). `chunk_rows` forces it (tests).
    if shard.world == 1 and N > M and not median and not isinstance(X_, DeviceCounts) and (chunk_rows is not None or _needs_chunks(ctx, X_, streams)):
        if draws.cand_seed is None or draws.sampler is not None:
            raise ValueError("this matrix only fits the device in chunks of cells (atlas.sclens_chunked), which draws the zero candidates and "
                             "the samples on the device: pass draws made with make_draws_native(..., device_candidates=True) or none")
        from . import atlas

        rows = int(chunk_rows) if chunk_rows is not None else max(1, int((12 << 30) // (4 * M)))  # ~12 GB per scaled block
        cuts = list(range(0, N, rows)) + [N]
        Xr_, Xn_ = X_.tocsr(), _csc_f32(_resolve(draws.X_r)).tocsr()
        res = atlas.sclens_chunked([(a, Xr_[a:b].tocsc()) for a, b in zip(cuts[:-1], cuts[1:])],
                                   [(a, Xn_[a:b].tocsc()) for a, b in zip(cuts[:-1], cuts[1:])], draws, th=th, p_step=p_step,
                                   n_perturb=n_perturb, ctx=ctx, max_search_iters=max_search_iters, verbose=verbose, guard_band=guard_band)
        res.update({"cell_id": cell_id, "gene_id": gene_id})
        if not keep_warm:
            try:
                ctx.release_scratch("everything")
            except Exception:
                pass
            ctx.trim_pool()
        return res
    if ensemble_tail not in ("auto", "certified", "converged"):
        raise ValueError("ensemble_tail must be 'auto', 'certified' or 'converged'")
    tail_free = bool(partial_eig) and (ensemble_tail == "certified" or (ensemble_tail == "auto" and min(N, M) >= 16000))
    phase, phase_peak = {}, {}
    t_ph = time.perf_counter()
    ctx.pool_peak(reset=True)

    def lap(name):  # wall time of the phase that ends here and the pool's largest live footprint on this device during it
        nonlocal t_ph
        now = time.perf_counter()
        phase[name] = phase.get(name, 0.0) + (now - t_ph)
        phase_peak[name] = max(phase_peak.get(name, 0), ctx.pool_peak(reset=True))
        t_ph = now

    # The session starts with the counts only: the data / null / binarised decompositions (:676-721) do not involve the
    # zero candidates. Meanwhile a host thread resolves the candidate draw (:668-673) and builds + uploads the union
    # pattern on an auxiliary context; it is attached before the sparsity search.
    ses = Session(ctx, X_)
    ses.set_int("chefsi", 1 if partial_eig else 0)
    ses.set_int("centering", 1 if median else 0)
    lap("session_create")
    aux_ctx, aux_ctx2 = Context(ctx.device).copy_options_from(ctx), Context(ctx.device).copy_options_from(ctx)
    aux_pool = ThreadPoolExecutor(max_workers=2)

    def build_pattern():
        if draws.z_idx1 is None and draws.cand_seed is not None:  # R1 drawn on the device
            return Pattern.drawn(aux_ctx, X_, draws.cand_seed), None, None
        z1_, z2_ = _resolve(draws.z_idx1), _resolve(draws.z_idx2)
        return Pattern(aux_ctx, X_.to_scipy() if isinstance(X_, DeviceCounts) else X_, z1_, z2_), z1_, z2_

    aux_log = []  # (what, start, end) of the host-side preparation that runs beside the first decompositions

    def build_null_pattern():  # the null matrix's pattern, ready when the null decomposition starts
        t0_ = time.perf_counter() - t_all
        Xr_ = _csc_f32(_resolve(draws.X_r))
        t1_ = time.perf_counter() - t_all
        out_ = Pattern(aux_ctx2, Xr_, [], [])
        aux_log.extend([("null_matrix_draw_wait", round(t0_, 3), round(t1_, 3)), ("null_pattern_build", round(t1_, 3), round(time.perf_counter() - t_all, 3))])
        return out_

    null_future = aux_pool.submit(build_null_pattern)
    pat_future = aux_pool.submit(build_pattern)
    pat = None
    workers, wctx, pool = [ses], [], None
    try:
        # local workers: `streams` sessions on this GPU (own stream + scratch each, shared read-only data), one host
        # thread per session (ctypes releases the GIL), so independent decompositions overlap on the device
        for _ in range(max(1, int(streams)) - 1):
            c2 = Context(ctx.device).copy_options_from(ctx)
            wctx.append(c2)
            workers.append(ses.clone(c2))
        gb0 = sum(w.get_int("gram_bits_used") for w in workers)  # context-lifetime counters: this call's share is the difference
        gs0 = sum(w.get_int("gram_sparse_used") for w in workers)
        W = len(workers)
        pool = ThreadPoolExecutor(max_workers=W) if W > 1 else None
        def run_all(jobs):
            """jobs: list of (worker_index, callable) -> results in job order. Jobs of one worker run sequentially in
            one thread (a session is single-threaded); different workers run concurrently."""
            if pool is None:
                return [f() for _, f in jobs]
            out = [None] * len(jobs)
            by_worker = {}
            for pos, (wk, f) in enumerate(jobs):
                by_worker.setdefault(wk, []).append((pos, f))

            def run_group(grp):
                for pos, f in grp:
                    out[pos] = f()

            futs = [pool.submit(run_group, g) for g in by_worker.values()]
            for f in futs:
                f.result()
            return out

        def guarded(f, where):
            """f() on this rank; with several ranks every rank then learns whether ANY rank failed (Shard.all_ok) BEFORE the next
            collective, and all raise together -- a rank-local error must not leave the peers waiting in a broadcast / gather"""
            if shard.world == 1:
                return f()
            try:
                out = f()
            except BaseException as e:
                shard.all_ok(e, where)  # raises e here and a RuntimeError on the other ranks
                raise
            shard.all_ok(None, where)
            return out

        fp_log = []  # (what, start, end) in seconds since the call began: the jobs of the first phase (res["first_phase_s"])

        def timed(what, f):
            def g(*a_, **k_):
                t0_ = time.perf_counter() - t_all
                try:
                    return f(*a_, **k_)
                finally:
                    fp_log.append((what, round(t0_, 3), round(time.perf_counter() - t_all, 3)))
            return g

        def null_spectrum_job(w_):
            pat_null = timed("null_matrix_wait", null_future.result)()
            return timed("null_spectrum", w_.null_spectrum_pattern)(pat_null)

        # ---- get_sigev (:704) and Vr2 (:717-721): the data, null and binarised matrices are independent decompositions
        spread = shard.world > 1 and spread_initial
        if spread:
            # one decomposition per rank (rank 0 data, rank 1 null, rank 2 -- or 1 -- binarised), each running alone on its
            # GPU instead of three sharing every GPU; spectra go over the wire as host arrays, Vr2 and the seed block of the
            # partial eigensolver as device-to-device broadcasts (RCCL)
            r_null, r_bin = 1, (2 if shard.world >= 3 else 1)
            free = workers[1:] if shard.rank == 0 else workers  # sessions for the null / binarised matrix on this rank
            jobs, w_bin = [], ses
            if shard.rank == 0:
                jobs.append((0, lambda: ("data", ses.data_spectrum(want_rec))))
            if shard.rank == r_null:
                w_null = free[0]
                jobs.append((workers.index(w_null), lambda: ("null", w_null.null_spectrum_pattern(null_future.result()))))
            if shard.rank == r_bin:
                w_bin = free[1] if (shard.rank == r_null and len(free) > 1) else free[0]
                jobs.append((workers.index(w_bin), lambda: ("bin", w_bin.binary_basis())))
            got = guarded(lambda: dict(run_all(jobs)) if jobs else {}, "the first decompositions")
            L, rec_vals = got.get("data", (np.zeros(ses.n), {}))
            L = shard.bcast_host(L, 0)
            Lr = shard.bcast_host(got.get("null", np.zeros(ses.n)), r_null)
            r_vr2 = int(shard.bcast_host(np.array([float(got["bin"][1]) if "bin" in got else 0.0]), r_bin)[0])
        elif W == 1:  # serial order of the reference: null and data spectra, signal vectors, then Vr2
            w_bin = ses
            Lr = ses.null_spectrum_pattern(null_future.result())
            L, rec_vals = ses.data_spectrum(want_rec)
            r_vr2 = None
        elif W == 2:
            # Two streams: data | null first. The binarised matrix (the longest of the three: all of its eigenvectors are wanted)
            # runs afterwards beside signal_vectors, whose second back-transformation of a few vectors is one latency-bound wave
            # (0.5 s at n = 3 * 10^4) that costs nothing next to a decomposition and 0.5 s on its own
            # (profiles/r03_first_phase_cfg4.log). Three other schedules (null -> binarised chained on worker 1, binarised -> null, all three
            # at once on a third session) were measured in rounds 4 and 5 and removed: none shortens this phase by more than 0.1 s and
            # each makes the search that follows 0.8-2 s slower (profiles/r05_schedules_release_stagger_chefsi_skinny.log).
            w_null = w_bin = workers[1]
            (L, rec_vals), Lr = run_all([(0, lambda: timed("data_spectrum", ses.data_spectrum)(want_rec)), (1, lambda: null_spectrum_job(w_null))])
            r_vr2 = -1  # decomposed below, next to the signal vectors
        else:  # the main session keeps the data matrix's reflectors for signal_vectors; workers take the other two
            w_null, w_bin = workers[1], workers[2]
            (L, rec_vals), Lr, (_, r_vr2) = run_all([(0, lambda: ses.data_spectrum(want_rec)), (1, lambda: w_null.null_spectrum_pattern(null_future.result())),
                                                     (2, w_bin.binary_basis)])
        L_mp, _, b_min = _mp_calculation(L, Lr[:-1])  # Lr[1:end-1] (:537, :576)
        lambda_c = _tw(L, L_mp)[0]
        # guard band: eigenvalues within +-4 sqrt(n) eps32 lambda_max of the cut are replaced by float64 Rayleigh quotients
        # before `L .> lambda_c` (:541, :580) is taken (the rank that holds the data matrix refines, all ranks use its values)
        def refine(lo_i, hi_i):
            rho = ses.refine_eigenvalues(lo_i, hi_i) if (not spread or shard.rank == 0) else np.zeros(hi_i - lo_i)
            return shard.bcast_host(rho, 0) if spread else rho

        L, k, nL, guard = cut_with_guard_band(L, lambda_c, guard_band, refine)
        if verbose:
            print(f"(Using hip) number of signal ev: {k}")
        if r_vr2 == -1:
            nV, (_, r_vr2) = run_all([(0, lambda: timed("signal_vectors", ses.signal_vectors)(k)), (1, timed("binary_basis", w_bin.binary_basis))])
        else:
            nV = ses.signal_vectors(k) if (not spread or shard.rank == 0) else None
        if r_vr2 is None:
            _, r_vr2 = ses.binary_basis()
        if spread:
            # seed block of the partial eigensolver: rank 0 -> all (b0 rows of ldz floats + their eigenvalues)
            meta = np.zeros(130)
            if shard.rank == 0:
                ptr0, b0_, th0 = ses.shared_buffer(2)
                meta[0] = b0_
                meta[1: 1 + b0_] = th0
            meta = shard.bcast_host(meta, 0)
            b0_ = int(meta[0])
            if b0_ > 0:
                if shard.rank != 0:
                    ptr0, _, _ = ses.shared_buffer(2, rows=b0_, k=k, theta0=meta[1: 1 + b0_])
                shard.bcast_dev(ctx, ptr0, b0_ * ses.ldz(), 0)
            # Vr2: the rank that decomposed the binarised matrix -> all
            if r_vr2 > 0:
                holder = w_bin if shard.rank == r_bin else ses
                ptr1, _, _ = holder.shared_buffer(1, rows=0 if shard.rank == r_bin else r_vr2)
                shard.bcast_dev(ctx, ptr1, r_vr2 * ses.ldz(), r_bin)
                w_bin = holder
        # the split images of the dense Gram products (12 GB per context at 100 000 x 30 000) are idle from here on: back to the pool,
        # where the union pattern and the search's workspaces find them
        for c_ in [ctx] + wctx:
            c_.release_scratch("gram")
        lap("spectra_signal_vectors_vr2")
        pat, z1, z2 = pat_future.result()
        n_cand = pat.ncand
        ses.set_pattern(pat)
        for w in workers:  # share Vr2 (from w_bin), the seed block of the partial eigensolver and the pattern (from ses)
            if w is not w_bin:
                w.adopt(w_bin, 1)
            if w is not ses:
                w.adopt(ses, 2 | 4)
                w.ncand = ses.ncand
        mpC = mp_check(L_mp)  # :706
        p_th = draws.p_th  # :709-712
        lap("attach_candidates")
        # ---- sparsity search (:715-762); world x W consecutive p_ values are evaluated per round
        n_2 = int(round(r_vr2 / 2))  # :722
        p_list = search_schedule(p_step)

        job_log = []  # (worker, iteration, seconds) of every search evaluation: res["search_job_s"]

        def search_job(wk, my_it):
            def f():
                t_job = time.perf_counter()
                try:
                    return f_inner()
                finally:
                    job_log.append((wk, my_it, round(time.perf_counter() - t_job, 4)))

            def f_inner():
                nnzidx = int(round((1 - p_list[my_it]) * M * N))  # :726
                out = np.full(6, np.nan)
                if n_cand >= nnzidx:
                    if draws.sampler is not None:
                        idx = draws.sampler("search", my_it, n_cand, nnzidx)  # :731
                        d5, _r = workers[wk].search_step(idx, n_2)  # :733-747
                    else:
                        d5, _r = workers[wk].search_step_seeded(sample_seed_for(draws.sample_seed, "search", my_it), nnzidx, n_2)
                    out[:5], out[5] = d5, 1.0
                return out
            return f

        tank = np.zeros((5, 0))
        it = 0
        p_ = None

        try:
            while p_ is None:
                base = it + shard.rank * W
                mine = guarded(lambda: np.stack(run_all([(w, search_job(w, base + w)) for w in range(W)])), "the sparsity search")  # W x 6
                allr = shard.allgather_small(mine).reshape(shard.world * W, 6)
                results = [allr[q, :5] if allr[q, 5] == 1.0 else None for q in range(shard.world * W)]
                tank, used, stopped, p_fin = consume_search_round(tank, results, p_list, it, p_th, p_step, max_search_iters)
                it += used
                if stopped:
                    p_ = p_fin
            trace = [(p_list[q], tank[:, q].copy()) for q in range(tank.shape[1])]
            if verbose:
                print(f"Selected perturb sparisty: {p_}")
            # the eigensolver's scratch (band reduction, chase, inverse iteration, both back-transformations: 30-40 GB per context at
            # 100 000 x 30 000) and the search statistic's images are idle during the ensemble, whose partial eigensolver wants 24 GB of
            # split images per context instead
            for c_ in [ctx] + wctx:
                c_.release_scratch("eigensolver")
                c_.release_scratch("corr")
            lap("sparsity_search")

            # ---- perturbation ensemble (:767-778): member t runs on rank t % world, local worker round-robin
            min_s = k
            min_pc = int(math.ceil(min_s * 1.5))
            m_pert = int(round((1 - p_) * M * N))
            if tail_free:
                for w in workers:
                    w.set_int("chefsi_tail_free", 1)
            nL_set = [None] * n_perturb
            ncols = [0] * n_perturb
            if min_s > 0:
                mine_t = owned_perturbations(shard.rank, shard.world, n_perturb)

                def pert_job(wk, t):
                    def f():
                        if draws.sampler is not None:
                            idx = draws.sampler("perturb", t, n_cand, m_pert)
                            return workers[wk].perturb(t, idx, min_pc)
                        return workers[wk].perturb_seeded(t, sample_seed_for(draws.sample_seed, "perturb", t), m_pert, min_pc)
                    return f

                def my_members():
                    for q0 in range(0, len(mine_t), W):
                        chunk = mine_t[q0: q0 + W]
                        outs = run_all([(w, pert_job(w, t)) for w, t in enumerate(chunk)])
                        for w, t in enumerate(chunk):
                            nL_set[t], ncols[t] = outs[w]
                            if w > 0:  # move the slot from the worker session into the main session (device-to-device)
                                buf = ctx.malloc(4 * min_pc * ses.slot_ld())
                                try:
                                    workers[w].export_slot(t, min_pc, buf)
                                    ses.import_slot(t, min_pc, ncols[t], buf)
                                finally:
                                    ctx.free(buf)

                guarded(my_members, "the perturbation ensemble")
                if shard.world > 1:
                    _exchange_ensemble(ses, shard, n_perturb, min_pc, nL_set, ncols)
            pe_counts = (sum(w.get_int("chefsi_used") for w in workers), sum(w.get_int("chefsi_fallback") for w in workers))
            gram_bits_used = sum(w.get_int("gram_bits_used") for w in workers) - gb0
            gram_sparse_used = sum(w.get_int("gram_sparse_used") for w in workers) - gs0
            lap("perturbation_ensemble")
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            for w in workers[1:]:
                w.close()
            for c2 in wctx:
                c2.close()

        res: Dict[str, object] = {"L": L, "Lr": Lr, "L_mp": L_mp, "λ": lambda_c, "lambda_c": lambda_c, "cell_id": cell_id,
                                  "p_": p_, "p_th": p_th, "n_search": it, "search_trace": trace,
                                  "partial_eig": pe_counts, "guard_band": guard,
                                  "gram_bits_used": gram_bits_used, "gram_sparse_used": gram_sparse_used, "search_job_s": sorted(job_log, key=lambda q: (q[1], q[0])),
                                  "first_phase_s": sorted(fp_log + aux_log, key=lambda q: q[1])}
        if min_s == 0:  # :780-784
            res["wall_s"] = time.perf_counter() - t_all
            return res
        if spread and shard.rank != 0:  # the signal vectors and the scaled data matrix live on rank 0, which scores
            res["wall_s"] = time.perf_counter() - t_all
            return res
        # ---- robustness (:786-807)
        th_ = math.cos(math.radians(th))
        a_b, b_ = ses.robustness(k, n_perturb)
        # The partial eigensolver holds the first k pairs of a member (the signals) to a gap-aware residual target; the tail pairs
        # k .. min_pc-1 sit at the edge of the bulk and are accepted at 5e-3 theta. They only matter if the matching (:788) PICKS one:
        # then that member is solved again with the gap-aware target on the tail as well (chefsi.hip; +0.1 s per member instead of
        # +2.5 s per call for applying it to every member) and the matching is redone -- every vector that enters b_ has met it.
        # `ensemble_tail = "certified"` (the default from order 16 000, where a member's eigensolver applies its operator implicitly and
        # two thirds of its passes over the scaled matrix went into those four or five tail pairs): the tail is not converged at all, and
        # the matching is accepted only where it provably does not depend on it -- signal i correlates with ANY unit vector orthogonal to
        # a member's first k eigenvectors by at most sqrt(1 - sum_{j<k} c_ij^2); where the best of the first k beats that bound the
        # argmax lies among them whatever the tail columns hold (session_robustness, `match_uncertain`). Members without that proof
        # are solved again like the ones above. nL_set[t][k:] are then Ritz estimates (a few per cent), not eigenvalues.
        tail_redo = []
        if partial_eig and pe_counts[0] > 0 and ncols and max(ncols) > k:
            tail_redo = [t for t in range(n_perturb) if ncols[t] > k and np.any(a_b[:, t] >= k)]
            if tail_free:
                tail_redo = sorted(set(tail_redo) | {t for t in range(n_perturb) if ses.get_int(f"match_uncertain:{t}")})
                ses.set_int("chefsi_tail_free", 0)
            if tail_redo:
                ses.set_int("chefsi_tail_gap_milli", 50)
                try:
                    for t in tail_redo:
                        if draws.sampler is not None:
                            nL_set[t], ncols[t] = ses.perturb(t, draws.sampler("perturb", t, n_cand, m_pert), min_pc)
                        else:
                            nL_set[t], ncols[t] = ses.perturb_seeded(t, sample_seed_for(draws.sample_seed, "perturb", t), m_pert, min_pc)
                finally:
                    ses.set_int("chefsi_tail_gap_milli", 0)
                a_b, b_ = ses.robustness(k, n_perturb)
        res["tail_redo"] = tail_redo
        res["ensemble_tail"] = "certified" if tail_free else "converged"
        if tail_free and pe_counts[0] > 0:
            # certified mode: the entries k .. min_pc-1 of a member that was not solved again are Ritz values of an unconverged block
            # (a few per cent low), not eigenvalues -- they leave `nL_set` (NaN there) and are kept under a name that says what they are
            res["nL_tail_ritz_estimates"] = [np.asarray(v, dtype=np.float64)[k:].copy() for v in nL_set]
            for t in range(n_perturb):
                if t not in tail_redo and nL_set[t] is not None and len(nL_set[t]) > k:
                    nL_set[t] = np.asarray(nL_set[t], dtype=np.float64).copy()
                    nL_set[t][k:] = np.nan
        m_score, sd_score = _robust_scores(b_)
        rob_score = m_score
        sig_id = np.flatnonzero(rob_score > th_)
        # ---- reconstruction (:809-830)
        Xout0 = nV * np.sqrt(nL)[None, :].astype(np.float32)
        Xout1 = nV[:, sig_id] * np.sqrt(nL[sig_id])[None, :].astype(np.float32)
        gmat = ses.gene_basis(nL)
        res.update({"pca": Xout0, "pca_n1": Xout1, "sig_id": sig_id,
                    "robustness_scores": {"b_": b_, "rob_score": rob_score, "m_scores": m_score, "sd_scores": sd_score,
                                          "a_b": a_b},
                    "signal_evec": nV, "signal_ev": nL, "gene_id": gene_id, "gene_basis": gmat, "pass": mpC["pass"],
                    "ks_static": mpC["ks_static"], "rec_vals": rec_vals, "nL_set": nL_set, "min_pc": min_pc})
        if keep_intermediates:
            res["nV_set"] = [ses.get_perturbed(t, ncols[t]) for t in range(n_perturb)]
        lap("robustness_gene_basis")
        res["phase_s"] = {k_: round(v_, 4) for k_, v_ in phase.items()}
        res["phase_peak_GB"] = {k_: round(v_ / 1e9, 1) for k_, v_ in phase_peak.items()}
        res["wall_s"] = time.perf_counter() - t_all
        return res
    finally:
        try:
            if pat is None:  # failed before the pattern was attached: let the builder finish, then drop its result
                pat = pat_future.result()[0]
        except Exception:
            pat = None
        try:
            null_future.result().close()
        except Exception:
            pass
        aux_pool.shutdown(wait=True)
        if pool is not None:
            pool.shutdown(wait=True)
        for w in workers[1:]:  # an error in the first phase leaves the worker sessions open (closing twice is harmless)
            w.close()
        for c2 in wctx:
            c2.close()
        ses.close()
        if pat is not None:
            pat.close()
        aux_ctx.close()
        aux_ctx2.close()
        if not keep_warm:
            # the caller's context outlives the call: its grow-only workspaces (vector blocks, partial-eigensolver images, re-grown
            # eigensolver scratch) go back to the pool first, THEN the pool's idle blocks go back to the driver
            try:
                ctx.release_scratch("everything")
            except Exception:
                pass
            ctx.trim_pool()
