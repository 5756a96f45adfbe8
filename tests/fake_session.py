"""A float64 NumPy stand-in for the device session, for CPU tests of the HOST control flow of `api.sclens` (threads over
worker sessions, speculative search rounds, slot traffic of the ensemble, guard band, scoring, result assembly).

Test infrastructure only: every operation is the oracle's (oracle/sclens_oracle.py), cut at the boundaries of the session
entry points of include/sclens_hip.h, so `api.sclens` run on these fakes must reproduce `oracle.sclens` on the same draws --
any difference is a defect of the host logic, not of a kernel. Nothing here is a fallback of the product: the fakes are
installed by monkeypatching inside a test and exist only under tests/.
"""
import threading

import numpy as np
import scipy.sparse as sp

from oracle import sclens_oracle as O
from sclens_amd import api


class FakeContext:
    """`malloc` hands out addresses of ONE fake address space shared by all contexts (device pointers are valid on every context
    of a GPU, and sclens() does pointer arithmetic on them); `mem` maps an address to the object stored there"""
    _lock = threading.Lock()
    live = 0  # open contexts (a test asserts that sclens() closes what it opens)
    mem = {}
    _next = 1
    SPAN = 1 << 40

    def __init__(self, device=0):
        self.device = int(device)
        self.closed = False
        with FakeContext._lock:
            FakeContext.live += 1

    def trim_pool(self):
        FakeContext.trims = getattr(FakeContext, "trims", 0) + 1

    def release_scratch(self, family):
        pass

    def pool_peak(self, reset=False):
        return 0

    def copy_options_from(self, other):  # worker contexts inherit the caller's options (sclens_hip_copy_options)
        self.options_from = other
        return self

    def malloc(self, nbytes):
        with FakeContext._lock:
            base = FakeContext._next * FakeContext.SPAN
            FakeContext._next += 1
        return base

    def free(self, p):
        with FakeContext._lock:
            for a in [a for a in FakeContext.mem if p <= a < p + FakeContext.SPAN]:
                del FakeContext.mem[a]

    def memset(self, p, value, nbytes):
        pass

    def sync(self):
        pass

    def close(self):
        if not self.closed:
            self.closed = True
            with FakeContext._lock:
                FakeContext.live -= 1


class FakePattern:
    live = 0

    def __init__(self, ctx, X, z1, z2):
        self.ctx, self.X = ctx, X
        self.z1 = np.asarray(z1, dtype=np.uint32)
        self.z2 = np.asarray(z2, dtype=np.uint32)
        self.ncand = int(self.z1.size)
        self.closed = False
        FakePattern.live += 1

    def close(self):
        if not self.closed:
            self.closed = True
            FakePattern.live -= 1


class _Shared:
    """what clones of one session share: the counts and everything derived from them once"""

    def __init__(self, X):
        self.X = O._as_csc_f32(X)
        self.N, self.M = self.X.shape
        coo = self.X.tocoo()
        order = np.lexsort((coo.row, coo.col))
        self.nz_row, self.nz_col, self.nz_val = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order]
        self.lock = threading.Lock()
        self.calls = []  # (thread id, session id, what): the tests look at who ran what
        FakeSession.last_calls = self.calls


class FakeSession:
    live = 0
    _ids = 0

    def __init__(self, ctx, X, z1=None, z2=None, _shared=None):
        self.ctx = ctx
        self.sh = _shared or _Shared(X)
        self.N, self.M = self.sh.N, self.sh.M
        self.n = min(self.N, self.M)
        self.ncand = 0
        self.median = False
        self.z1 = self.z2 = None
        self._vr2, self._vr2_at = None, None
        self.slots = {}
        self.closed = False
        self.busy = threading.Lock()  # a session is single-threaded: overlapping calls are a host bug
        FakeSession._ids += 1
        self.id = FakeSession._ids
        FakeSession.live += 1
        if z1 is not None and len(z1):
            self.z1, self.z2, self.ncand = np.asarray(z1), np.asarray(z2), len(z1)

    # ---- bookkeeping
    def _enter(self, what):
        assert not self.closed, f"{what} on a closed session"
        assert self.busy.acquire(blocking=False), f"session {self.id}: {what} while another call is running on it"
        with self.sh.lock:
            self.sh.calls.append((threading.get_ident(), self.id, what))

    def _ls(self, Y):
        return O.logn_scale_median(Y) if self.median else O.logn_scale(Y)

    def set_int(self, name, value):
        if name == "centering":
            self.median = bool(value)

    def get_int(self, name):
        return 0

    def clone(self, ctx2):
        w = FakeSession(ctx2, None, _shared=self.sh)
        w.median = self.median
        return w

    def close(self):
        if not self.closed:
            self.closed = True
            FakeSession.live -= 1

    # ---- first phase
    def data_spectrum(self, with_rec_vals=True):
        self._enter("data_spectrum")
        try:
            if self.median:
                self.scaled, rec = self._ls(O.pre_scale(self.sh.X)), {}
            else:
                self.scaled, rec = O.scale_main(self.sh.X)
            self.L, self.V = O.get_eigen(O.wishart_matrix(self.scaled, 2 if self.N > self.M else 1))
            return self.L.copy(), rec
        finally:
            self.busy.release()

    def null_spectrum_pattern(self, pat):
        self._enter("null_spectrum")
        try:
            Xr = self._ls(O.pre_scale(O._as_csc_f32(pat.X)))
            return O.get_eigen(O.wishart_matrix(Xr, 2 if self.N > self.M else 1))[0]
        finally:
            self.busy.release()

    def refine_eigenvalues(self, lo, hi):
        self._enter("refine")
        try:
            return self.L[lo:hi].copy()
        finally:
            self.busy.release()

    def signal_vectors(self, k):
        self._enter("signal_vectors")
        try:
            order = np.argsort(-self.L, kind="stable")[:k]
            nL, nVs = self.L[order], self.V[:, order]
            if self.N > self.M:
                nVs = O._normalize_cols(self.scaled @ (nVs * np.sqrt(1.0 / nL)[None, :]))
            self.nV = nVs
            return nVs.astype(np.float32)
        finally:
            self.busy.release()

    def binary_basis(self):
        self._enter("binary_basis")
        try:
            s = self.sh
            binary = sp.csc_matrix((np.ones_like(s.nz_val), (s.nz_row, s.nz_col)), shape=(s.N, s.M), dtype=np.float32)
            sb = self._ls(O.pre_scale(binary))
            self.Vr2 = O.get_eigvec(sb.T if s.N > s.M else sb, O.NULL_DROP)[1]
            return np.zeros(self.n), int(self.Vr2.shape[1])
        finally:
            self.busy.release()

    def set_pattern(self, pat):
        self.z1, self.z2, self.ncand = pat.z1, pat.z2, pat.ncand

    def adopt(self, src, what):
        if what & 1:
            self.Vr2 = src.Vr2
        if what & 4:
            self.z1, self.z2, self.ncand = src.z1, src.z2, src.ncand

    # ---- search and ensemble
    def search_step(self, idx, n_2):
        self._enter("search_step")
        try:
            s = self.sh
            pert = O._with_ones(s.N, s.M, s.nz_row, s.nz_col, s.nz_val, self.z1, self.z2, np.asarray(idx, dtype=np.int64), binary=True)
            sp_ = self._ls(O.pre_scale(pert))
            nV_2 = O.get_eigvec(sp_.T if s.N > s.M else sp_, O.NULL_DROP)[1]
            Cm = O.corr_mat(self.Vr2, nV_2[:, nV_2.shape[1] - n_2 - 1:])
            return np.sort(np.nanmax(np.abs(Cm), axis=0))[:5].copy(), int(nV_2.shape[1])
        finally:
            self.busy.release()

    def search_step_seeded(self, seed, m, n_2):
        return self.search_step(api.sample_indices(self.ncand, m, seed), n_2)

    def perturb(self, t, idx, min_pc):
        self._enter("perturb")
        try:
            s = self.sh
            tmp = O._with_ones(s.N, s.M, s.nz_row, s.nz_col, s.nz_val, self.z1, self.z2, np.asarray(idx, dtype=np.int64), binary=False)
            tL, tV = O.get_eigvec(self._ls(O.pre_scale(tmp)), O.NULL_DROP)
            c = min(min_pc, tV.shape[1])
            self.slots[t] = tV[:, :c]
            return tL[:c].copy(), c
        finally:
            self.busy.release()

    def perturb_seeded(self, t, seed, m, min_pc):
        return self.perturb(t, api.sample_indices(self.ncand, m, seed), min_pc)

    def slot_ld(self):
        return self.N

    def export_slot(self, t, min_pc, dst):
        FakeContext.mem[dst] = self.slots[t]

    def import_slot(self, t, min_pc, ncols, src):
        block = FakeContext.mem[src]
        assert block.shape[1] == ncols
        self.slots[t] = block

    # ---- what the spread first phase of a multi-rank call moves between ranks (api.sclens, world > 1)
    def ldz(self):
        return self.n

    def shared_buffer(self, what, rows=0, k=0, theta0=None):
        if what == 2:  # no seed block: the ensemble of the fake always runs the full solver
            return 0, 0, np.zeros(0)
        key = self.ctx.malloc(0)
        if rows == 0:  # the rank that decomposed the binarised matrix
            FakeContext.mem[key] = self.Vr2
            return key, int(self.Vr2.shape[1]), np.zeros(0)
        self._vr2_at = key  # a receiver: the broadcast lands at this address
        return key, int(rows), np.zeros(0)

    @property
    def Vr2(self):
        if self._vr2 is None and self._vr2_at is not None:
            self._vr2 = FakeContext.mem[self._vr2_at]
        return self._vr2

    @Vr2.setter
    def Vr2(self, v):
        self._vr2 = v

    def get_perturbed(self, t, ncols):
        return self.slots[t][:, :ncols].astype(np.float32)

    def robustness(self, k, P):
        self._enter("robustness")
        try:
            sets = [self.slots[t] for t in range(P)]
            a_b = np.stack([np.argmax(np.abs(self.nV.T @ j), axis=1) for j in sets], axis=1)
            sub = [sets[s][:, a_b[:, s]] for s in range(P)]
            b = [np.max(np.abs(sub[i].T @ sub[j]), axis=1) for i in range(P) for j in range(i + 1, P)]
            return a_b.astype(np.int32), (np.stack(b, axis=1) if b else np.zeros((k, 0)))
        finally:
            self.busy.release()

    def gene_basis(self, nL):
        nL = np.asarray(nL, dtype=np.float64)
        return (((1.0 / np.sqrt(nL))[:, None] * self.nV.T) @ self.scaled / np.sqrt(self.M)).astype(np.float32)


def install(monkeypatch):
    """route api.sclens onto the fakes; returns the main context to pass as `ctx`"""
    FakeContext.live = FakePattern.live = FakeSession.live = 0
    monkeypatch.setattr(api, "Context", FakeContext)
    monkeypatch.setattr(api, "Session", FakeSession)
    monkeypatch.setattr(api, "Pattern", FakePattern)
    FakeContext.mem.clear()
    return FakeContext(0)


# ---- row-sharded sessions (sclens_amd/atlas.py): every rank holds a block of cells ---------------------------------------------
class FakeShardedSession(FakeSession):
    """Stand-in for a row-sharded session. The ranks are threads of one process: each registers its block of cells, the whole
    matrix is assembled from the blocks, and every rank evaluates the oracle's operation on the WHOLE matrix (what the partial
    sums + reductions of the library amount to), handing back its own cells of every cell-side result. What the tests exercise
    is atlas.py: rounds, roots, stop rule, gathers."""
    registry = {}

    @classmethod
    def create_sharded(cls, ctx, X_local, row0, N_global, z1, z2, reducer):
        s = cls.__new__(cls)
        FakeSession.__init__(s, ctx, None, _shared=_Pending())
        s.row0, s.N_local, s.N_global = int(row0), X_local.shape[0], int(N_global)
        s.z1, s.z2, s.ncand = np.asarray(z1), np.asarray(z2), len(z1)
        s.reduce_to = None
        with FakeContext._lock:
            cls.registry.setdefault("X", {})[s.row0] = X_local.tocsr()
        return s

    def _assemble(self, key):
        import time

        t0 = time.time()
        while True:
            with FakeContext._lock:
                blocks = dict(self.registry.get(key, {}))
            if sum(b.shape[0] for b in blocks.values()) == self.N_global:
                return sp.vstack([blocks[r] for r in sorted(blocks)]).tocsc()
            assert time.time() - t0 < 60, "the other ranks never registered their cells"
            time.sleep(0.002)

    def _full(self):
        if isinstance(self.sh, _Pending):
            self.sh = _Shared(self._assemble("X"))
            self.N, self.M, self.n = self.sh.N, self.sh.M, min(self.sh.N, self.sh.M)
        return self.sh

    def _rows(self, A):
        return A[self.row0: self.row0 + self.N_local]

    def set_reduce_to(self, reducer_to):
        self.reduce_to = reducer_to

    def null_spectrum(self, Xr_local):
        with FakeContext._lock:
            self.registry.setdefault("Xr", {})[self.row0] = sp.csr_matrix(Xr_local)
        self._full()

        class P:
            X = self._assemble("Xr")

        return self.null_spectrum_pattern(P)

    def data_spectrum(self, with_rec_vals=True):
        self._full()
        L, rec = FakeSession.data_spectrum(self, with_rec_vals)
        rec = dict(rec)
        for key in ("TGC", "norm_tgc"):  # per-cell vectors: this rank's cells
            rec[key] = self._rows(np.ravel(rec[key]))
        return L, rec

    def signal_vectors(self, k):
        return self._rows(FakeSession.signal_vectors(self, k))

    def binary_basis(self):
        self._full()
        return FakeSession.binary_basis(self)

    def search_round_seeded(self, seeds, ms, roots, my_slot, n_2):
        assert len(seeds) == len(ms) == len(roots) and my_slot < len(seeds)
        if my_slot < 0:
            return None, 0
        return self.search_step(api.sample_indices(self.ncand, ms[my_slot], seeds[my_slot]), n_2)

    def perturb_round_seeded(self, ts, seeds, ms, roots, my_slot, min_pc):
        nl, nc = [], []
        for e, t in enumerate(ts):  # every rank ends up with every member's vectors (its own cells of them in the library)
            a, c = self.perturb(t, api.sample_indices(self.ncand, ms[e], seeds[e]), min_pc)
            nl.append(a)
            nc.append(c)
        return nl, nc


class _Pending:
    N = M = 0
