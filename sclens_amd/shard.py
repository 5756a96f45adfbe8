"""Multi-GPU sharding of one sclens() call: one process per GPU; the few exchanges the path needs are RCCL
calls made inside libsclens_hip.so (csrc/comm.hip), torch.distributed only launches the ranks and ships the unique id.

  * sparsity search (scLENS.jl:725-761): sequential early-exit loop, but each p_ evaluation is independent given
    its sample -> evaluate `world` consecutive p_ values per round, all-gather the five numbers each produces,
    consume them in order with the reference's stop rule (same decision sequence as the serial loop).
  * perturbation ensemble (scLENS.jl:771-778): member t runs on rank t % world; one gather of the N x min_pc
    eigenvector blocks to rank 0 at the end, which then scores robustness.
  * first phase (scLENS.jl:704, :717-721): with 2 or more ranks the data | null | binarised decompositions run on different
    ranks (api.sclens); Vr2 and the spectra travel by one broadcast each.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


class Shard:
    """rank / world + the exchanges of a multi-GPU sclens() call. `world == 1` needs nothing.

    Two transports, the same control flow on both:
      * `comm` (an `_lib.Comm`): the library's own RCCL communicator (csrc/comm.hip). Every exchange -- the ensemble gather,
        the Vr2 / seed-block broadcasts, the all-reduces of a row-sharded session, and the few doubles of the control flow --
        is a library call on library-owned buffers and the library's stream: no host-framework tensor, no staging copy.
        This is what one process per GPU uses (`Shard.create(..., backend="nccl")`).
      * no `comm`: `torch.distributed` with the gloo backend, device buffers staged through host memory. The transport of
        the CPU tests (world 2) and of several ranks sharing one GPU; never used when a communicator exists.
    torch.distributed itself is only the launcher and the channel that ships the 128-byte RCCL unique id.
    """

    def __init__(self, rank: int = 0, world: int = 1, comm=None):
        self.rank, self.world, self.comm = int(rank), int(world), comm

    @classmethod
    def create(cls, ctx, rank: int, world: int, backend: str = "nccl", force_comm: bool = False) -> "Shard":
        """backend "nccl": build the library communicator on `ctx` (the unique id travels through the already initialised
        torch.distributed group); "gloo": host-staged test transport. `force_comm`: a communicator even for one rank
        (tests of the RCCL calls on a one-GPU box)."""
        if backend != "nccl" or (world == 1 and not force_comm):
            return cls(rank, world, None)
        from ._lib import Comm

        def ship(uid):
            """rank 0's unique id to every rank; rank 0 ships an Exception instead when it could not make one (the other ranks would
            wait in this broadcast for ever otherwise) and every rank raises"""
            if world == 1:
                if isinstance(uid, BaseException):
                    raise uid
                return uid
            import torch.distributed as dist

            box = [uid if not isinstance(uid, BaseException) else ("__error__", repr(uid))]
            dist.broadcast_object_list(box, src=0)
            if isinstance(uid, BaseException):
                raise uid
            if isinstance(box[0], tuple) and box[0] and box[0][0] == "__error__":
                raise RuntimeError(f"rank 0 could not create the RCCL unique id: {box[0][1]}")
            return box[0]

        return cls(rank, world, Comm(ctx, rank, world, ship))

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def describe(self) -> dict:
        """what bench.py prints: the rank count RCCL itself reports for the library's communicator"""
        if self.comm is None:
            return {"transport": "torch.distributed/gloo (host-staged)" if self.world > 1 else "none", "world": self.world}
        return {"transport": "rccl (library communicator)", "rccl_ranks": self.comm.world, "rccl_version": self.comm.rccl_version,
                **self.comm.stats()}

    def selfcheck(self, ctx) -> dict:
        """Start-up test of a multi-rank job: a 4-element all-reduce with a known answer on a library buffer (raises if
        wrong) and a broadcast from the last rank."""
        out = self.describe()
        if self.world == 1 and self.comm is None:
            return out
        x = (np.arange(4, dtype=np.float64) + float(self.rank))
        buf = ctx.malloc(x.nbytes)
        try:
            ctx.h2d(buf, x)
            self.allreduce_dev(ctx, buf, 4, 0)
            got = np.empty(4)
            ctx.d2h(got, buf)
            want = self.world * np.arange(4) + self.world * (self.world - 1) / 2.0
            if not np.array_equal(got, want):
                raise RuntimeError(f"all-reduce self-check failed on rank {self.rank}: got {got} instead of {want}")
            y = np.full(4, float(self.rank), dtype=np.float32)
            ctx.h2d(buf, y)
            self.bcast_dev(ctx, buf, 4, self.world - 1)
            ctx.d2h(y, buf)
            if not np.all(y == float(self.world - 1)):
                raise RuntimeError(f"broadcast self-check failed on rank {self.rank}: got {y}")
        finally:
            ctx.free(buf)
        out["selfcheck"] = "ok"
        return out

    # -- equally sized blocks of library device memory: every rank contributes `count` floats, receives world * count
    def allgather_dev(self, ctx, send_ptr: int, recv_ptr: int, count_f32: int):
        """recv[r * count : (r + 1) * count] = rank r's send buffer (both are library allocations on this rank's GPU):
        THE gather of the perturbation ensemble"""
        if self.comm is not None:
            ctx.sync()  # the blocks were written on ctx's stream; the collective runs on the communicator's
            self.comm.allgather(send_ptr, recv_ptr, 4 * int(count_f32))
            return
        if self.world == 1:
            ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, recv_ptr, send_ptr, 4 * int(count_f32), 3))
            return
        import torch
        import torch.distributed as dist

        h = np.empty(int(count_f32), dtype=np.float32)
        ctx.d2h(h, send_ptr)
        outs = [torch.empty(int(count_f32), dtype=torch.float32) for _ in range(self.world)]
        dist.all_gather(outs, torch.from_numpy(h))
        ctx.h2d(recv_ptr, torch.cat(outs).numpy())

    # -- small host arrays (search statistics): fixed-shape float64 all-gather
    def allgather_small(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.comm is not None:
            return self.comm.allgather_host(arr)
        if self.world == 1:
            return arr[None]
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(arr.copy())
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    # -- in-place sum of a raw device buffer over the ranks (row-sharded sessions)
    def allreduce_dev(self, ctx, dev_ptr: int, count: int, dtype: int):
        """dtype 0 = float64, 1 = float32"""
        if count == 0:
            return
        if self.comm is not None:
            ctx.sync()
            self.comm.allreduce(dev_ptr, count, dtype)
            return
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist

        h = np.empty(int(count), dtype=np.float64 if dtype == 0 else np.float32)
        ctx.d2h(h, dev_ptr)
        t = torch.from_numpy(h)
        dist.all_reduce(t)
        ctx.h2d(dev_ptr, h)

    def reducer(self, ctx):
        """the `sclens_hip_allreduce_fn` (+ user pointer) of a row-sharded session: with a communicator the library's own
        entry point (no host callback in the data path), otherwise a ctypes callback into `allreduce_dev`"""
        if self.comm is not None:
            return self.comm.reducer()
        import traceback

        from . import _lib

        def cb(_user, dev_ptr, count, dtype):
            try:
                self.allreduce_dev(ctx, dev_ptr, count, dtype)
                return 0
            except Exception:  # never let an exception cross the C ABI
                traceback.print_exc()
                return 1

        return _lib.ALLREDUCE_FN(cb), None

    def reducer_to(self, ctx):
        """the `sclens_hip_reduce_fn` (+ user pointer) of a row-sharded session: the sum of a buffer onto ONE rank. With a
        communicator ncclReduce inside the library; otherwise a host callback that all-reduces (every rank then holds the sum,
        which is a valid implementation: non-root buffers are simply unspecified)"""
        if self.comm is not None:
            return self.comm.reducer_to()
        import traceback

        from . import _lib

        def cb(_user, dev_ptr, count, dtype, _root):
            try:
                self.allreduce_dev(ctx, dev_ptr, count, dtype)
                return 0
            except Exception:
                traceback.print_exc()
                return 1

        return _lib.REDUCE_FN(cb), None

    # -- one-to-all copies for the spread initial phase (api.sclens, world > 1)
    def bcast_host(self, arr: np.ndarray, src: int) -> np.ndarray:
        """float64 host array of the same shape on every rank; returns rank `src`'s content"""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.comm is not None:
            return self.comm.broadcast_host(arr, src)
        if self.world == 1:
            return arr
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(arr.copy())
        dist.broadcast(t, src=src)
        return t.numpy()

    def bcast_dev(self, ctx, dev_ptr: int, count_f32: int, src: int):
        """`count_f32` floats at a raw device pointer (allocated on every rank), from rank `src` to all, in place"""
        if count_f32 == 0:
            return
        if self.comm is not None:
            ctx.sync()
            self.comm.broadcast(dev_ptr, 4 * int(count_f32), src)
            return
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist

        h = np.empty(int(count_f32), dtype=np.float32)
        if self.rank == src:
            ctx.d2h(h, dev_ptr)
        t = torch.from_numpy(h)
        dist.broadcast(t, src=src)
        if self.rank != src:
            ctx.h2d(dev_ptr, h)

    def agree(self, arr: np.ndarray) -> np.ndarray:
        """rank 0's copy of a small host array on every rank (decisions must not diverge by a rounding bit)"""
        return self.allgather_small(arr)[0]

    def all_ok(self, exc=None, where: str = ""):
        """Every rank calls this at the end of a phase whose next step is a collective, with the exception the phase raised on THIS
        rank (or None). One small all-gather of a status word; if any rank failed, EVERY rank raises -- the failing ones their own
        exception, the others a RuntimeError naming the ranks -- so that no rank walks into a collective its peers will never join
        (a rank-local error used to leave the others blocked in the next all-gather / reduce for ever)."""
        if self.world == 1:
            if exc is not None:
                raise exc
            return
        flags = self.allgather_small(np.array([0.0 if exc is None else 1.0]))[:, 0]
        if exc is not None:
            raise exc
        bad = [int(r) for r in np.flatnonzero(flags != 0.0)]
        if bad:
            raise RuntimeError(f"sclens: rank(s) {bad} failed{(' in ' + where) if where else ''}; rank {self.rank} stops with them")

    def barrier(self):
        if self.comm is not None:
            self.comm.allgather_host(np.zeros(1))
        elif self.world > 1:
            import torch.distributed as dist

            dist.barrier()


def owner_of_perturbation(t: int, world: int) -> int:
    return t % world


def owned_perturbations(rank: int, world: int, n_perturb: int) -> List[int]:
    return [t for t in range(n_perturb) if owner_of_perturbation(t, world) == rank]


class SearchSchedule:
    """p_ of iteration `it`, produced by the same repeated `p_ -= p_step` as the reference (Appendix A21), extended
    on demand: the loop runs until p_ < 0.9 for ANY p_step (about 0.099 / p_step iterations) and a speculative round may
    look `world x streams` iterations past the stop."""

    def __init__(self, p_step: float):
        self.p_step = float(p_step)
        self._p = [0.999]

    def __getitem__(self, it: int) -> float:
        while len(self._p) <= it:
            self._p.append(self._p[-1] - self.p_step)
        return self._p[it]


def search_schedule(p_step: float) -> SearchSchedule:
    return SearchSchedule(p_step)


def consume_search_round(tank: np.ndarray, results: Sequence[np.ndarray], p_list: Sequence[float], it0: int,
                         p_th: float, p_step: float, max_search_iters=None):
    """Apply the stop rule of scLENS.jl:747-760 to the results of iterations it0, it0+1, ... in order.
    results[i] = d5 of iteration it0+i, or None if that iteration hit the early exit of :727-730.
    Returns (tank, n_consumed, stopped, p_final_or_None)."""
    for i, d5 in enumerate(results):
        it = it0 + i
        p_ = p_list[it]
        if d5 is None:  # fewer candidates than requested (:727-730)
            return tank, i, True, p_ + p_step
        tank = np.hstack([tank, np.asarray(d5, dtype=np.float64)[:, None]])
        ppj = tank[1, :] if tank.shape[1] < 5 else tank[1, -5:]
        if (np.sum(ppj < p_th) > 4) or (p_ < 0.9) or (max_search_iters is not None and it + 1 >= max_search_iters):
            return tank, i + 1, True, p_ + 4 * p_step
    return tank, len(results), False, None
