"""Multi-GPU sharding of one sclens() call: one process per GPU, torch.distributed (RCCL on ROCm) for the few
exchanges the path needs.

  * sparsity search (scLENS.jl:725-761): sequential early-exit loop, but each p_ evaluation is independent given
    its sample -> evaluate `world` consecutive p_ values per round, all-gather the five numbers each produces,
    consume them in order with the reference's stop rule (same decision sequence as the serial loop).
  * perturbation ensemble (scLENS.jl:771-778): member t runs on rank t % world; one gather of the N x min_pc
    eigenvector blocks to rank 0 at the end, which then scores robustness.
The data/null/binary decompositions are replicated on every rank (3 of ~3+S+P; noted in DESIGN.md).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def raw_device_tensor(dev_ptr: int, count: int, typestr: str, device):
    """zero-copy torch view of `count` elements of library-owned device memory (`__cuda_array_interface__`)"""
    import torch

    class _Raw:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(dev_ptr), False), "version": 3}

    return torch.as_tensor(_Raw(), device=device)


class Shard:
    """rank/world + the collectives. `world == 1` needs no torch.

    Device buffers of the library reach RCCL in one of three ways (`staging`):
      "host"   (default) library buffer -> host (the library's own copy) -> torch tensor on the device (torch's own copy) ->
               RCCL -> back the same way. No pointer ever crosses between the two HIP runtime instances of the process
               (the PyTorch-ROCm wheel bundles its own libamdhip64 next to the system one libsclens_hip.so links), so it is
               correct by construction; the volumes of this path are small against its wall-clock.
      "device" a device-to-device copy by the library into a torch-allocated staging tensor (saves the two PCIe hops;
               the copy uses a pointer of the other runtime instance). Selected by `selfcheck()` when it proves that
               such copies round-trip bit-exactly on every rank.
      "zero_copy" RCCL directly on a view of the library's buffer (never selected automatically).
    """

    def __init__(self, rank: int = 0, world: int = 1, device=None, zero_copy: bool = False, staging: str = "host"):
        """device: the torch device of this rank when the process group is RCCL (`nccl`), None for gloo."""
        self.rank, self.world, self.device = rank, world, device
        self.staging = "zero_copy" if zero_copy else staging
        self.zero_copy = zero_copy

    def selfcheck(self, ctx) -> dict:
        """Start-up test of a multi-rank RCCL job: a 4-element all-reduce, and whether library <-> torch device copies
        round-trip on every rank (then `staging` becomes "device"). Raises if the all-reduce is wrong."""
        out = {"world": self.world, "staging": self.staging}
        if self.world == 1 or self.device is None:
            return out
        import torch
        import torch.distributed as dist

        t = torch.arange(4, dtype=torch.float32, device=self.device) + float(self.rank)
        dist.all_reduce(t)
        want = self.world * np.arange(4) + self.world * (self.world - 1) / 2.0
        if not np.array_equal(t.cpu().numpy(), want.astype(np.float32)):
            raise RuntimeError(f"RCCL self-check failed on rank {self.rank}: all-reduce gave {t.cpu().numpy()} instead of {want}")
        ok = 1.0
        try:
            h = (np.arange(4096, dtype=np.float32) * 0.5 + self.rank).astype(np.float32)
            buf = ctx.malloc(h.nbytes)
            try:
                ctx.h2d(buf, h)
                tt = torch.zeros(4096, dtype=torch.float32, device=self.device)
                torch.cuda.synchronize(self.device)
                ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, tt.data_ptr(), buf, h.nbytes, 3))
                if not np.array_equal(tt.cpu().numpy(), h):
                    ok = 0.0
                tt.mul_(2.0)
                torch.cuda.synchronize(self.device)
                ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, buf, tt.data_ptr(), h.nbytes, 3))
                back = np.empty_like(h)
                ctx.d2h(back, buf)
                if not np.array_equal(back, 2.0 * h):
                    ok = 0.0
            finally:
                ctx.free(buf)
        except Exception:
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if self.staging == "host" and float(flag.item()) == 1.0:
            self.staging = "device"
        out.update({"allreduce": "ok", "cross_runtime_d2d": bool(flag.item() == 1.0), "staging": self.staging})
        return out

    def _dev_tensor(self, ctx, dev_ptr: int, count: int, dtype: int):
        """(tensor RCCL operates on, copy-back function)"""
        import torch

        tdt, ndt = (torch.float64, np.float64) if dtype == 0 else (torch.float32, np.float32)
        nbytes = int(count) * (8 if dtype == 0 else 4)
        if self.staging == "zero_copy":
            return raw_device_tensor(dev_ptr, count, "<f8" if dtype == 0 else "<f4", self.device), (lambda: None)
        if self.staging == "device":
            t = torch.empty(int(count), dtype=tdt, device=self.device)
            ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, t.data_ptr(), dev_ptr, nbytes, 3))  # synchronous on the library's stream

            def back():
                torch.cuda.synchronize(self.device)
                ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, dev_ptr, t.data_ptr(), nbytes, 3))

            return t, back
        h = np.empty(int(count), dtype=ndt)
        ctx.d2h(h, dev_ptr)
        t = torch.from_numpy(h).to(self.device)

        def back_host():
            ctx.h2d(dev_ptr, t.cpu().numpy())

        return t, back_host

    # -- equally sized blocks of library device memory: every rank contributes `count` floats, receives world * count
    def allgather_dev(self, ctx, send_ptr: int, recv_ptr: int, count_f32: int):
        """recv[r * count : (r + 1) * count] = rank r's send buffer (both are library allocations on this rank's GPU)"""
        if self.world == 1:
            ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, recv_ptr, send_ptr, 4 * int(count_f32), 3))
            return
        import torch
        import torch.distributed as dist

        if self.device is None:  # gloo (tests): through host memory
            h = np.empty(int(count_f32), dtype=np.float32)
            ctx.d2h(h, send_ptr)
            outs = [torch.empty(int(count_f32), dtype=torch.float32) for _ in range(self.world)]
            dist.all_gather(outs, torch.from_numpy(h))
            ctx.h2d(recv_ptr, torch.cat(outs).numpy())
            return
        t, _ = self._dev_tensor(ctx, send_ptr, count_f32, 1)
        out = torch.empty(self.world * int(count_f32), dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize(self.device)
        if self.staging == "host":
            ctx.h2d(recv_ptr, out.cpu().numpy())
        else:
            ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, recv_ptr, out.data_ptr(), 4 * self.world * int(count_f32), 3))

    # -- small host arrays (search statistics): fixed-shape float64 all-gather
    def allgather_small(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr[None]
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(arr.copy())
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    # -- ensemble blocks: every rank contributes `per_rank` equally sized tensors (padded), all ranks receive all
    def allgather_blocks(self, local):
        """local: tensor [per_rank, ...] (same shape on every rank) -> tensor [world, per_rank, ...]."""
        import torch

        if self.world == 1:
            return local[None]
        import torch.distributed as dist

        if self.device is None:  # gloo (tests): stage through host memory
            h = local.detach().cpu().contiguous()
            outs = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(outs, h)
            return torch.stack(outs).to(local.device)
        out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1))
        return out

    # -- in-place sum of a raw device buffer over the ranks (row-sharded sessions)
    def allreduce_dev(self, ctx, dev_ptr: int, count: int, dtype: int, _force: bool = False):
        """dtype 0 = float64, 1 = float32. RCCL when the process group has a device (backend nccl): the buffer is wrapped as
        a torch tensor without a copy; otherwise (gloo, tests) it is staged through host memory. `_force`: run the
        collective even in a one-rank group (tests of the RCCL branch on a single GPU)."""
        if (self.world == 1 and not _force) or count == 0:
            return
        import torch
        import torch.distributed as dist

        np_t = np.float64 if dtype == 0 else np.float32
        if self.device is None:
            h = np.empty(int(count), dtype=np_t)
            ctx.d2h(h, dev_ptr)
            t = torch.from_numpy(h)
            dist.all_reduce(t)
            ctx.h2d(dev_ptr, h)
            return

        t, back = self._dev_tensor(ctx, dev_ptr, count, dtype)
        dist.all_reduce(t)
        torch.cuda.synchronize(self.device)
        back()

    # -- one-to-all copies for the spread initial phase (api.sclens, world > 1)
    def bcast_host(self, arr: np.ndarray, src: int) -> np.ndarray:
        """float64 host array of the same shape on every rank; returns rank `src`'s content"""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(arr.copy())
        if self.device is not None:
            t = t.to(self.device)
        dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def bcast_dev(self, ctx, dev_ptr: int, count_f32: int, src: int, _force: bool = False):
        """`count_f32` floats at a raw device pointer (allocated on every rank), from rank `src` to all, in place"""
        if (self.world == 1 and not _force) or count_f32 == 0:
            return
        import torch
        import torch.distributed as dist

        if self.device is None:  # gloo (tests): through host memory
            h = np.empty(int(count_f32), dtype=np.float32)
            if self.rank == src:
                ctx.d2h(h, dev_ptr)
            t = torch.from_numpy(h)
            dist.broadcast(t, src=src)
            if self.rank != src:
                ctx.h2d(dev_ptr, h)
            return

        t, back = self._dev_tensor(ctx, dev_ptr, count_f32, 1)
        dist.broadcast(t, src=src)
        torch.cuda.synchronize(self.device)
        if self.rank != src or _force:
            back()

    def agree(self, arr: np.ndarray) -> np.ndarray:
        """rank 0's copy of a small host array on every rank (decisions must not diverge by a rounding bit)"""
        return self.allgather_small(arr)[0]

    def barrier(self):
        if self.world > 1:
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
