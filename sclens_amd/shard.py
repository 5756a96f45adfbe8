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
    """rank/world + the two collectives. `world == 1` needs no torch."""

    def __init__(self, rank: int = 0, world: int = 1, device=None, zero_copy: bool = False):
        """device: the torch device of this rank when the process group is RCCL (`nccl`), None for gloo.
        zero_copy: hand RCCL a view of the library's own device buffers instead of a torch-allocated staging tensor.
        Off by default: the PyTorch-ROCm wheel and libsclens_hip.so carry two instances of the HIP runtime, and memory
        allocated by one is a foreign pointer to the other; device-to-device copies between the two work (measured),
        RCCL on a foreign buffer across GPUs could not be tested on a one-GPU box, so the collectives only ever see
        torch-allocated memory (one extra copy each way: 0.3 ms for the 0.4 GB Vr2 broadcast at cfg2)."""
        self.rank, self.world, self.device, self.zero_copy = rank, world, device, zero_copy

    def _dev_tensor(self, ctx, dev_ptr: int, count: int, dtype: int):
        """(tensor RCCL operates on, copy-back function)"""
        import torch

        if self.zero_copy:
            return raw_device_tensor(dev_ptr, count, "<f8" if dtype == 0 else "<f4", self.device), (lambda: None)
        t = torch.empty(int(count), dtype=torch.float64 if dtype == 0 else torch.float32, device=self.device)
        nbytes = int(count) * (8 if dtype == 0 else 4)
        ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, t.data_ptr(), dev_ptr, nbytes, 3))  # synchronous on the library's stream

        def back():
            torch.cuda.synchronize(self.device)
            ctx.check(ctx.lib.sclens_hip_dev_memcpy(ctx.h, dev_ptr, t.data_ptr(), nbytes, 3))

        return t, back

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


def search_schedule(p_step: float, max_iters: int = 200) -> List[float]:
    """p_ of iteration it, produced by the same repeated `p_ -= p_step` as the reference (Appendix A21)."""
    p = 0.999
    out = []
    for _ in range(max_iters):
        out.append(p)
        p -= p_step
    return out


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
