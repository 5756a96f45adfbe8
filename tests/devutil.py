"""Helpers for the -m gpu tests: move numpy arrays through the C ABI's raw device entry points."""
import ctypes as C
import threading

import numpy as np


class DevArray:
    def __init__(self, ctx, arr=None, nbytes=None):
        self.ctx = ctx
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            self.nbytes = arr.nbytes
            self.p = ctx.malloc(max(16, arr.nbytes))
            ctx.h2d(self.p, arr)
        else:
            self.nbytes = nbytes
            self.p = ctx.malloc(max(16, nbytes))
            ctx.memset(self.p, 0, nbytes)

    def get(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.d2h(out, self.p)
        return out

    def free(self):
        if self.p:
            self.ctx.free(self.p)
            self.p = None


def pad_rows(a, ld):
    """row-major 2-D array -> same rows with leading dimension ld (zero padded)."""
    a = np.asarray(a)
    out = np.zeros((a.shape[0], ld), dtype=a.dtype)
    out[:, : a.shape[1]] = a
    return out


def rup(x, m):
    return (x + m - 1) // m * m


class ThreadShard:
    """rank `rank` of `world` threads of this process; collectives by a barrier and shared slots"""

    class Group:
        def __init__(self, world):
            self.world = world
            self.bar = threading.Barrier(world, timeout=300)
            self.slots = [None] * world
            self.nreduce = 0
            self.bytes = 0

    def __init__(self, group, rank):
        self.g, self.rank, self.world, self.comm = group, rank, group.world, None

    def _exchange(self, value):
        self.g.slots[self.rank] = value
        self.g.bar.wait()
        out = list(self.g.slots)
        self.g.bar.wait()
        return out

    def allgather_small(self, arr):
        return np.stack(self._exchange(np.ascontiguousarray(arr, dtype=np.float64).copy()))

    def agree(self, arr):
        return self.allgather_small(arr)[0]

    def all_ok(self, exc=None, where=""):  # the product's own status agreement on this transport
        from sclens_amd.shard import Shard

        return Shard.all_ok(self, exc, where)

    def allreduce_dev(self, ctx, dev_ptr, count, dtype):
        h = np.empty(int(count), dtype=np.float64 if dtype == 0 else np.float32)
        ctx.d2h(h, dev_ptr)
        parts = self._exchange(h)
        tot = parts[0].copy()
        for p in parts[1:]:  # same order on every rank: identical bits everywhere
            tot += p
        ctx.h2d(dev_ptr, tot)
        if self.rank == 0:
            self.g.nreduce += 1
            self.g.bytes += tot.nbytes

    def barrier(self):
        self.g.bar.wait()

    def bcast_host(self, arr, src):
        return self.allgather_small(arr)[src]

    def bcast_dev(self, ctx, dev_ptr, count_f32, src):
        h = np.empty(int(count_f32), dtype=np.float32)
        if self.rank == src:
            ctx.d2h(h, dev_ptr)
        got = self._exchange(h if self.rank == src else None)[src]
        if self.rank != src:
            ctx.h2d(dev_ptr, got)

    def allgather_dev(self, ctx, send_ptr, recv_ptr, count_f32):
        h = np.empty(int(count_f32), dtype=np.float32)
        ctx.d2h(h, send_ptr)
        ctx.h2d(recv_ptr, np.concatenate(self._exchange(h)))

    def reducer(self, ctx):
        """(sclens_hip_allreduce_fn, user) of a row-sharded session, as Shard.reducer on the host-staged transport"""
        from sclens_amd import _lib

        def cb(_user, dev_ptr, count, dtype):
            try:
                self.allreduce_dev(ctx, dev_ptr, count, dtype)
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        return _lib.ALLREDUCE_FN(cb), None

    def reducer_to(self, ctx):
        """(sclens_hip_reduce_fn, user): the sum lands on `root` only; the other ranks' buffers are POISONED (NaN), so a rank that
        wrongly consumes a sum it is not the root of fails loudly"""
        from sclens_amd import _lib

        def cb(_user, dev_ptr, count, dtype, root):
            try:
                h = np.empty(int(count), dtype=np.float64 if dtype == 0 else np.float32)
                ctx.d2h(h, dev_ptr)
                parts = self._exchange(h)
                if self.rank == root:
                    tot = parts[0].copy()
                    for p in parts[1:]:
                        tot += p
                else:
                    tot = np.full_like(h, np.nan)
                ctx.h2d(dev_ptr, tot)
                if self.rank == 0:
                    self.g.nreduce += 1
                    self.g.bytes += tot.nbytes
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        return _lib.REDUCE_FN(cb), None
