"""CPU, world_size 2, gloo: the two collectives of the N > 1 path (search statistics all-gather, ensemble block
all-gather) and the end-to-end exchange logic with a fake session that stores slots in host memory."""
import os
import socket

import numpy as np
import pytest
import torch  # noqa: F401
import torch.distributed as dist
import torch.multiprocessing as mp

from sclens_amd.shard import Shard, owned_perturbations


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class HostCtx:
    """stands in for a device context on the CPU box: "device pointers" are keys of host byte buffers, so the host-staged
    branches of Shard.allgather_dev / allreduce_dev / bcast_dev (the control flow every multi-rank sclens() call drives) run
    without a GPU"""

    def __init__(self):
        self.mem, self.next = {}, 1 << 20

    def malloc(self, nbytes):
        p = self.next
        self.next += (int(nbytes) + 255) // 256 * 256
        self.mem[p] = np.zeros(int(nbytes), dtype=np.uint8)
        return p

    def free(self, p):
        self.mem.pop(p, None)

    def _view(self, p, nbytes):
        base = max(k for k in self.mem if k <= p)
        return self.mem[base][p - base: p - base + nbytes]

    def h2d(self, dst, arr):
        a = np.ascontiguousarray(arr)
        self._view(dst, a.nbytes)[:] = a.view(np.uint8).ravel()

    def d2h(self, arr, src):
        arr.view(np.uint8).ravel()[:] = self._view(src, arr.nbytes)

    def sync(self):
        pass


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = Shard.create(None, rank, world, backend="gloo")
        assert sh.comm is None and sh.describe()["world"] == world
        got = sh.allgather_small(np.array([rank, 10.0 + rank, np.nan]))
        ok = got.shape == (world, 3) and all(got[r, 0] == r and got[r, 1] == 10.0 + r for r in range(world))
        # ensemble exchange (api._exchange_ensemble's data movement): rank r owns perturbations t % world == r; every owned
        # min_pc x ld block is exported into a send buffer and all-gathered between "device" buffers
        ctx = HostCtx()
        P, min_pc, ld = 5, 3, 8
        per = -(-P // world)
        blk = min_pc * ld
        send, recv = ctx.malloc(4 * per * blk), ctx.malloc(4 * world * per * blk)
        mine = np.zeros((per, min_pc, ld), dtype=np.float32)
        for qi, t in enumerate(owned_perturbations(rank, world, P)):
            mine[qi] = t * 100 + np.arange(min_pc)[:, None]
        ctx.h2d(send, mine)
        sh.allgather_dev(ctx, send, recv, per * blk)
        allb = np.empty((world, per, min_pc, ld), dtype=np.float32)
        ctx.d2h(allb, recv)
        for r in range(world):
            for qi, t in enumerate(owned_perturbations(r, world, P)):
                ok = ok and bool(np.all(allb[r, qi, :, 0] == t * 100 + np.arange(min_pc)))
        # row-sharded all-reduce (fp64 statistics, fp32 partial Gram matrix) and the Vr2 broadcast
        for dtype, npdt in ((0, np.float64), (1, np.float32)):
            x = (np.arange(37) * 0.25 + rank).astype(npdt)
            buf = ctx.malloc(x.nbytes)
            ctx.h2d(buf, x)
            sh.allreduce_dev(ctx, buf, x.size, dtype)
            out = np.empty_like(x)
            ctx.d2h(out, buf)
            ok = ok and np.array_equal(out, (world * np.arange(37) * 0.25 + sum(range(world))).astype(npdt))
        v = np.full(11, float(rank), dtype=np.float32)
        buf = ctx.malloc(v.nbytes)
        ctx.h2d(buf, v)
        sh.bcast_dev(ctx, buf, v.size, world - 1)
        ctx.d2h(v, buf)
        ok = ok and bool(np.all(v == world - 1))
        ok = ok and sh.selfcheck(ctx).get("selfcheck") == "ok"
        # the reducer a row-sharded session gets on this transport is a host callback into allreduce_dev
        fn, user = sh.reducer(ctx)
        ok = ok and user is None and callable(fn)
        # sclens(draws=None, seed=None): every rank takes rank 0's clock-derived seed
        seed = int(sh.bcast_host(np.array([float(1000 * rank + 5)]), 0)[0])
        ok = ok and seed == 5
        ok = ok and np.array_equal(sh.agree(np.array([float(rank), 2.0])), np.array([0.0, 2.0]))
        sh.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_collectives_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)]
