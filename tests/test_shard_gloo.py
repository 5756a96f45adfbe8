"""CPU, world_size 2, gloo: the two collectives of the N > 1 path (search statistics all-gather, ensemble block
all-gather) and the end-to-end exchange logic with a fake session that stores slots in host memory."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sclens_amd.shard import Shard, owned_perturbations


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = Shard(rank, world, None)
        got = sh.allgather_small(np.array([rank, 10.0 + rank, np.nan]))
        ok = got.shape == (world, 3) and all(got[r, 0] == r and got[r, 1] == 10.0 + r for r in range(world))
        # ensemble exchange: rank r owns perturbations t % world == r; block content encodes (t, row)
        P, min_pc, ld = 5, 3, 8
        per = -(-P // world)
        blocks = torch.zeros((per, min_pc, ld))
        for qi, t in enumerate(owned_perturbations(rank, world, P)):
            blocks[qi] = t * 100 + torch.arange(min_pc)[:, None] + torch.zeros(ld)
        allb = sh.allgather_blocks(blocks)
        for r in range(world):
            for qi, t in enumerate(owned_perturbations(r, world, P)):
                ok = ok and bool(torch.all(allb[r, qi, :, 0] == t * 100 + torch.arange(min_pc)))
        # sclens(draws=None, seed=None): every rank takes rank 0's clock-derived seed
        seed = int(sh.bcast_host(np.array([float(1000 * rank + 5)]), 0)[0])
        ok = ok and seed == 5
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
