"""Row-sharded sclens() under real process-level collectives (torch.distributed):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tests/shard_rows_worker.py [--backend gloo|nccl] [--out result.json]

gloo: the ranks may share one GPU (the all-reduce is staged through host memory); nccl: one GPU per rank, the library's own
RCCL communicator on its device buffers. Every rank builds the same synthetic matrix and draws, takes its block of cells, runs
atlas.sclens_row_sharded; rank 0 also runs the unsharded api.sclens and compares (same checks as tests/test_gpu_atlas.py)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--out", default="")
    ap.add_argument("--cells", type=int, default=900)
    ap.add_argument("--genes", type=int, default=400)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from sclens_amd import api, atlas
    from sclens_amd._lib import Context
    from sclens_amd.shard import Shard
    from sclens_amd.synth import synth_counts

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_id = local_rank if args.backend == "nccl" else min(local_rank, torch.cuda.device_count() - 1)
    torch.cuda.set_device(dev_id)
    dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    ctx = Context(dev_id)
    shard = Shard.create(ctx, rank, world, backend=args.backend)
    N, M = args.cells, args.genes
    X = api._csc_f32(synth_counts(N, M, seed=2, C=5, marker_frac=0.2, marker_sd=1.5))
    d = api.make_draws_native(X, seed=41)
    a, b = atlas.row_block(rank, world, N)
    res = atlas.sclens_row_sharded(X.tocsr()[a:b].tocsc(), a, N, d, shard, n_perturb=6, ctx=ctx)
    ok, info = True, {}
    if rank == 0:
        from test_gpu_atlas import _compare

        ref = api.sclens(X, draws=d, n_perturb=6, ctx=ctx)
        try:
            _compare(res, ref)
        except AssertionError as e:
            ok = False
            info["error"] = repr(e)
        info.update({"backend": args.backend, "world": world, "cells": N, "genes": M, "signals": int(len(res["signal_ev"])),
                     "n_search": int(res["n_search"]), "p_": float(res["p_"]), "sig_id": [int(i) for i in res["sig_id"]],
                     "max_rel_eig_diff": float(np.abs(res["L"] - ref["L"]).max() / ref["L"].max()),
                     "sharded_wall_s": round(res["wall_s"], 3), "unsharded_wall_s": round(ref["wall_s"], 3), "ok": ok})
        print(json.dumps(info))
        if args.out:
            with open(args.out, "w") as fh:
                json.dump(info, fh)
    dist.barrier()
    shard.close()
    dist.destroy_process_group()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
