#!/usr/bin/env python3
"""bench.py -- sclens() wall-clock and cells*genes/s on synthetic count matrices (BASELINE.json metric).

One "step" = one complete sclens() call (scLENS.jl:649-832: normalisation, data/null/binary decompositions,
MP/TW thresholding, sparsity search, 20-member perturbation ensemble, robustness scoring, gene basis) on a
seeded synthetic cells x genes count matrix that is already resident in host CSC form; every random draw of the
call is generated inside the timed region. Default workload = BASELINE.json configs[1] (10 000 x 20 000).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : dominant kernel `trd_colB` (HBM-bound symmetric matrix-vector product of the tridiagonalisation),
                 algorithmic bytes = the lower triangle of the symmetric trailing matrix, 2 n'(n'+1) B with n' = n-j-1,
                 per launch (half of SURVEY 8(d)'s full-read figure 4 n'^2, which is also reported); duration = one
                 HIP-event pair on the library's stream around all n-1 launches of one tridiagonalisation, back to back.
  cpu_baseline : the oracle (float64 NumPy/SciPy port of the reference CPU path) timed on the host cores on a
                 bounded sample, stage-extrapolated to the workload (see `sample`).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {  # name -> (N cells, M genes, index in BASELINE.json configs)
    "tiny": (600, 900, 0),
    "tiny_gt": (900, 400, 0),  # cells > genes, for --row-shard smoke runs
    "cfg2": (10000, 20000, 1),
    "cfg3": (50000, 30000, 2),
    "cfg4": (100000, 30000, 3),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per trd_colB launch / algorithmic bytes, from the PMC passes in profiles/r01_pmc (rocprofv3 --pmc FETCH_SIZE
# and --pmc WRITE_SIZE in separate runs, first panel at n = 10000: FETCH_SIZE 99 731 KB x 2 (gfx950 reports half the bytes
# of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE 7 149 KB = 211.6 MB against 197.5 MB algorithmic).
PMC_TRAFFIC_RATIO = 1.07


def roofline_probe(ctx, n):
    """Every trd_colB launch of one tridiagonalisation of order n (same grids / arguments as the real reduction), back
    to back on the library's stream between one pair of HIP events (sclens_hip_symv_probe)."""
    import ctypes as C

    launches, ms, nbytes = C.c_int64(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.sclens_hip_symv_probe(ctx.h, n, C.byref(launches), C.byref(ms), C.byref(nbytes)))  # warm-up
    ctx.check(ctx.lib.sclens_hip_symv_probe(ctx.h, n, C.byref(launches), C.byref(ms), C.byref(nbytes)))
    gbs = nbytes.value / (ms.value * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "trd_colB", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": round(PMC_TRAFFIC_RATIO * nbytes.value / max(1, launches.value), 1), "launches": launches.value,
            "avg_launch_us": round(ms.value * 1e3 / max(1, launches.value), 2),
            "algorithmic_bytes_per_launch_avg": round(nbytes.value / max(1, launches.value), 1), "n": n,
            "achieved_vs_full_read_4n2": round(2 * gbs, 1)}


def cpu_baseline(N, M, n_search, n_perturb):
    """Oracle (port of the reference CPU path) on the host cores: every stage timed once on a bounded sample and
    scaled to the workload by its complexity, times the call counts observed in the GPU run."""
    from oracle import sclens_oracle as O  # checker / baseline only
    from sclens_amd.synth import synth_counts

    n, K = min(N, M), max(N, M)
    ns = min(n, 4000)  # ~10-20 s of CPU work on the box's host cores; n^3 extrapolation factor <= 16 at cfg2
    Ns, Ms = (ns, int(ns * M / N)) if N <= M else (int(ns * N / M), ns)
    Xs = synth_counts(Ns, Ms, seed=11)
    t0 = time.perf_counter()
    S = O.logn_scale(O.pre_scale(Xs))
    t_scale = time.perf_counter() - t0
    t0 = time.perf_counter()
    Y = O.wishart_matrix(S, 2 if Ns > Ms else 1)
    t_gram = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.get_eigen(Y)
    t_eig = time.perf_counter() - t0
    calls = 3 + n_search + n_perturb
    f3 = (n / ns) ** 3
    T = calls * (t_scale * (N * M) / (Ns * Ms) + t_gram * (n * n * K) / (ns * ns * max(Ns, Ms)) + t_eig * f3)
    T += n_search * t_gram * (n ** 3) / (ns * ns * max(Ns, Ms))  # corr_mat (scLENS.jl:742): ~n^3 flop per iteration
    cores = os.cpu_count() or 1
    try:  # threads the BLAS/LAPACK behind NumPy/SciPy actually uses
        from threadpoolctl import threadpool_info

        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [cores])
    except Exception:
        pass
    return {"value": round(N * M / T, 1), "unit": "cells*genes/s", "cores": cores, "kind": "port",
            "wall_s_extrapolated": round(T, 1),
            "sample": (f"oracle normalise+Gram+dsyevr timed once at {Ns}x{Ms} ({t_scale:.2f}s, {t_gram:.2f}s, {t_eig:.2f}s), "
                       f"scaled by NM, n^2K and n^3 to {N}x{M}, times {calls} decompositions (S={n_search}, P={n_perturb}) "
                       f"+ {n_search} corr GEMMs; BLAS threads = {cores} (host has {os.cpu_count()} cores)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", default="cfg2", choices=list(CONFIGS))
    ap.add_argument("--n-perturb", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--batch", action="store_true", help="merge the column steps of concurrent tridiagonalisations into shared launches (opt-in)")
    ap.add_argument("--row-shard", action="store_true",
                    help="cells > genes configs on N > 1 GPUs: every rank holds a block of cells, partial Gram matrices are "
                         "all-reduced (SURVEY 8e-iii, sclens_amd/atlas.py) instead of distributing whole decompositions")
    ap.add_argument("--streams", type=int, default=3, help="concurrent decompositions per GPU (worker sessions on own HIP streams)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--stage-timing", action="store_true", help="per-stage HIP-event totals on stderr (adds syncs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if args.backend != "nccl":  # gloo test mode: several ranks may share one GPU
        local_rank = min(local_rank, torch.cuda.device_count() - 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world,
                                device_id=dev if args.backend == "nccl" else None)
    from sclens_amd import api
    from sclens_amd._lib import Context
    from sclens_amd.shard import Shard
    from sclens_amd.synth import synth_counts

    ctx = Context(local_rank)
    shard = Shard(rank, world, dev if (world > 1 and args.backend == "nccl") else None)
    N, M, cfg_index = CONFIGS[args.config]
    t0 = time.perf_counter()
    X = api._csc_f32(synth_counts(N, M, seed=20240427 + cfg_index))  # PCG64(20240427 + config_index)
    t_synth = time.perf_counter() - t0

    row_shard = args.row_shard and N > M
    if row_shard:
        from sclens_amd import atlas

        r0, r1 = atlas.row_block(rank, world, N)
        X_rows = api._csc_f32(X.tocsr()[r0:r1].tocsc())

    def one_step(step):
        t_d = time.perf_counter()
        draws = api.make_draws_native(X, seed=1000 + step, async_null=True, async_candidates=not row_shard)
        one_step.draws_s = time.perf_counter() - t_d  # R1-R3 inside the timed region; R4/R5 on the device inside sclens()
        if row_shard:  # global draws (identical on every rank), local cells
            return atlas.sclens_row_sharded(X_rows, r0, N, draws, shard, n_perturb=args.n_perturb, ctx=ctx, gather=False,
                                            verbose=args.verbose)
        return api.sclens(X, draws=draws, ctx=ctx, n_perturb=args.n_perturb, shard=shard, streams=args.streams,
                          batch=args.batch, verbose=args.verbose and rank == 0)

    def fence():
        shard.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    if args.stage_timing:
        ctx.set_timing(True)
    res = None
    for w in range(args.warmup):
        res = one_step(-1 - w)
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        res = one_step(s)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if args.stage_timing and rank == 0:
        st = {k: ctx.timing(k) for k in ("scale", "gram", "sytrd", "stebz", "stein", "ormtr", "chefsi", "corr", "recover")}
        print("stage totals (ms, calls):", {k: (round(v[0], 1), v[1]) for k, v in st.items()}, "wall_s", round(dt, 2), file=sys.stderr)
    if rank == 0:
        ms_per_step = dt / max(1, args.steps) * 1e3
        out = {
            "metric": "sclens() cells*genes/s (wall-clock of one full sclens() call)", "value": round(N * M * args.steps / dt, 1),
            "unit": "cells*genes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic Poisson-lognormal counts {N} cells x {M} genes, sparsity "
                                   f"{1 - X.nnz / (N * M):.3f}, full sclens() incl. sparsity search and {args.n_perturb}-member "
                                   f"perturbation ensemble", "N": N, "M": M, "nnz": int(X.nnz), "n_perturb": args.n_perturb,
                       "parallelism": (f"cells row-sharded over {world} ranks: per decomposition 4 small all-reduces + one "
                                       f"all-reduce of the {M}x{M} fp32 partial Gram matrix, eigen-solver replicated" if row_shard
                                       else f"single GPU, {args.streams} concurrent decompositions (HIP streams)" if world == 1 else
                                       f"search rounds of {world}x{args.streams} + ensemble t%{world}, 1 RCCL all-gather")},
            "sclens_wall_s": round(dt / max(1, args.steps), 3),
            "observed": {"signals": int(len(res.get("signal_ev", []))), "robust_signals": int(len(res.get("sig_id", []))),
                         "search_iters": int(res["n_search"]), "p_": res["p_"], "synth_s": round(t_synth, 1),
                         "ensemble_partial_eig": {"used": int(res["partial_eig"][0]), "fallback_to_full": int(res["partial_eig"][1])},
                         "phase_s_rank0_last_step": dict({"draws_host": round(one_step.draws_s, 4)}, **res.get("phase_s", {}))},
        }
        if not args.no_roofline:
            out["roofline"] = roofline_probe(ctx, min(N, M))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, M, int(res["n_search"]), args.n_perturb)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
