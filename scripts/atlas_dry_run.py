"""Atlas configuration (BASELINE.json configs[4]: 1M cells x 30k genes on 8 GPUs) -- dry run of ONE rank on one
GPU: rank 0's slab of 125 000 cells (`sclens_amd.atlas.synth_slabs`: generated chunk-wise, never dense on the host) goes through
the row-sharded session in the atlas mode (`sclens_hip_session_create_sharded_drawn`: this rank's candidates drawn on the device;
search / ensemble in rounds of `world`, this rank decomposing one evaluation per round) with the inter-rank exchange STUBBED (the
callbacks count calls and bytes and multiply the buffer by the number of ranks, as if all ranks held this slab: no data moves), so
the numbers are this rank's compute time and HBM footprint of the 8-GPU form. (The matrix itself is checked against float64 on ONE GPU through the
chunked session: tests/test_gpu_chunked.py, scripts/atlas_chunked_run.py.) Usage: atlas_dry_run.py [N_total world out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sclens_amd import _lib, api
from sclens_amd.atlas import row_block


def raw_device_tensor(dev_ptr, count, typestr, device):
    """zero-copy torch view of library-owned device memory (`__cuda_array_interface__`); this script only"""

    class _Raw:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(dev_ptr), False), "version": 3}

    return torch.as_tensor(_Raw(), device=device)


N_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = int(os.environ.get("ATLAS_M", "30000"))
out_path = sys.argv[3] if len(sys.argv) > 3 else None
import torch  # device-level free memory (mem_get_info) and runtime order

torch.cuda.init()
ctx = api.Context(0)
free0, total = torch.cuda.mem_get_info(0)
log = {"N_total": N_total, "M": M, "world": world, "hbm_total_GB": round(total / 1e9, 1)}
r0, r1 = row_block(0, world, N_total)
t0 = time.perf_counter()


def cached_slab():
    """rank 0's slab of the seeded atlas matrix, from the slab files of sclens_amd.atlas.synth_slabs (SCLENS_BENCH_CACHE, default the temp
    dir): generated once per box by as many processes as memory allows and shared with tests/test_gpu_chunked.py's cfg5 case"""
    import shutil
    import tempfile

    from sclens_amd import atlas

    d = os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir())
    try:
        room = d not in ("", "0") and shutil.disk_usage(d).free > 8.5 * N_total * M * 0.11  # the data slabs: 8 bytes per stored entry
    except OSError:
        room = False
    if room:
        return atlas.synth_slabs(N_total, M, 20240427 + 4, world).slab(0)
    from sclens_amd.synth import synth_counts_rows  # no room for the slab files: this rank's slab alone, not kept

    return synth_counts_rows(N_total, M, 20240427 + 4, r0, r1)


X = api._csc_f32(cached_slab())
log["slab"] = {"rows": [r0, r1], "nnz": int(X.nnz), "synth_s": round(time.perf_counter() - t0, 1)}
print("[atlas] slab generated", log["slab"], file=sys.stderr, flush=True)
t0 = time.perf_counter()
d = api.make_draws_native(X, seed=1000, device_candidates=True)  # R1 on the device (this rank's part of the global draw); R2 on the host
Xr = api._resolve(d.X_r)
log["draws_s"] = round(time.perf_counter() - t0, 1)
print("[atlas] host draws (null matrix)", log["draws_s"], "s", file=sys.stderr, flush=True)
stat = {"calls": 0, "bytes": 0, "largest": 0, "reduce_calls": 0, "reduce_bytes": 0}


def _scale(dev_ptr, count, dtype):
    t = raw_device_tensor(dev_ptr, int(count), "<f8" if dtype == 0 else "<f4", torch.device("cuda", 0))
    t.mul_(float(world))
    torch.cuda.synchronize()


def stub(_user, dev_ptr, count, dtype):
    """stands in for the sum over `world` ranks: as if every rank held this same slab (buffer *= world), so that the
    statistics stay consistent with N_total; no data moves"""
    nb = int(count) * (8 if dtype == 0 else 4)
    stat["calls"] += 1
    stat["bytes"] += nb
    stat["largest"] = max(stat["largest"], nb)
    _scale(dev_ptr, count, dtype)
    return 0


def stub_to(_user, dev_ptr, count, dtype, root):
    """the sum onto one rank (the Gram matrix of an evaluation that `root` decomposes)"""
    stat["reduce_calls"] += 1
    stat["reduce_bytes"] += int(count) * (8 if dtype == 0 else 4)
    if root == 0:
        _scale(dev_ptr, count, dtype)
    return 0


reducer, reducer_to = _lib.ALLREDUCE_FN(stub), _lib.REDUCE_FN(stub_to)
times = {}


def timed(name, f):
    ctx.sync()
    t = time.perf_counter()
    r = f()
    ctx.sync()
    times[name] = round(time.perf_counter() - t, 3)
    print(f"[atlas] {name}: {times[name]} s", file=sys.stderr, flush=True)
    return r


nnz_global = int(X.nnz) * world  # as if every rank held a slab like this one
ses = timed("session_create_sharded_drawn", lambda: api.Session.create_sharded_drawn(ctx, X, r0, N_total, nnz_global, d.cand_seed, (reducer, None)))
try:
    nloc = ses.ncand_local
    ses.set_candidate_range(0, nloc * world)
    ses.set_reduce_to((reducer_to, None))
    log["candidates_local"] = nloc
    Lr = timed("null_spectrum", lambda: ses.null_spectrum(Xr))
    L, _ = timed("data_spectrum", lambda: ses.data_spectrum())
    k = 8  # a fixed number of signal vectors: the slab's own spectrum is not the atlas's
    timed("signal_vectors", lambda: ses.signal_vectors(k))
    _, r_vr2 = timed("binary_basis", lambda: ses.binary_basis())
    n_2 = int(round(r_vr2 / 2))
    # one ROUND of the search: `world` sparsities, this rank contributes to all and decomposes the first
    for rd in range(2):
        its = [rd * world + e for e in range(world)]
        ms = [int(round((1 - (0.999 - 0.001 * it)) * M * N_total)) for it in its]
        seeds = [api.sample_seed_for(d.sample_seed, "search", it) for it in its]
        timed(f"search_round_{rd}", lambda: ses.search_round_seeded(seeds, ms, list(range(world)), 0, n_2))
    min_pc = 12
    m_pert = int(round(0.015 * M * N_total))
    mine = {}
    for rd in range(2):
        ts = [rd * world + e for e in range(world)]
        _, nc = timed(f"perturb_round_{rd}", lambda: ses.perturb_round_seeded(ts, [api.sample_seed_for(d.sample_seed, "perturb", t) for t in ts],
                                                                              [m_pert] * world, list(range(world)), 0, min_pc))
        mine[ts[0]] = nc[0]

    def fill_other_slots():
        # the members the other ranks decomposed arrive by the ensemble exchange; here: this rank's own slot copied into theirs
        buf = ctx.malloc(4 * min_pc * ses.slot_ld())
        try:
            for t0, c0 in mine.items():
                ses.export_slot(t0, min_pc, buf)
                for e in range(1, world):
                    ses.import_slot(t0 + e, min_pc, c0, buf)
        finally:
            ctx.free(buf)

    timed("ensemble_exchange_stand_in", fill_other_slots)
    timed("robustness", lambda: ses.robustness(k, 2 * world))
    timed("gene_basis", lambda: ses.gene_basis(np.sort(L)[::-1][:k].copy()))
    free1, _ = torch.cuda.mem_get_info(0)
    log["hbm_used_GB"] = round((free0 - free1) / 1e9, 1)
finally:
    ses.close()
log["times_s"] = times
log["stubbed_allreduce"] = {"calls": stat["calls"], "total_GB": round(stat["bytes"] / 1e9, 2), "largest_GB": round(stat["largest"] / 1e9, 2),
                            "ring_estimate_s_at_153GBps_per_link": round(2 * (world - 1) / world * stat["bytes"] / 153e9, 2)}
log["stubbed_reduce_to_root"] = {"calls": stat["reduce_calls"], "total_GB": round(stat["reduce_bytes"] / 1e9, 2),
                                 "estimate_s_at_153GBps_per_link": round(stat["reduce_bytes"] / 153e9, 2)}
S_est, P = 19, 20
rounds_s, rounds_p = -(-S_est // world), -(-P // world)
log["projected_rank_wall_s"] = round(times["session_create_sharded_drawn"] + times["null_spectrum"] + times["data_spectrum"] +
                                     times["signal_vectors"] + times["binary_basis"] + rounds_s * times["search_round_1"] +
                                     rounds_p * times["perturb_round_1"] + times["robustness"] + times["gene_basis"], 1)
log["projection"] = (f"S = {S_est} evaluations in {rounds_s} rounds of {world}, P = {P} members in {rounds_p} rounds; first phase replicated; "
                     "exchange not included (stubbed; estimates above)")
txt = json.dumps(log, indent=1)
print(txt)
if out_path:
    open(out_path, "w").write(txt + "\n")
