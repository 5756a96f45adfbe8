"""BASELINE configs[4] (1 000 000 cells x 30 000 genes) on ONE MI355X through the chunked session (sclens_amd.atlas.sclens_chunked): the
data matrix and the null matrix are generated as 8 row slabs (sclens_amd.atlas.synth_slabs / null_slabs, cached under SCLENS_BENCH_CACHE),
uploaded chunk by chunk (25 GB of CSC in HBM) and every decomposition sums the chunks' Gram contributions. Logs the phase times, the
pattern builds / chunk visits, the pool's peak, and -- when tests/golden/cfg5_f64_spectra.npz exists -- the distance of the spectra,
lambda_c and the signal count from float64.
Usage: atlas_chunked_run.py [--n-total N] [--m M] [--chunks W] [--stop-after spectra] [--precision 0|1] [--n-perturb P] [--max-search S] [--out f.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sclens_amd import _lib, api, atlas

ap = argparse.ArgumentParser()
ap.add_argument("--n-total", type=int, default=1_000_000)
ap.add_argument("--m", type=int, default=30_000)
ap.add_argument("--chunks", type=int, default=8)
ap.add_argument("--stop-after", default=None)
ap.add_argument("--precision", type=int, default=1)
ap.add_argument("--n-perturb", type=int, default=20)
ap.add_argument("--max-search", type=int, default=None)
ap.add_argument("--cache-gb", type=int, default=None)
ap.add_argument("--out")
a = ap.parse_args()
T0 = time.perf_counter()
log = lambda m: print(f"[{time.perf_counter() - T0:7.1f} s] {m}", file=sys.stderr, flush=True)
SYNTH_SEED, DRAW_SEED = 20240427 + 4, 1000
S = atlas.synth_slabs(a.n_total, a.m, SYNTH_SEED, a.chunks, log=log)
log(f"data slabs: {S.nnz_total} stored entries")
R = atlas.null_slabs(S, DRAW_SEED, log=log)
log(f"null slabs: {R.nnz_total} stored entries")
gen_s = time.perf_counter() - T0
ctx = api.Context(0)
ctx.set_option("precision", a.precision)
if a.cache_gb is not None:
    ctx.set_option("chunk_cache_gb", a.cache_gb)
lib = _lib.load()
lib.sclens_hip_pool_peak(0, 1)
d = api.Draws(None, None, None, float(lib.sclens_noise_baseline_exact(min(a.n_total, a.m))), None, DRAW_SEED)
d.cand_seed = DRAW_SEED
res = atlas.sclens_chunked(S, R, d, ctx=ctx, stop_after=a.stop_after, n_perturb=a.n_perturb, max_search_iters=a.max_search, log=log, verbose=True)
out = {"N": a.n_total, "M": a.m, "chunks": a.chunks, "precision": a.precision, "generation_s": round(gen_s, 1), "wall_s": round(res["wall_s"], 2),
       "phase_s": res["phase_s"], "lambda_c": float(res["lambda_c"]), "k": int(res["k"]), "lambda_max": float(res["L"][-1]),
       "chunk_builds": res.get("chunk_builds"), "chunk_visits": res.get("chunk_visits"),
       "pool_peak_GB": round(lib.sclens_hip_pool_peak(0, 0) / 1e9, 1)}
for key in ("p_", "n_search", "n_cand", "min_pc", "partial_eig"):
    if key in res:
        out[key] = res[key] if not hasattr(res[key], "tolist") else res[key].tolist()
if "sig_id" in res:
    out["sig_id"] = res["sig_id"].tolist()
    out["rob_score"] = np.round(res["robustness_scores"]["rob_score"], 5).tolist()
    out["a_b_first_members"] = res["robustness_scores"]["a_b"][:, :4].tolist()
    out["search_d5_second_smallest"] = [round(float(t[1][1]), 6) for t in res["search_trace"]]
    out["signal_ev"] = np.round(res["signal_ev"], 6).tolist()
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg5_f64_spectra.npz")
if os.path.exists(gold) and a.n_total == 1_000_000 and a.m == 30_000:
    g = np.load(gold)
    out["vs_float64"] = {"max_abs_err_L": float(np.abs(res["L"] - g["L"]).max()), "max_abs_err_Lr": float(np.abs(res["Lr"] - g["Lr"]).max()),
                         "tolerance_4_sqrt_n_eps32_lmax": float(4 * np.sqrt(a.m) * 5.96e-8 * g["L"][-1]), "lambda_c_f64": float(g["lambda_c"]),
                         "k_f64": int(g["k"])}
txt = json.dumps(out, indent=1)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")
