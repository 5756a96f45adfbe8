"""Randomised end-to-end parity sweep against the oracle (GPU box; not part of the test-suite).
Usage: fuzz_parity.py [cases] [seed] [ensemble_tail = auto | certified | converged]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sclens_oracle as O  # checker
from sclens_amd import api
from sclens_amd.synth import synth_counts

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tail = sys.argv[3] if len(sys.argv) > 3 else "auto"
bad = again = 0
t0 = time.time()
for c in range(cases):
    N, M = int(rng.integers(90, 420)), int(rng.integers(90, 420))
    C = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 10 ** 6))
    cent = "median" if rng.random() < 0.25 else "mean"
    streams = int(rng.integers(1, 4))
    only = os.environ.get("FUZZ_ONLY")  # replay single cases of a sweep (the parameters of every case are still drawn)
    mf, ms = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.8, 1.8))
    if only and str(c) not in only.split(","):
        continue
    try:
        X = synth_counts(N, M, seed=seed, C=C, marker_frac=mf, marker_sd=ms,
                         min_genes_per_cell=5, min_cells_per_gene=4)
    except Exception as e:  # the generator can fail its QC invariants for tiny shapes
        print(c, "skip (synth):", e)
        continue
    d = api.make_draws_native(X, seed=seed, host_sampler=True)
    od = O.Draws(d.z_idx1, d.z_idx2, d.X_r, d.p_th, d.sampler)
    ref = O.sclens(X, od, n_perturb=4, null_tol=O.NULL_DROP, centering=cent, max_search_iters=25)
    res = api.sclens(X, draws=d, n_perturb=4, centering=cent, streams=streams, max_search_iters=25, ensemble_tail=tail)
    again += len(res.get("tail_redo", []))
    msgs = []
    if np.abs(res["L"] - ref["L"]).max() > 2e-4 * ref["L"].max():
        msgs.append("L")
    if len(res.get("signal_ev", [])) != len(ref.get("signal_ev", [])):
        msgs.append(f"k {len(res.get('signal_ev', []))} vs {len(ref.get('signal_ev', []))}")
    if res["p_"] != ref["p_"] or res["n_search"] != ref["n_search"]:
        msgs.append(f"search {res['n_search']}/{res['p_']} vs {ref['n_search']}/{ref['p_']}")
    if "sig_id" in ref and not np.array_equal(res.get("sig_id"), ref["sig_id"]):
        msgs.append(f"sig_id {res.get('sig_id')} vs {ref['sig_id']}")
    if "robustness_scores" in ref and "robustness_scores" in res and not np.array_equal(res["robustness_scores"]["a_b"], ref["robustness_scores"]["a_b"]):
        msgs.append("a_b")
    if msgs:
        bad += 1
        # how close was the decision that differs? (margins of the thresholds involved)
        extra = ""
        if "robustness_scores" in ref and "robustness_scores" in res:
            extra = f" rob {np.round(res['robustness_scores']['rob_score'], 4)} vs {np.round(ref['robustness_scores']['rob_score'], 4)}"
        gap = np.min(np.abs(ref["L"] - ref["lambda_c"])) / ref["lambda_c"]
        if "search_trace" in res and "search_trace" in ref:
            da = [float(t[1]) for _, t in res["search_trace"]]
            db = [float(t[1]) for _, t in ref["search_trace"]]
            print("   p_th", ref["p_th"], "device d5[1]:", np.round(da, 5), "oracle:", np.round(db, 5))
        print(c, f"N={N} M={M} C={C} seed={seed} {cent} streams={streams}: MISMATCH", msgs, f"min |L - lambda_c|/lambda_c = {gap:.2e}", extra, flush=True)
    else:
        print(c, f"N={N} M={M} C={C} {cent} streams={streams}: ok k={len(ref.get('signal_ev', []))} S={ref['n_search']}", flush=True)
print(f"{cases} cases, {bad} mismatches, ensemble_tail = {tail}: {again} members solved again, {time.time() - t0:.0f} s")
