"""float64 check of the retained-signal count at a BASELINE.json configuration (GPU box, minutes of
host time): the float64 NumPy/SciPy oracle (test infrastructure, oracle/sclens_oracle.py) scales the same synthetic matrix,
forms both Gram matrices (data, null) in float64, takes their eigenvalues with LAPACK (values only), and runs the MP / TW
statistics; the device path runs its data / null spectra and the same statistics. Compared: L (max abs difference relative to
lambda_max), lambda_c, the signal count k, the eigenvalues on either side of the cut, and what the guard band refined.
Usage: check_signal_count_f64.py [cfg3|cfg4] [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla

from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd.synth import synth_counts

CFG = {"tiny": (900, 400, 0), "cfg3": (50000, 30000, 2), "cfg4": (100000, 30000, 3)}
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M, idx = CFG[cfg]
t0 = time.perf_counter()
X = api._csc_f32(synth_counts(N, M, seed=20240427 + idx))
draws = api.make_draws_native(X, seed=1000)
X_r = api._resolve(draws.X_r)
log = {"config": cfg, "N": N, "M": M, "nnz": int(X.nnz), "synth_s": round(time.perf_counter() - t0, 1)}

# ---- device: data and null spectra, thresholds, guard band (the first phase of api.sclens)
t0 = time.perf_counter()
res = api.sclens(X, draws=draws, n_perturb=2, max_search_iters=5, streams=1)
log["device_s"] = round(time.perf_counter() - t0, 1)
L32, lc32, k32 = res["L"], float(res["lambda_c"]), int(len(res.get("signal_ev", [])))


def spectrum64(Xs):
    S = O.scale_main(Xs)[0] if Xs is X else O.logn_scale(O.pre_scale(Xs))  # float64 (data: inline twin; null: closure path)
    t = time.perf_counter()
    G = O.wishart_matrix(S, 2 if N > M else 1)
    del S
    tg = time.perf_counter() - t
    t = time.perf_counter()
    w = sla.eigh(G, eigvals_only=True, driver="evd", overwrite_a=True, check_finite=False)
    return w, tg, time.perf_counter() - t


L64, tg, te = spectrum64(X)
log["host_gram_s"], log["host_eigvals_s"] = round(tg, 1), round(te, 1)
Lr64, _, _ = spectrum64(_r := api._csc_f32(X_r))
L_mp, _, b_min = O.mp_calculation(L64, Lr64[:-1])
lc64 = float(O.tw(L64, L_mp)[0])
k64 = int(np.sum(L64 > lc64))
lmax = float(L64[-1])
order = np.argsort(np.abs(L64 - lc64))[:6]
log.update({
    "lambda_max": lmax, "lambda_c_f64": lc64, "lambda_c_device": lc32, "lambda_c_rel_diff": abs(lc32 - lc64) / lc64,
    "k_f64": k64, "k_device": k32, "signal_count_identical": bool(k64 == k32),
    "max_abs_eig_diff_over_lambda_max": float(np.abs(L32 - L64).max() / lmax),
    "sqrt_n_eps32": float(np.sqrt(len(L64)) * 5.96e-8),
    "nearest_to_cut": [{"index": int(i), "L_f64": float(L64[i]), "L_device": float(L32[i]), "minus_cut_f64": float(L64[i] - lc64)} for i in sorted(order)],
    "gap_below_cut_over_lambda_max": float((lc64 - L64[L64 <= lc64].max()) / lmax),
    "gap_above_cut_over_lambda_max": float((L64[L64 > lc64].min() - lc64) / lmax),
    "guard_band": {"band_over_lambda_max": res["guard_band"]["band"] / lmax, "refined": res["guard_band"]["refined"]},
    "signal_ev_device": [float(v) for v in res.get("signal_ev", [])], "signal_ev_f64": [float(v) for v in L64[L64 > lc64][::-1]],
    "host_cores": os.cpu_count(),
})
out = json.dumps(log, indent=1)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
