"""Where the first phase of a cfg4 sclens() call goes: the three decompositions (null, data, binarised matrix) and the signal
vectors run ONE AT A TIME on one stream with the library's stage timers on; then the two placements of the three jobs on two
streams (what api.sclens does). Usage: first_phase_cfg4.py [N M]   (GPU box; reuses bench.py's cached matrix when present)"""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from sclens_amd import api
from sclens_amd._lib import Context
from sclens_amd.synth import synth_counts

N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 30000)
path = os.path.join(os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir()), f"sclens_bench_v2_cfg4_{N}x{M}_20240430.npz")
if os.path.exists(path):
    z = np.load(path)
    X = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(N, M))
else:
    X = synth_counts(N, M, seed=20240430)
X = api._csc_f32(X)
ctx = Context(0)
STAGES = ("scale", "gram", "sy2sb", "sb2st", "stebz", "stein", "sbr_q2", "sbr_q1", "sytrd", "ormtr", "refine", "recover", "corr")


def timed(name, f, c=ctx):
    c.sync()
    c.reset_timing()
    t = time.perf_counter()
    r = f()
    c.sync()
    dt = time.perf_counter() - t
    st = {}
    for s in STAGES:
        try:
            ms, calls = c.timing(s)
            if calls:
                st[s] = round(ms, 1)
        except Exception:
            pass
    print(f"{name}: {dt:.3f} s  stages(ms) {st}  sum {sum(st.values()) / 1e3:.3f} s", flush=True)
    return r


for rep in range(2):
    print(f"--- pass {rep} (0 = cold: first allocations)", flush=True)
    t0 = time.perf_counter()
    d = api.make_draws_native(X, seed=1000 + rep, async_null=True, device_candidates=True)
    print(f"make_draws_native: {time.perf_counter() - t0:.3f} s", flush=True)
    ctx.set_timing(True)
    ses = timed("session_create", lambda: api.Session(ctx, X))
    aux = Context(0)
    t0 = time.perf_counter()
    Xr = api._csc_f32(api._resolve(d.X_r))
    print(f"null matrix resolved on the host: {time.perf_counter() - t0:.3f} s", flush=True)
    npat = timed("null pattern (aux context)", lambda: api.Pattern(aux, Xr, [], []), aux)
    Lr = timed("null_spectrum_pattern", lambda: ses.null_spectrum_pattern(npat))
    L, _ = timed("data_spectrum", lambda: ses.data_spectrum(True))
    nV = timed("signal_vectors(8)", lambda: ses.signal_vectors(8))
    _, r = timed("binary_basis", lambda: ses.binary_basis())
    print("r_vr2", r, flush=True)
    ctx.set_timing(False)
    # two streams: {data | null -> binary} (api.sclens today) and {data -> null | binary}
    for label, plan in (("data | null+binary", ((0, "data"), (1, "null"), (1, "bin"))), ("data+null | binary", ((0, "data"), (0, "null"), (1, "bin")))):
        c2 = Context(0)
        w = ses.clone(c2)
        sess = [ses, w]
        fn = {"data": lambda s: s.data_spectrum(True), "null": lambda s: s.null_spectrum_pattern(npat), "bin": lambda s: s.binary_basis()}
        groups = {}
        for wk, job in plan:
            groups.setdefault(wk, []).append(job)

        def run(wk):
            for job in groups[wk]:
                fn[job](sess[wk])

        ctx.sync()
        t = time.perf_counter()
        with ThreadPoolExecutor(2) as ex:
            list(ex.map(run, list(groups)))
        ctx.sync()
        c2.sync()
        print(f"two streams, {label}: {time.perf_counter() - t:.3f} s", flush=True)
        w.close()
        c2.close()
    npat.close()
    aux.close()
    ses.close()
