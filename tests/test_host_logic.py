"""CPU: host-side product code (C++ statistics through the C ABI, draw generators, sharding logic) against the oracle."""
import os

import numpy as np
import pytest

from oracle import sclens_oracle as O
from sclens_amd import api
from sclens_amd.shard import consume_search_round, owned_perturbations, search_schedule
from sclens_amd.synth import synth_counts

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_cpp_statistics_match_golden_and_oracle():
    g = np.load(os.path.join(GOLD, "mp_known_answers.npz"))
    for name in ("white", "spiked"):
        L, Lr, exp = g[name + "_L"], g[name + "_Lr"], g[name + "_expect"]
        L_mp, bp, bm = api._mp_calculation(L, Lr[:-1])
        lam, gamma, p, sigma = api._tw(L, L_mp)
        chk = api.mp_check(L_mp)
        got = np.array([len(L_mp), bp, bm, lam, gamma, p, sigma, chk["ks_static"], float(chk["pass"]), float(np.sum(L > lam))])
        assert np.allclose(got, exp, rtol=1e-11)


def test_cpp_mp_calculation_empty_selection_runs_to_max_iter():
    # Appendix A12: a null spectrum whose edges select nothing -> NaN parameters -> loop ends at max_iter, empty L_mp
    L = np.linspace(5.0, 6.0, 50)
    Lr = np.linspace(0.1, 0.2, 49)
    a, _, _ = api._mp_calculation(L, Lr)
    b, _, _ = O.mp_calculation(L, Lr)
    assert len(a) == len(b) == 0


def test_cpp_robust_scores_match_oracle():
    rng = np.random.default_rng(1)
    k, P = 6, 20
    nV = np.linalg.qr(rng.standard_normal((200, k)))[0]
    nV_set = [np.linalg.qr(nV + 0.3 * rng.standard_normal((200, k)))[0] for _ in range(P)]
    rob = O.robustness(nV, nV_set)
    m, sd = api._robust_scores(rob["b_"])
    assert np.allclose(m, rob["rob_score"], rtol=1e-12)
    assert np.allclose(sd, rob["sd_scores"], rtol=1e-10)
    # heavy-tailed rows exercise the Tukey fence
    b = rng.random((4, 190))
    b[:, :5] = 0.0
    rows = []
    for s in range(4):
        row = b[s]
        q1, q3 = np.quantile(row, 0.25), np.quantile(row, 0.75)
        f = row[(q1 - 1.5 * (q3 - q1) <= row) & (row <= q3 + 1.5 * (q3 - q1))]
        rows.append(np.median(f))
    assert np.allclose(api._robust_scores(b)[0], rows)


def test_noise_baseline_exact_is_the_expectation_the_reference_samples():
    lib = api._lib.load()
    for n in (300, 5000):
        mc = O.noise_baseline(n, np.random.default_rng(0), trials=3000)
        ex = lib.sclens_noise_baseline_exact(n)
        assert abs(mc - ex) < 4 * 0.4 / np.sqrt(2 * np.log(n)) / np.sqrt(n) / np.sqrt(3000) + 1e-4


def test_draw_generators_contract():
    X = api._csc_f32(synth_counts(70, 110, seed=3, C=3))
    rng = np.random.default_rng(0)
    z1, z2 = api.draw_zero_candidates(X, rng)
    assert np.all(X[z1.astype(int), z2.astype(int)] == 0)
    assert len(np.unique(z1.astype(np.int64) + z2.astype(np.int64) * 70)) == len(z1)
    Xr = api.draw_null_matrix(X, rng)
    assert np.array_equal(np.diff(Xr.indptr), np.diff(X.indptr))
    assert np.array_equal(np.sort(Xr.data), np.sort(X.data))
    d = api.make_draws(X, seed=5, p_th_trials=50)
    a = d.sampler("search", 3, len(d.z_idx1), 40)
    assert len(np.unique(a)) == 40 and np.array_equal(a, d.sampler("search", 3, len(d.z_idx1), 40))


def test_sclens_refuses_cpu_and_maps_unknown_centering_onto_the_reference_fallback(capsys):
    """No CPU fallback inside the package. An unsupported `centering` string is NOT an error in the reference (scLENS.jl:655-657:
    a warning, then the mean branch's scaling in Float32): sclens() prints that warning and proceeds (here: up to the missing
    device); the per-call scaling drop-in has no such branch in the reference's API and keeps refusing."""
    X = synth_counts(60, 90, seed=2, C=3)
    with pytest.raises(NotImplementedError):
        api.sclens(X, device_="cpu")
    with pytest.raises(Exception) as ei:  # no GPU in the CPU suite: the call gets as far as creating the context
        api.sclens(X, centering="mode")
    assert not isinstance(ei.value, NotImplementedError)
    assert "not supported in the current algorithm" in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        api.logn_scale(X, centering="mode")


def _serial_search(d_list, p_th, p_step):
    """The reference's loop (scLENS.jl:725-761) over a pre-computed list of d5 results."""
    p_ = 0.999
    tank = np.zeros((5, 0))
    it = 0
    while True:
        tank = np.hstack([tank, d_list[it][:, None]])
        ppj = tank[1, :] if tank.shape[1] < 5 else tank[1, -5:]
        it += 1
        if (np.sum(ppj < p_th) > 4) or (p_ < 0.9):
            p_ += 4 * p_step
            break
        p_ -= p_step
    return p_, it


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_speculative_search_rounds_reproduce_the_serial_decisions(world):
    rng = np.random.default_rng(world)
    p_step, p_th = 0.001, 0.2
    d_list = [np.sort(rng.random(5) * (0.5 - 0.012 * i) + 0.05) for i in range(150)]
    p_ref, it_ref = _serial_search(d_list, p_th, p_step)
    p_list = search_schedule(p_step)
    tank, it, p_ = np.zeros((5, 0)), 0, None
    while p_ is None:
        results = [d_list[it + r] for r in range(world)]
        tank, used, stopped, p_fin = consume_search_round(tank, results, p_list, it, p_th, p_step)
        it += used
        if stopped:
            p_ = p_fin
    assert (p_, it) == (p_ref, it_ref)
    assert tank.shape[1] == it_ref


def test_ensemble_ownership_is_a_partition():
    for world in (1, 2, 4, 8):
        owned = [owned_perturbations(r, world, 20) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(20))
        assert max(map(len, owned)) == -(-20 // world)


def test_native_draw_generators_contract():
    X = api._csc_f32(synth_counts(90, 140, seed=4, C=3))
    d = api.make_draws_native(X, seed=3, host_sampler=True)
    z1, z2 = d.z_idx1, d.z_idx2
    assert 0 < len(z1) <= X.nnz
    assert np.all(X[z1.astype(int), z2.astype(int)] == 0)  # disjoint from the stored entries (setdiff, scLENS.jl:671)
    assert len(np.unique(z1.astype(np.int64) + z2.astype(np.int64) * 90)) == len(z1)  # unique
    # about nnz * sparsity candidates survive
    assert abs(len(z1) / X.nnz - (1 - X.nnz / (90 * 140))) < 0.1  # minus ~nnz/(2NM) duplicate draws
    Xr = d.X_r
    assert np.array_equal(np.diff(Xr.indptr), np.diff(X.indptr))
    assert np.array_equal(np.sort(Xr.data), np.sort(X.data))
    for j in range(140):
        rows = Xr.indices[Xr.indptr[j]: Xr.indptr[j + 1]]
        assert len(np.unique(rows)) == len(rows) and (len(rows) == 0 or rows.max() < 90)
    assert np.array_equal(api.make_draws_native(X, seed=3).z_idx1, z1)  # deterministic
    assert not np.array_equal(api.make_draws_native(X, seed=4).z_idx1[:50], z1[:50])


def test_keyed_permutation_sampler():
    for population, m in ((1000, 1000), (12345, 400), (17, 5), (1 << 20, 5000)):
        idx = api.sample_indices(population, m, seed=99)
        assert idx.max() < population and len(np.unique(idx)) == m  # distinct = without replacement
        assert np.array_equal(idx, api.sample_indices(population, m, seed=99))
        assert not np.array_equal(idx, api.sample_indices(population, m, seed=100))
    # uniformity: chi-square of 10 equal bins over many seeds
    counts = np.zeros(10)
    for s in range(200):
        idx = api.sample_indices(5000, 50, seed=s)
        counts += np.bincount(idx // 500, minlength=10)
    exp = counts.sum() / 10
    assert ((counts - exp) ** 2 / exp).sum() < 30  # chi2(9) 99.9 % quantile is 27.9
    assert api.sample_seed_for(1, "search", 0) != api.sample_seed_for(1, "perturb", 0)


def test_row_blocks_partition_the_cells():
    """atlas.row_block: contiguous, disjoint, covering, sizes within one of each other."""
    from sclens_amd import atlas

    for N, world in ((10, 3), (601, 2), (1000000, 8), (7, 7)):
        blocks = [atlas.row_block(r, world, N) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == N
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


def test_search_schedule_is_unbounded_and_matches_repeated_subtraction():
    """the schedule must not run out for small p_step or for speculative look-ahead (world x streams)."""
    for p_step in (0.001, 0.0002):
        sched = search_schedule(p_step)
        p, ref = 0.999, []
        for _ in range(700):
            ref.append(p)
            p -= p_step
        assert [sched[i] for i in range(700)] == ref  # bitwise the reference's accumulation (Appendix A21)
        assert sched[5] == ref[5]
    # a full serial search with p_step = 0.0002 that only stops on p_ < 0.9 needs ~496 iterations
    p_step, p_th = 0.0002, 0.0
    d_list = [np.full(5, 0.5)] * 600
    p_ref, it_ref = _serial_search(d_list, p_th, p_step)
    sched = search_schedule(p_step)
    tank, it, p_ = np.zeros((5, 0)), 0, None
    while p_ is None:
        tank, used, stopped, p_fin = consume_search_round(tank, [d_list[it + r] for r in range(6)], sched, it, p_th, p_step)
        it += used
        if stopped:
            p_ = p_fin
    assert (p_, it) == (p_ref, it_ref) and it_ref > 450


def test_host_side_under_address_and_ub_sanitizers():
    """SURVEY section 5 (race / memory checking of the native side): stats.cpp, rng.cpp and the host pattern builder are
    compiled with -fsanitize=address,undefined (`make asan`) and driven over their C entry points, error paths included."""
    import subprocess

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sclens_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(csrc, "host_selftest_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host selftest ok" in r.stdout


def test_guard_band_cut_stays_contiguous():
    """refined eigenvalues of a near-degenerate pair may swap; the cut must be taken from the top so that the
    k values returned belong to the k eigenvectors signal_vectors(k) returns."""
    from sclens_amd.api import cut_with_guard_band

    L = np.array([0.1, 0.5, 1.0, 1.00001, 1.00002, 3.0, 9.0])
    lam = 1.000012
    # no refinement requested
    Lo, k, nL, g = cut_with_guard_band(L, lam, 0.0, lambda lo, hi: 1 / 0)
    assert k == 3 and np.array_equal(nL, [9.0, 3.0, 1.00002]) and g["refined"] == []
    # band covers indices 2..4; the refined values of 3 and 4 swap sides of the threshold
    band_units = 2e-5 / (np.sqrt(len(L)) * 5.96e-8 * L[-1])
    calls = []

    def refine(lo, hi):
        calls.append((lo, hi))
        return np.array([1.0, 1.000015, 1.000011])[: hi - lo]

    Lo, k, nL, g = cut_with_guard_band(L, lam, band_units, refine)
    assert calls == [(2, 5)]
    assert k == 2 and np.array_equal(nL, [9.0, 3.0])  # index 4 fell below: the cut stops there although index 3 is above
    assert not g["monotone"] and np.all(np.diff(Lo) >= 0) and len(g["refined"]) == 3
    # monotone refinement: values are substituted in place
    Lo, k, nL, g = cut_with_guard_band(L, lam, band_units, lambda lo, hi: np.array([1.0, 1.000011, 1.000013]))
    assert k == 3 and g["monotone"] and nL[2] == 1.000013 and Lo[3] == 1.000011
    # everything above / below
    assert cut_with_guard_band(L, 0.0, 0.0, None)[1] == len(L)
    assert cut_with_guard_band(L, 100.0, 0.0, None)[1] == 0


def test_options_are_abi_not_environment():
    """Round 5: the library's tunables are named options of a context (csrc/common.h, SCL_OPTION_TABLE) reached
    through sclens_hip_set_option; the environment is read in a handful of places, once each. INTEGRATION.md section 5 documents every
    option and every SCLENS_HIP_* variable the sources mention, and nothing else."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, n_getenv = set(), 0
    for pat in ("sclens_amd/csrc/*.hip", "sclens_amd/csrc/*.h", "sclens_amd/csrc/*.cpp", "sclens_amd/*.py"):
        for f in glob.glob(os.path.join(root, pat)):
            text = open(f).read()
            src |= set(re.findall(r"SCLENS_HIP_[A-Z0-9_]+", text))
            if "/csrc/" in f:
                n_getenv += len(re.findall(r"\bgetenv\(", text))
    src.discard("SCLENS_HIP_COMM_ID_BYTES")  # a compile-time constant of the header
    assert n_getenv < 10, n_getenv
    doc_text = open(os.path.join(root, "INTEGRATION.md")).read()
    doc = set(re.findall(r"SCLENS_HIP_[A-Z0-9_]+", doc_text))
    assert src - doc == set(), f"undocumented variables: {sorted(src - doc)}"
    assert doc - src == set(), f"documented but not read anywhere: {sorted(doc - src)}"
    table = open(os.path.join(root, "sclens_amd/csrc/common.h")).read()
    table = table[table.index("#define SCL_OPTION_TABLE(X)"):table.index("struct Options {")]
    names = re.findall(r"X\((\w+),", table)
    assert len(names) > 30 and "precision" in names
    missing = [q for q in names if f"`{q}`" not in doc_text]
    assert not missing, f"options missing from INTEGRATION.md section 5: {missing}"


def test_null_matrix_draw_does_not_depend_on_the_thread_count(monkeypatch):
    """R2 (scLENS.jl:701 -> :261-289): the two-level shuffle of the stored values draws its buckets from a counter-based hash since round 4
    (the pass runs on all host threads); chunking by thread must not show in the result."""
    X = api._csc_f32(synth_counts(400, 700, seed=6, C=4))
    out = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SCLENS_HIP_HOST_THREADS", threads)
        Xr = api._resolve(api.make_draws_native(X, seed=21, host_sampler=True).X_r)
        out.append((Xr.indptr.copy(), Xr.indices.copy(), Xr.data.copy()))
    for o in out[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(o, out[0]))
    assert np.array_equal(np.sort(out[0][2]), np.sort(X.data)) and np.array_equal(np.diff(out[0][0]), np.diff(X.indptr))


def test_bench_line_fits_what_the_driver_reads():
    """round 4's stdout line grew to 25 KB (per-step decisions of 20 steps, job timelines, every stage) and the driver,
    which keeps the tail of stdout, could not parse it. `bench.compact_line` must stay under 4 KB for the driver's own `--steps 20
    --warmup 5`, parse as JSON, and still carry the contract's keys + roofline + cpu_baseline; the full record goes to a side file."""
    import json

    import bench

    steps = 20
    dec = [{"seed": 1000 + s, "signals": 7, "robust_signals": 7, "search_iters": 19, "p_": 0.985, "min_abs_margin": 2.3e-5 + 1e-6 * s,
            "d5_second_smallest": [0.0301 - 0.0003 * q for q in range(19)], "p_th": 0.024661, "wall_s": 28.4 + 0.01 * s,
            "phase_s": {"session_create": 0.06, "first_decompositions": 3.8, "sparsity_search": 20.7, "ensemble": 3.7, "scoring": 0.18},
            "first_phase_jobs_s": [["data_spectrum", 0.07, 1.35], ["null_spectrum", 0.93, 2.0], ["binary_basis", 2.0, 3.9]]} for s in range(steps)]
    stage = lambda ms, frac: {"bound": "mfma", "ms": ms, "achieved": 100.0, "peak": 157.3, "unit": "TFLOP/s", "frac": frac, "work": "w" * 150}
    full = {
        "metric": "sclens() cells*genes/s (wall-clock of one full sclens() call)", "value": 1.056e8, "unit": "cells*genes/s", "n_gpus": 1,
        "steps": steps, "warmup": 5, "steps_requested": steps, "warmup_requested": 5, "budget_s": 1500.0, "ms_per_step": 28410.0,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": bench.DTYPE_SPLIT, "data": "synthetic",
        "dtype_note": bench.DTYPE_NOTE,
        "config": {"workload": "cfg4: " + "x" * 200, "N": 100000, "M": 30000, "nnz": 310629691, "n_perturb": 20, "parallelism": "p" * 150,
                   "comm": "c" * 300, "precision": 1, "keep_warm": True},
        "sclens_wall_s": 28.41,
        "observed": {"signals": 7, "robust_signals": 7, "search_iters": 19, "p_": 0.985, "synth_s": 2.0, "decisions_per_step": dec,
                     "hbm_in_use_GB_after_timed_steps": 264.0, "search_job_s_last_step": [[q, 0, 1.0 * q, 1.0 * q + 1] for q in range(20)],
                     "first_phase_jobs_s_last_step": [["a", 0.0, 1.0]] * 8},
        "value_strict_fp32": 5.41e7, "strict_steps": 3, "strict_ms_per_step": 55400.0, "decisions_differ": False,
        "extra": {"strict_fp32": {"decisions": dec[-3:]}},
        "roofline": {"bound": "mfma", "kernel": "two-stage symmetric eigensolver " + "k" * 200, "achieved": 57.6, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": 0.366, "traffic": 2.296e12, "traffic_source": "t" * 500, "n": 30016, "vectors": 15008, "launch_ms": 1093.0,
                     "stage_ms": {k: 100.0 for k in ("sy2sb_dense_to_band", "sb2st_bulge_chasing", "stebz", "stein", "q2_back_transform", "q1_back_transform")},
                     "stages": {k: stage(100.0, 0.4) for k in ("normalise", "normalise_search_step", "gram", "gram_binary_f16", "sy2sb_dense_to_band",
                                                               "sb2st_bulge_chasing", "q2_back_transform", "q1_back_transform")}, "note": "n" * 400},
        "cpu_baseline": {"value": 72224.6, "unit": "cells*genes/s", "cores": 16, "kind": "port", "wall_s_extrapolated": 15913.2,
                         "wall_s_lower_bound": 41537.1, "samples": [{"Ns": 1, "Ms": 2}] * 2, "sample": "s" * 700, "lower_bound_note": "l" * 600},
        "bench_wall_s": 856.8}
    line = bench.compact_line(full, "/somewhere/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= 4096 and len(json.dumps(full)) > 15000
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "value_strict_fp32", "strict_steps", "decisions_differ"):
        assert key in back, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(back["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(back["cpu_baseline"])
    assert back["config"]["workload"].startswith("cfg4") and back["detail"] == "bench_detail.json"
    assert all(v <= 1.0 for v in back["roofline"]["stage_frac"].values())
    # twice the steps and absurdly long strings: the optional parts are dropped until it fits
    full["observed"]["decisions_per_step"] = dec * 4
    full["config"]["comm"] = "c" * 5000
    assert len(json.dumps(bench.compact_line(full, None))) <= 4096


def test_matching_certificate_bound():
    """The bound behind `ensemble_tail = "certified"` (session_robustness, api.sclens): for a unit vector v and an orthonormal basis
    Q, every unit vector u orthogonal to the first k columns satisfies |v'u| <= sqrt(1 - sum_{j<k} (v'q_j)^2) -- so if the best of the
    first k beats that bound, the argmax over ANY set of further orthonormal columns (exact tail eigenvectors or unconverged Ritz
    vectors alike) lies among the first k. Checked on random bases, for the remaining columns and for random unit combinations of them."""
    rng = np.random.default_rng(12)
    for n, k in ((40, 3), (120, 7), (300, 11)):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        for _ in range(20):
            v = Q[:, :k] @ rng.standard_normal(k) * rng.uniform(0.2, 3.0) + Q[:, k:] @ rng.standard_normal(n - k) * rng.uniform(0.0, 1.0)
            v /= np.linalg.norm(v)
            c = Q.T @ v
            bound = np.sqrt(max(0.0, 1.0 - float(np.sum(c[:k] ** 2))))
            assert np.all(np.abs(c[k:]) <= bound + 1e-12)
            for _ in range(5):
                u = Q[:, k:] @ rng.standard_normal(n - k)
                u /= np.linalg.norm(u)
                assert abs(float(v @ u)) <= bound + 1e-12
            if np.max(np.abs(c[:k])) ** 2 > 1.0 - float(np.sum(c[:k] ** 2)):
                assert int(np.argmax(np.abs(c))) < k
