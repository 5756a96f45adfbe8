#!/usr/bin/env python3
"""bench.py -- sclens() wall-clock and cells*genes/s on synthetic count matrices (BASELINE.json metric).

One "step" = one complete sclens() call (scLENS.jl:649-832: normalisation, data/null/binary decompositions,
MP/TW thresholding, sparsity search, 20-member perturbation ensemble, robustness scoring, gene basis) on a
seeded synthetic cells x genes count matrix that is already resident in host CSC form; every random draw of the
call is generated inside the timed region.

Default workload = the configuration BASELINE.json's `metric` is quoted on: 100 000 cells x 30 000 genes (`cfg4`).

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (a child `python -m
torch.distributed.run`, before anything here touches the GPU) and relays the child's JSON line and exit code.
Under torch.distributed.run (WORLD_SIZE set) it is one rank of the job.

A full sclens() at cfg4 takes about a minute, so K timed steps may not fit the time a caller allows: the run is held
to `--budget-s` seconds of wall-clock (default 1500 s, SCLENS_BENCH_BUDGET_S) and executes as many of the requested
warm-up / timed steps as fit. The JSON line reports the TRUE counts in `steps` / `warmup` and the requested ones in
`steps_requested` / `warmup_requested`.

Prints ONE JSON line on rank 0 (contract in the task statement), at most 4 KB (`compact_line`; everything else -- per-step decisions,
job timelines, every stage's rate, the CPU samples -- goes to `bench_detail.json` beside this file and to stderr), with two objects:
  roofline     : the STAGE that owns the wall clock, measured live with HIP events on the library's own stream.
                 n >= 8 192: the two-stage symmetric eigensolver of one sparsity-search step (dense -> band -> tridiagonal,
                 eigenvalues, inverse iteration, both back-transformations of n/2 vectors): achieved = the algorithmic
                 4/3 n^3 + 2 n^2 (n/2) flop of SURVEY 8(d) / the sum of its stage times, against 157.3 TF/s (fp32 MFMA) --
                 the time-weighted rate of everything that replaces `syevd!`, not of its best kernel. Otherwise `trd_colB`
                 (HBM-bound symmetric matrix-vector product of the one-stage tridiagonalisation) against 8 TB/s.
                 `stages` holds the same ratio for every stage of one decomposition, the Gram launch included.
  cpu_baseline : the oracle (float64 NumPy/SciPy port of the reference CPU path) timed on the host cores on two bounded
                 samples (the exponent of the eigensolver's cost is fitted, not assumed), stage-extrapolated to the workload.
The timed steps run with the context option precision = 0 (default since round 6): every dense product on the fp32 matrix cores, the
arithmetic of the reference's own GPU path (scLENS.jl:335-343, :377) -- `dtype` "f32".
  value_split_f16 : the same metric over `split_steps` further steps with precision = 1 (the large products from operands split into two
                 fp16 pieces, 22 bits), on the draws of those TIMED steps whose search statistic came closest to its threshold, so that
                 both arithmetic variants are timed by the same run; `decisions_differ`: "k of m" = how many of those m steps ended
                 with another (signals, robust signals, search length, p_) than their fp32 twin.
(`--precision 1` swaps the roles: the split variant is timed, `value_strict_fp32` reports the fp32 one.)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {  # name -> (N cells, M genes, index in BASELINE.json configs)
    "tiny": (600, 900, 0),
    "tiny_gt": (900, 400, 0),  # cells > genes, for --row-shard smoke runs
    "rs20k": (20000, 6000, 0),  # cells > genes at the order of the row-sharded GPU tests (tests/test_gpu_multirank.py)
    "cfg2": (10000, 20000, 1),
    "cfg3": (50000, 30000, 2),
    "cfg4": (100000, 30000, 3),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFS = 2500.0  # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
# HBM bytes per trd_colB launch / algorithmic bytes: constant from profiles/r01_pmc_final (rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- NOT measured in this run
PMC_TRAFFIC_RATIO_R01 = 1.07


def symv_probe(ctx, n):
    """Every trd_colB launch of one tridiagonalisation of order n (same grids / arguments as the real reduction), back
    to back on the library's stream between one pair of HIP events (sclens_hip_symv_probe)."""
    import ctypes as C

    launches, ms, nbytes = C.c_int64(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.sclens_hip_symv_probe(ctx.h, n, C.byref(launches), C.byref(ms), C.byref(nbytes)))  # warm-up
    ctx.check(ctx.lib.sclens_hip_symv_probe(ctx.h, n, C.byref(launches), C.byref(ms), C.byref(nbytes)))
    gbs = nbytes.value / (ms.value * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "trd_colB", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": round(PMC_TRAFFIC_RATIO_R01 * nbytes.value / max(1, launches.value), 1),
            "traffic_source": "constant 1.07 x algorithmic from profiles/r01_pmc_final, not measured in this run",
            "launches": launches.value, "avg_launch_us": round(ms.value * 1e3 / max(1, launches.value), 2),
            "algorithmic_bytes_per_launch_avg": round(nbytes.value / max(1, launches.value), 1), "n": n,
            "achieved_vs_full_read_4n2": round(2 * gbs, 1)}


def stage_probe(ctx, X, N, M):
    split_products = ctx.get_option("precision") != 0
    """One decomposition of the data matrix with per-stage HIP-event timing on the library's stream (normalise, Gram,
    eigensolver stages, eigenvectors of the lower half + the corr product as in one search step): achieved rate of each
    stage against the roofline that bounds it."""
    from sclens_amd import api

    n, K = min(N, M), max(N, M)
    ctx.set_timing(True)
    ctx.reset_timing()
    ses = api.Session(ctx, X)
    bits_ms, extra, search_scale = None, {}, None
    try:
        ses.data_spectrum(False)
        if N > M:  # the Gram product of the binarised matrix as the sparsity search forms it (fp16 MFMA for large problems)
            snap = lambda: {s: ctx.timing(s) for s in ("scale", "gram", "sytrd", "sy2sb", "sb2st", "stebz", "stein", "ormtr", "sbr_q2", "sbr_q1")}
            before, gb0 = snap(), ses.get_int("gram_bits_used")
            _, r_vr2 = ses.binary_basis()
            after = snap()
            extra = {s: (after[s][0] - before[s][0], after[s][1] - before[s][1]) for s in after}  # kept out of the averages below
            if ses.get_int("gram_bits_used") > gb0:
                bits_ms = extra["gram"][0]
            # one evaluation of the sparsity search on the union pattern (counts + zero candidates): its normalisation runs on the
            # pattern's CSR companion copies, which the counts-only patterns of the three first decompositions do not carry
            pat = api.Pattern.drawn(ctx, X, 12345)
            try:
                ses.set_pattern(pat)
                b2 = snap()
                ses.search_step_seeded(777, int(round(0.01 * N * M)), int(round(r_vr2 / 2)))
                a2 = snap()
                step = {s: (a2[s][0] - b2[s][0], a2[s][1] - b2[s][1]) for s in a2}
                search_scale = step["scale"]
                extra = {s: (extra[s][0] + step[s][0], extra[s][1] + step[s][1]) for s in extra}
            finally:
                ses.close()
                pat.close()
    finally:
        ses.close()
    if os.environ.get("SCLENS_BENCH_PROBE_VECTORS", "1") != "0":
        # eigenvectors of the lower half of a spectrum, as one search step computes them (Gram of a random n x 2048 block)
        rup = lambda x, q: (x + q - 1) // q * q
        lda, Kp = rup(n, 32), 2048
        rng = np.random.default_rng(5)
        blk = rng.standard_normal((min(n, 2048), Kp)).astype(np.float32)
        dB, dA = ctx.malloc(4 * n * Kp), ctx.malloc(4 * n * lda)
        dw, dZ = ctx.malloc(8 * n), ctx.malloc(4 * (n // 2) * lda)
        try:
            for r0 in range(0, n, blk.shape[0]):
                rows = min(blk.shape[0], n - r0)
                ctx.h2d(dB + 4 * r0 * Kp, np.roll(blk[:rows], r0 // blk.shape[0], axis=1))
            g0 = ctx.timing("gram")
            ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB, n, Kp, Kp, float(Kp), dA, lda))
            ctx.check(ctx.lib.sclens_hip_dev_eigh_f32(ctx.h, dA, n, lda, dw, 0, n // 2, dZ, lda))
            ctx.sync()
        finally:
            for q in (dB, dA, dw, dZ):
                ctx.free(q)
    names = ("scale", "gram", "sytrd", "sy2sb", "sb2st", "stebz", "stein", "ormtr", "sbr_q2", "sbr_q1")
    t = {s: ctx.timing(s) for s in names}
    t = {s: (t[s][0] - extra.get(s, (0.0, 0))[0], t[s][1] - extra.get(s, (0.0, 0))[1]) for s in names}
    if "g0" in locals():  # the Gram of the data matrix only (the probe's small product is not the workload's)
        t["gram"] = (g0[0] - extra.get("gram", (0.0, 0))[0], g0[1] - extra.get("gram", (0.0, 0))[1])
    ctx.set_timing(False)
    stages = {}

    def add(name, key, work, unit, peak, bound, note):
        ms, calls = t[key]
        if calls > 0 and ms > 0:
            per = ms / calls  # average duration of one call of the stage
            ach = work / (per * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
            stages[name] = {"bound": bound, "ms": round(per, 3), "achieved": round(ach, 2), "peak": peak, "unit": unit,
                            "frac": round(ach / peak, 4), "work": note}

    nnz = int(X.nnz)
    add("normalise", "scale", 8.0 * nnz + 4.0 * N * M, "GB/s", HBM_PEAK_GBS, "hbm", "8 nnz read + 4 N M written (SURVEY 8d B_norm)")
    if search_scale and search_scale[1] > 0:
        per = search_scale[0] / search_scale[1]
        ach = (8.0 * nnz + 4.0 * N * M) / (per * 1e-3) / 1e9
        stages["normalise_search_step"] = {"bound": "hbm", "ms": round(per, 3), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(ach / HBM_PEAK_GBS, 4),
                                           "work": "the same bytes for one evaluation of the sparsity search (binarised values + sampled "
                                                   "candidates on the union pattern, row reductions streamed from the CSR companion copy)"}
    gs_min = ctx.get_option("gram_split_min_n")
    if split_products and gs_min > 0 and n >= gs_min and ctx.get_option("gram_bits") != 0:
        # the data matrix's Gram product runs on the fp16 MFMA from split operands: three matrix instructions per product
        ms, calls = t["gram"]
        if calls > 0 and ms > 0:
            per = ms / calls
            issued = 3.0 * float(n) * (n + 1) * K / (per * 1e-3) / 1e12
            stages["gram"] = {"bound": "mfma", "ms": round(per, 3), "achieved": round(issued, 1), "peak": MFMA_F16_PEAK_TFS, "unit": "TFLOP/s",
                              "frac": round(issued / MFMA_F16_PEAK_TFS, 4),
                              "work": "3 n (n+1) K flop issued on the fp16 MFMA (operands split into two fp16 pieces: ah bh + ah bl + al bh, "
                                      "fp32 accumulation) for the n (n+1) K flop of the fp32 product (computed lower half)",
                              "fp32_equivalent_TFLOPs": round(issued / 3.0, 1)}
    else:
        add("gram", "gram", float(n) * (n + 1) * K, "TFLOP/s", MFMA_F32_PEAK_TFS, "mfma", "n (n+1) K flop (computed lower half)")
    if bits_ms:
        nw = 3 if (not split_products or ctx.get_option("gram_bits_terms") == 3) else 2  # fp16 pieces of the cell weights (22 / 33 bits)
        work = float(nw) * float(n) * (n + 1) * K  # one MFMA product per piece, gene pair and cell
        ach = work / (bits_ms * 1e-3) / 1e12
        stages["gram_binary_f16"] = {"bound": "mfma", "ms": round(bits_ms, 3), "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TFS,
                                     "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFS, 4),
                                     "work": f"{nw} n (n+1) K flop issued on the fp16 MFMA (exact 0/1 pattern x {11 * nw}-bit cell weights, fp32 "
                                             "accumulation) for the n (n+1) K flop of the fp32 product it replaces in the sparsity search",
                                     "fp32_equivalent_TFLOPs": round(ach / nw, 1)}
    add("sytrd_one_stage", "sytrd", sum(2.0 * q * (q + 1) for q in range(1, n)), "GB/s", HBM_PEAK_GBS, "hbm",
        "lower triangle of the trailing matrix once per column, whole reduction")
    add("sy2sb_dense_to_band", "sy2sb", 4.0 / 3.0 * float(n) ** 3, "TFLOP/s", MFMA_F32_PEAK_TFS, "mfma",
        "4/3 n^3 flop, fp32-equivalent (its updates and W = A22 V issue 3x that on the fp16 MFMA when precision = 1; the panel algebra is fp64)")
    add("sb2st_bulge_chasing", "sb2st", 6.0 * float(n) ** 2 * 64, "TFLOP/s", MFMA_F32_PEAK_TFS, "latency",
        "6 n^2 b flop, b = 64 (latency-bound chain of 2 n dependent steps; rate shown for scale only)")
    m = n // 2
    if split_products:
        # both back-transformations issue THREE fp16 matrix instructions per fp32-equivalent product (hi hi + hi lo + lo hi): the rate
        # that compares with a peak is the issued one against the fp16 peak (a fp32-equivalent rate over the fp32 peak can exceed 1)
        for name, key in (("q2_back_transform", "sbr_q2"), ("q1_back_transform", "sbr_q1")):
            add(name, key, 3.0 * 2.0 * float(n) ** 2 * m, "TFLOP/s", MFMA_F16_PEAK_TFS, "mfma",
                "3 x 2 n^2 m flop issued on the fp16 MFMA (split operands), m = n/2")
            if name in stages:
                stages[name]["fp32_equivalent_TFLOPs"] = round(stages[name]["achieved"] / 3.0, 1)
    else:
        add("q2_back_transform", "sbr_q2", 2.0 * float(n) ** 2 * m, "TFLOP/s", MFMA_F32_PEAK_TFS, "mfma", "2 n^2 m flop, m = n/2")
        add("q1_back_transform", "sbr_q1", 2.0 * float(n) ** 2 * m, "TFLOP/s", MFMA_F32_PEAK_TFS, "mfma", "2 n^2 m flop, m = n/2")
    add("ormtr_back_transform", "ormtr", 2.0 * float(n) ** 2 * m, "TFLOP/s", MFMA_F32_PEAK_TFS, "mfma", "2 n^2 m flop, m = n/2")
    for key in ("stebz", "stein"):
        ms, calls = t[key]
        if calls:
            stages[key] = {"bound": "latency", "ms": round(ms / calls, 3)}
    return stages


# HBM traffic of ONE two-stage eigensolve of order 30 016 with 15 008 vectors (the roofline's launch), from separate `rocprofv3 --pmc
# FETCH_SIZE` / `--pmc WRITE_SIZE` passes over scripts/perf_eig.py (counters collected for the kernels that move the bytes,
# --kernel-include-regex), FETCH_SIZE doubled on the 16-byte-per-lane operand streams of the 256 x 256 kernels as MI355X_MICROARCH.md (HBM)
# prescribes for gfx950. precision = 0 (round-6 build, profiles/r06_pmc_eig_fp32/summary.txt): 1006 GB fetched as counted + 504 GB for the
# half-count + 868 GB written = 2.38e12 B = 1.17 x the algorithmic bytes (the fp32 Q2 kernel takes the vectors down the matrix in passes
# of four blocks). precision = 1 (round-5 build, profiles/r05_pmc_eig/summary.txt): 2.20e12 B = 1.02 x. Constants measured on the build at
# this size, not in this run (a PMC pass serialises every profiled dispatch); the line carries them only for the workload they were
# measured on.
PMC_EIG_TRAFFIC = {1: {"bytes": 2.196e12, "n": 30016, "vectors": 15008, "algorithmic_bytes": 2.15e12, "source": "profiles/r05_pmc_eig/summary.txt"},
                   0: {"bytes": 2.378e12, "n": 30016, "vectors": 15008, "algorithmic_bytes": 2.04e12, "source": "profiles/r06_pmc_eig_fp32/summary.txt"}}


# A full-size CPU data point kept in the repository (profiles/r02_signal_count_cfg4.json, GPU box, 16 usable CPUs): LAPACK dsyevd,
# VALUES ONLY, of ONE float64 30 000 x 30 000 Gram matrix took 903.1 s, the float64 Gram product (dsyrk) 110.0 s. The reference
# computes all eigenVECTORS of 3 + S + P such matrices (scLENS.jl:384), which costs more than values only.
CPU_FULL_SIZE_POINT = {"n": 30000, "dsyevd_values_only_s": 903.1, "dsyrk_100000x30000_s": 110.0, "cores": 16,
                       "source": "profiles/r02_signal_count_cfg4.json"}


def cpu_baseline(N, M, n_search, n_perturb, budget_s):
    """Oracle (port of the reference CPU path) on the host cores: normalise, Gram and dsyevr (all vectors) timed at TWO sample
    orders; the eigensolver's cost exponent is fitted from the two (clamped to [2.5, 3.2]) instead of assuming n^3, the other
    stages scale exactly (N M, n^2 K). Everything is multiplied by the call counts the GPU run observed."""
    from oracle import sclens_oracle as O  # checker / baseline only
    from sclens_amd.synth import synth_counts

    n, K = min(N, M), max(N, M)
    cores = usable_cpus()
    try:  # BLAS / LAPACK threads = the CPUs the process may use (a 256-thread pool on a 16-CPU quota only thrashes)
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=cores)
    except Exception:
        pass
    # dsyevr with vectors: ~7 s at n = 4000 on a 16-CPU quota. Larger sample: its n^3 estimate fits ~45 % of the budget;
    # smaller sample: 0.6 x that order (0.22 of the time)
    est = lambda q: 1.1e-10 * q ** 3 * max(1.0, 16.0 / cores) + 0.5
    n_hi = n
    while n_hi > 1500 and est(n_hi) > 0.45 * budget_s:
        n_hi = int(n_hi * 0.9)
    n_hi = min(n, max(1500, n_hi))
    orders = [n_hi] if n_hi == n else [max(1000, int(0.6 * n_hi)), n_hi]
    samples = []
    for ns in orders:
        Ns, Ms = (ns, int(ns * M / N)) if N <= M else (int(ns * N / M), ns)
        if N > M:  # keep the sample's contraction length affordable: the Gram scales exactly with K
            Ns = min(Ns, 4 * ns)
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
        samples.append({"Ns": Ns, "Ms": Ms, "n": min(Ns, Ms), "scale_s": round(t_scale, 3), "gram_s": round(t_gram, 3),
                        "dsyevr_s": round(t_eig, 3)})
    hi = samples[-1]
    if len(samples) == 2 and samples[0]["dsyevr_s"] > 0.05:
        p_fit = float(np.log(hi["dsyevr_s"] / samples[0]["dsyevr_s"]) / np.log(hi["n"] / samples[0]["n"]))
    else:
        p_fit = 3.0
    p_use = min(3.2, max(2.5, p_fit))
    calls = 3 + n_search + n_perturb
    f_eig = (n / hi["n"]) ** p_use
    t_eig_full = hi["dsyevr_s"] * f_eig
    T = calls * (hi["scale_s"] * (N * M) / (hi["Ns"] * hi["Ms"]) + hi["gram_s"] * (n * n * K) / (hi["n"] ** 2 * max(hi["Ns"], hi["Ms"])) + t_eig_full)
    T += n_search * hi["gram_s"] * (n ** 3) / (hi["n"] ** 2 * max(hi["Ns"], hi["Ms"]))  # corr_mat (scLENS.jl:742): ~n^3 flop per iteration
    out = {"value": round(N * M / T, 1), "unit": "cells*genes/s", "cores": cores, "kind": "port",
           "wall_s_extrapolated": round(T, 1), "eig_exponent_fitted": round(p_fit, 3), "eig_exponent_used": round(p_use, 3),
           "eig_all_vectors_s_extrapolated": round(t_eig_full, 1), "samples": samples,
           "sample": (f"oracle normalise+Gram+dsyevr (all vectors) timed at {' and '.join(str(q['Ns']) + 'x' + str(q['Ms']) for q in samples)} "
                      f"(dsyevr {', '.join(str(q['dsyevr_s']) + ' s' for q in samples)}: exponent {p_fit:.2f} fitted, {p_use:.2f} used); "
                      f"scaled by NM, n^2K and n^{p_use:.2f} (x{f_eig:.1f}) to {N}x{M}, times {calls} decompositions (S={n_search}, "
                      f"P={n_perturb}) + {n_search} corr GEMMs; BLAS threads = {cores} = the CPUs this process may use "
                      f"(cgroup quota; {os.cpu_count()} visible)")}
    if n == CPU_FULL_SIZE_POINT["n"]:
        # the one measured full-size number bounds the wall clock from below: every one of the `calls` decompositions costs at least a
        # values-only dsyevd (the reference computes all vectors too, :384) plus its Gram product
        lb = calls * (CPU_FULL_SIZE_POINT["dsyevd_values_only_s"] + CPU_FULL_SIZE_POINT["dsyrk_100000x30000_s"] * K / 100000.0)
        out["wall_s_lower_bound"] = round(lb, 1)
        out["value_upper_bound"] = round(N * M / lb, 1)
        if lb > T:  # the measured point contradicts the extrapolation: the baseline quoted is the one the measurement supports
            out["value_two_point_extrapolation"] = out["value"]
            out["value"] = out["value_upper_bound"]
        out["lower_bound_note"] = (f"{calls} decompositions x (dsyevd values only {CPU_FULL_SIZE_POINT['dsyevd_values_only_s']} s + dsyrk "
                                   f"{CPU_FULL_SIZE_POINT['dsyrk_100000x30000_s'] * K / 100000.0:.0f} s), both measured at full size on "
                                   f"{CPU_FULL_SIZE_POINT['cores']} cores ({CPU_FULL_SIZE_POINT['source']}); the two-sample extrapolation above "
                                   "(`wall_s_extrapolated`) is BELOW this bound whenever its fitted exponent is under 3 -- quote the bound")
        out["full_size_point"] = dict(CPU_FULL_SIZE_POINT, note=(
            f"measured once at full size: dsyevd VALUES ONLY {CPU_FULL_SIZE_POINT['dsyevd_values_only_s']} s per matrix against "
            f"{t_eig_full:.0f} s extrapolated here for dsyevr with ALL vectors; {calls} such decompositions per call"))
    return out


DTYPE_SPLIT = "f32 (large products: operands as 2 x f16 pieces = 22 bit, f32 accumulate)"
DTYPE_NOTE_F32 = ("context option precision = 0: fp32 data, every dense product (Gram, search statistic, band reduction, both back-transformations) on "
                  "the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32: 24-bit operands, fp32 accumulate -- the arithmetic of the "
                  "reference's cuBLAS SGEMM / cuSOLVER ssyevd path, scLENS.jl:335-343, :377), fp64 statistics / eigenvalues / panel algebra. The "
                  "Gram matrix of a BINARISED matrix (sparsity search, scLENS.jl:735-738) is formed as the exact co-occurrence product of its 0/1 "
                  "pattern (exact in fp16) with the cell weights as three fp16 pieces = 33 bits, fp32 accumulate, rank-one terms in fp64: no "
                  "operand narrower than fp32 (context option gram_bits_strict = 0 sends it through the fp32 product as well). value_split_f16 = "
                  "the same call with precision = 1 (large products from operands split into two fp16 pieces, 22 bits)")
DTYPE_NOTE = ("fp32 data and accumulation, fp64 statistics / eigenvalues / panel algebra. Context option precision = 1 (default): the large "
              "products run on the fp16 MFMA from operands split into two fp16 pieces (22 significant bits, three matrix instructions per "
              "product; the Gram matrix of a binarised matrix as an exact 0/1 x 22-bit-weight product) -- Gram products from n = 16 000, the "
              "search statistic, trailing updates and W = A22 V of the band reduction, both back-transformations. value_strict_fp32 = the "
              "same call with precision = 0: every product on the fp32 MFMA")

LINE_LIMIT = 4096  # bytes of the ONE JSON line on stdout (the driver keeps the tail of stdout; round 4's 25 KB line was cut and never parsed)


def compact_line(full, detail_path=None):
    """The stdout line: the contract's keys + value_strict_fp32 / strict_steps / decisions_differ + a trimmed roofline and cpu_baseline;
    every list, timeline and note stays in the detail file. Never longer than LINE_LIMIT bytes (tests/test_host_logic.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "steps_requested", "warmup_requested", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "value_strict_fp32", "strict_steps", "strict_ms_per_step", "value_split_f16", "split_steps",
            "split_ms_per_step", "decisions_differ", "bench_wall_s")
    line = {k: full[k] for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "N", "M", "n_perturb", "parallelism", "comm", "precision", "keep_warm", "ensemble_tail") if k in cfg}
    for k in ("workload", "parallelism", "comm"):
        if isinstance(line["config"].get(k), str) and len(line["config"][k]) > 220:
            line["config"][k] = line["config"][k][:217] + "..."
    obs = full.get("observed", {})
    line["observed"] = {k: obs[k] for k in ("signals", "robust_signals", "search_iters", "p_", "hbm_in_use_GB_after_timed_steps", "hbm_peak_live_GB") if k in obs}
    dec = obs.get("decisions_per_step") or []
    if dec:
        line["observed"]["wall_s_per_step"] = [d.get("wall_s") for d in dec][:40]
        margins = [d["min_abs_margin"] for d in dec if d.get("min_abs_margin") is not None]
        if margins:
            line["observed"]["min_abs_margin_over_steps"] = min(margins)
    rf = full.get("roofline")
    if rf:
        line["roofline"] = {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "n", "vectors", "launch_ms",
                                               "launches", "avg_launch_us", "stage_ms") if k in rf}
        if isinstance(line["roofline"].get("kernel"), str) and len(line["roofline"]["kernel"]) > 160:
            line["roofline"]["kernel"] = line["roofline"]["kernel"][:157] + "..."
        st = rf.get("stages") or {}
        line["roofline"]["stage_frac"] = {k: v["frac"] for k, v in st.items() if isinstance(v, dict) and "frac" in v}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "wall_s_lower_bound", "wall_s_extrapolated") if k in cb}
        line["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:300]
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    # last resort: drop the optional parts, longest first, until the line fits
    for victim in (("observed", "wall_s_per_step"), ("roofline", "stage_frac"), ("roofline", "stage_ms"), ("cpu_baseline", "sample"),
                   ("config", "comm"), ("config", "parallelism")):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.get(victim[0], {}).pop(victim[1], None)
    assert len(json.dumps(line)) <= LINE_LIMIT, len(json.dumps(line))
    return line


def usable_cpus():
    """CPUs this process may really use: affinity and cgroup quota (the GPU box shows 256 CPUs with a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return max(1, n)


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child job. Nothing in this process has touched
    the GPU yet (no torch import, no library context), and the child is a subprocess, not an exec."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", default="cfg4", choices=list(CONFIGS))
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("SCLENS_BENCH_BUDGET_S", "1500")),
                    help="wall-clock limit of the whole run; warm-up and timed steps are cut to fit (true counts are reported)")
    ap.add_argument("--n-perturb", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--row-shard", action="store_true",
                    help="cells > genes configs on N > 1 GPUs: every rank holds a block of cells, partial Gram matrices are "
                         "all-reduced (SURVEY 8e-iii, sclens_amd/atlas.py) instead of distributing whole decompositions")
    ap.add_argument("--streams", type=int, default=None,
                    help="concurrent decompositions per GPU (worker sessions on own HIP streams); default: 3 below n = 16 000, else 2")
    ap.add_argument("--strict-fp32", "--other-variant", dest="strict_fp32", default="auto", choices=["auto", "on", "off"],
                    help="further timed steps with the OTHER arithmetic variant (precision = 1 - the timed one), reported as value_split_f16 / "
                         "value_strict_fp32 (auto: for n >= 16 000)")
    ap.add_argument("--strict-steps", "--other-steps", dest="strict_steps", type=int, default=5,
                    help="how many of them (on the draws of the timed steps with the smallest margin of the search statistic)")
    ap.add_argument("--precision", type=int, default=0, choices=[0, 1],
                    help="context option precision of the timed steps (0: every dense product on the fp32 MFMA, the reference GPU path's "
                         "arithmetic; 1: split-fp16 products)")
    ap.add_argument("--extra-configs", default="", help="comma-separated further configs timed once each after the main one (reported under `extra`)")
    ap.add_argument("--seed-base", type=int, default=1000, help="timed step s draws with seed seed_base + s (warm-up steps: seed_base - 1 - w)")
    ap.add_argument("--ballast-gb", type=float, default=0.0,
                    help="diagnostic: hold this much extra device memory for the whole run (how close to a full HBM may the footprint come "
                         "before kernels slow down? DESIGN.md section 5)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--stage-timing", action="store_true", help="per-stage HIP-event totals on stderr (adds syncs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

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
        # torch.distributed is the launcher's rendezvous and the channel that ships ONE 128-byte id: always gloo (host side). With
        # --backend nccl the collectives of the path are RCCL calls made by the library on its own communicator (csrc/comm.hip);
        # torch never creates a second RCCL communicator beside it (round 3 held two per rank, a combination that had only ever
        # run with one rank).
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from sclens_amd import api
    from sclens_amd._lib import Context
    from sclens_amd.shard import Shard
    from sclens_amd.synth import synth_counts

    ballast = torch.empty(int(args.ballast_gb * 1e9), dtype=torch.uint8, device=dev) if args.ballast_gb > 0 else None  # noqa: F841
    ctx = Context(local_rank)
    ctx.set_option("precision", args.precision)
    # the library's own RCCL communicator (csrc/comm.hip); torch.distributed only launched the ranks and ships the unique id
    shard = Shard.create(ctx, rank, world, backend=args.backend)
    check = shard.selfcheck(ctx)  # raises (non-zero exit) when the all-reduce / broadcast on library buffers is wrong
    if rank == 0 and world > 1:
        print(f"[bench] {world} ranks, backend {args.backend}, self-check {check}", file=sys.stderr, flush=True)

    def agree(x):  # rank 0's decision on every rank (steps must match or the collectives dead-lock)
        return float(shard.bcast_host(np.array([float(x)]), 0)[0])

    def cached_counts(cfg, N, M, seed):
        """The seeded synthetic matrix, kept as an .npz in SCLENS_BENCH_CACHE (default: the temp dir; "0" disables) so that
        back-to-back runs on one box (N = 1, 2, 4, 8) and the ranks of one run do not each spend a minute regenerating it."""
        import scipy.sparse as sp
        import tempfile

        d = os.environ.get("SCLENS_BENCH_CACHE", tempfile.gettempdir())
        path = os.path.join(d, f"sclens_bench_v2_{cfg}_{N}x{M}_{seed}.npz") if d not in ("", "0") else None
        if path and os.path.exists(path):
            try:
                z = np.load(path)
                return sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(N, M))
            except Exception as e:  # truncated / foreign file: regenerate
                print(f"[bench] ignoring cache {path}: {e}", file=sys.stderr)
        X = synth_counts(N, M, seed=seed)
        if path and local_rank == 0:  # one writer per box; the other ranks of a first run generate their own copy
            try:
                tmp = f"{path}.{os.getpid()}.tmp.npz"
                np.savez(tmp, data=X.data, indices=X.indices, indptr=X.indptr)
                os.replace(tmp, path)
            except OSError as e:
                print(f"[bench] synthetic matrix not cached ({e})", file=sys.stderr)
        return X

    def decisions_of(res, seed):
        """what one sclens() call decided (scLENS.jl:742-760 stop rule, :541 cut): the numbers two arithmetic variants must share"""
        tr = res.get("search_trace", [])
        d2 = [float(d5[1]) for _, d5 in tr]  # the second smallest search statistic of every evaluation (the `ppj_` row, :753)
        p_th = float(res["p_th"])
        return {"seed": int(seed), "signals": int(len(res.get("signal_ev", []))), "robust_signals": int(len(res.get("sig_id", []))),
                "search_iters": int(res["n_search"]), "p_": res["p_"],
                "min_abs_margin": (round(min(abs(x - p_th) for x in d2), 6) if d2 else None),
                "d5_second_smallest": [round(x, 5) for x in d2], "p_th": round(p_th, 6),
                "phase_s": {k: round(float(v), 3) for k, v in (res.get("phase_s") or {}).items()},
                "ensemble_tail": res.get("ensemble_tail"), "members_solved_again": [int(t) for t in res.get("tail_redo", [])],
                "first_phase_jobs_s": [list(q) for q in res.get("first_phase_s", [])]}

    def run_config(cfg, steps_req, warm_req, deadline, tail_steps=0.0, step0=0, step_list=None):
        """tail_steps: keep this many step durations of the budget free for what follows (the steps of the other arithmetic variant);
        step0: index of the first timed step (its draws are seeded 1000 + step0); step_list: the step indices to run instead of
        step0, step0 + 1, ..."""
        N, M, cfg_index = CONFIGS[cfg]
        t0 = time.perf_counter()
        X = api._csc_f32(cached_counts(cfg, N, M, 20240427 + cfg_index))  # PCG64(20240427 + config_index)
        t_synth = time.perf_counter() - t0
        row_shard = args.row_shard and N > M
        if row_shard:
            from sclens_amd import atlas

            r0, r1 = atlas.row_block(rank, world, N)
            X_rows = api._csc_f32(X.tocsr()[r0:r1].tocsc())

        def one_step(step):
            t_d = time.perf_counter()
            draws = api.make_draws_native(X, seed=args.seed_base + step, async_null=True, device_candidates=True)
            one_step.draws_s = time.perf_counter() - t_d  # R1-R3 inside the timed region; R4/R5 on the device inside sclens()
            if row_shard:  # local cells, local candidates (each rank's part of the global draw), eigensolves of a round on different ranks
                return atlas.sclens_row_sharded(X_rows, r0, N, draws, shard, n_perturb=args.n_perturb, ctx=ctx, gather=False,
                                                verbose=args.verbose, nnz_global=int(X.nnz))
            return api.sclens(X, draws=draws, ctx=ctx, n_perturb=args.n_perturb, shard=shard, streams=args.streams,
                              verbose=args.verbose and rank == 0, keep_warm=True)  # a loop of calls: the pool keeps the blocks between them

        def fence():
            shard.barrier()
            torch.cuda.synchronize()
            ctx.sync()

        res, t_step, n_warm = None, None, 0
        t_warm0 = time.perf_counter()
        for w in range(warm_req):
            # a further warm-up step is taken only while warm-up stays below 15 % of the budget and it plus one timed step
            # still fit (the first one is never skipped): at ~80 s per step the budget goes to timed steps
            if agree(t_step is not None and (time.perf_counter() + 2.2 * t_step > deadline or
                                             time.perf_counter() - t_warm0 + t_step > 0.15 * args.budget_s)):
                break
            ts = time.perf_counter()
            res = one_step(-1 - w)
            t_step = time.perf_counter() - ts
            n_warm += 1
        fence()
        n_steps, decisions, phase_peaks = 0, [], {}  # phase_peaks: largest live device bytes of the library's pool per phase, timed steps
        t0 = time.perf_counter()
        for s in range(steps_req):
            if s > 0 and agree(time.perf_counter() + (1.1 + tail_steps) * t_step > deadline):
                break
            ts = time.perf_counter()
            this = step_list[s] if step_list is not None else step0 + s
            res = one_step(this)
            t_step = time.perf_counter() - ts
            n_steps += 1
            for k_, v_ in (res.get("phase_peak_GB") or {}).items():
                phase_peaks[k_] = max(phase_peaks.get(k_, 0.0), float(v_))
            if "search_trace" in res:
                decisions.append(dict(decisions_of(res, args.seed_base + this), wall_s=round(t_step, 3)))
        fence()
        dt = time.perf_counter() - t0
        try:  # device memory in use on this rank's GPU after the timed steps (the library's pool keeps the call's blocks cached)
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            run_config.hbm_in_use_gb = round((total_b - free_b) / 1e9, 1)
            run_config.hbm_peak_live_gb = max(phase_peaks.values()) if phase_peaks else None
            run_config.hbm_phase_peaks_gb = phase_peaks
        except Exception:
            run_config.hbm_in_use_gb = None
        if world > 1:  # MAX over the ranks (through the library's communicator when there is one)
            dt = float(shard.allgather_small(np.array([dt])).max())
        return {"N": N, "M": M, "X": X, "res": res, "dt": dt, "steps": n_steps, "warmup": n_warm, "synth_s": t_synth,
                "row_shard": row_shard, "draws_s": one_step.draws_s, "decisions": decisions, "last_step": step0 + n_steps - 1}

    if args.stage_timing:
        ctx.set_timing(True)
    deadline = T_PROCESS_START + args.budget_s
    # reserve for what follows the timed region: roofline probe, CPU baseline sample, extra configs
    reserve = (0 if args.no_roofline else 25) + (0 if args.no_cpu_baseline else 45)
    n_min0 = min(CONFIGS[args.config][:2])
    strict_planned = args.strict_fp32 == "on" or (args.strict_fp32 == "auto" and n_min0 >= 16000 and not args.row_shard)
    strict_req = max(1, args.strict_steps) if strict_planned else 0
    # a fp32 step takes ~1.7 x a split one at cfg4: keep that many step durations of the budget free for the other variant's steps
    main_r = run_config(args.config, args.steps, args.warmup, deadline - reserve,
                        tail_steps=(0.65 if args.precision == 0 else 1.8) * strict_req)
    N, M, X, res, dt = main_r["N"], main_r["M"], main_r["X"], main_r["res"], main_r["dt"]
    steps = main_r["steps"]

    if args.stage_timing and rank == 0:
        st = {k: ctx.timing(k) for k in ("scale", "gram", "sytrd", "sy2sb", "sb2st", "stebz", "stein", "ormtr", "sbr_q2", "sbr_q1",
                                          "chefsi", "corr", "recover")}
        print("stage totals (ms, calls):", {k: (round(v[0], 1), v[1]) for k, v in st.items()}, "wall_s", round(dt, 2), file=sys.stderr)
        ctx.set_timing(False)
    out = None
    precision = ctx.get_option("precision")
    if rank == 0:
        ms_per_step = dt / max(1, steps) * 1e3
        row_shard = main_r["row_shard"]
        n_streams = args.streams if args.streams is not None else (3 if min(N, M) < 16000 else 2)
        out = {
            "metric": "sclens() cells*genes/s (wall-clock of one full sclens() call)", "value": round(N * M * steps / dt, 1),
            "unit": "cells*genes/s", "n_gpus": world, "steps": steps, "warmup": main_r["warmup"],
            "steps_requested": args.steps, "warmup_requested": args.warmup, "budget_s": args.budget_s,
            "ms_per_step": round(ms_per_step, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE_SPLIT if precision != 0 else "f32", "data": "synthetic",
            "dtype_note": DTYPE_NOTE if precision != 0 else DTYPE_NOTE_F32,
            "config": {"workload": f"{args.config}: synthetic Poisson-lognormal counts {N} cells x {M} genes, sparsity "
                                   f"{1 - X.nnz / (N * M):.3f}, full sclens() incl. sparsity search and {args.n_perturb}-member "
                                   f"perturbation ensemble", "N": N, "M": M, "nnz": int(X.nnz), "n_perturb": args.n_perturb,
                       "parallelism": (f"cells row-sharded over {world} ranks (local candidates): per decomposition 4 small all-reduces + one "
                                       f"reduce of the {M}x{M} fp32 partial Gram matrix onto the rank that decomposes it; search and "
                                       f"ensemble in rounds of {world}" if row_shard
                                       else f"single GPU, {n_streams} concurrent decompositions (HIP streams)" if world == 1 else
                                       f"search rounds of {world}x{n_streams} + ensemble t%{world}, 1 RCCL all-gather "
                                       f"issued by the library on its own buffers"),
                       "comm": shard.describe(), "precision": precision, "keep_warm": True,
                       "ensemble_tail": res.get("ensemble_tail")},
            "sclens_wall_s": round(dt / max(1, steps), 3),
            "observed": {"signals": int(len(res.get("signal_ev", []))), "robust_signals": int(len(res.get("sig_id", []))),
                         "search_iters": int(res["n_search"]), "p_": res["p_"], "synth_s": round(main_r["synth_s"], 1),
                         "ensemble_partial_eig": {"used": int(res["partial_eig"][0]), "fallback_to_full": int(res["partial_eig"][1])},
                         "phase_s_rank0_last_step": dict({"draws_host": round(main_r["draws_s"], 4)}, **res.get("phase_s", {})),
                         "decisions_per_step": main_r["decisions"],
                         "hbm_in_use_GB_after_timed_steps": getattr(run_config, "hbm_in_use_gb", None),
                         "hbm_peak_live_GB": getattr(run_config, "hbm_peak_live_gb", None),
                         "hbm_peak_live_GB_by_phase": getattr(run_config, "hbm_phase_peaks_gb", None),
                         "search_job_s_last_step": [list(q) for q in res.get("search_job_s", [])],
                         "first_phase_jobs_s_last_step": [list(q) for q in res.get("first_phase_s", [])]},
        }
    # ---- the OTHER arithmetic variant (precision = 1 - the timed one; the worker contexts of a call inherit it): further steps on the draws of
    #      the timed steps whose search statistic came closest to its threshold (those are the ones that can decide differently)
    extra = {}
    n_min = min(N, M)
    other = 0 if precision != 0 else 1
    want_other = strict_req > 0 and not main_r["row_shard"]
    t_other_est = (1.9 if other == 0 else 0.6) * dt / max(1, steps)
    if want_other and steps > 0 and not agree(time.perf_counter() + t_other_est > deadline - reserve):
        k_other = min(strict_req, steps)
        ranked = sorted(main_r["decisions"], key=lambda d_: (d_["min_abs_margin"] if d_.get("min_abs_margin") is not None else 1e9))
        chosen = sorted(int(d_["seed"]) - args.seed_base for d_ in ranked[:k_other]) or list(range(max(0, main_r["last_step"] - k_other + 1), main_r["last_step"] + 1))
        chosen = [int(v) for v in shard.bcast_host(np.array(chosen, dtype=np.float64), 0)] if world > 1 else chosen
        with ctx.options(precision=other):
            r = run_config(args.config, len(chosen), 0, deadline - reserve, step_list=chosen)
        if rank == 0 and r["steps"] > 0:
            by_seed = {d_["seed"]: d_ for d_ in main_r["decisions"]}
            pairs = [(by_seed.get(d_["seed"]), d_) for d_ in r["decisions"]]
            same = lambda a, b: all(a[q] == b[q] for q in ("signals", "robust_signals", "search_iters", "p_"))
            differ = [b["seed"] for a, b in pairs if a is not None and not same(a, b)]
            key = "value_strict_fp32" if other == 0 else "value_split_f16"
            pre = "strict" if other == 0 else "split"
            out[key] = round(r["N"] * r["M"] * r["steps"] / r["dt"], 1)
            out[pre + "_steps"] = r["steps"]
            out[pre + "_ms_per_step"] = round(r["dt"] / r["steps"] * 1e3, 1)
            compared = [p_ for p_ in pairs if p_[0] is not None]
            out["decisions_differ"] = f"{len(differ)} of {len(compared)}" if compared else None
            extra["other_variant"] = {"precision": other, "sclens_wall_s": round(r["dt"] / r["steps"], 3), "steps": r["steps"], "value": out[key],
                                      "same_draws_as": "the timed steps with the smallest |search statistic - p_th| (seeds below)",
                                      "seeds": [d_["seed"] for d_ in r["decisions"]], "decisions": r["decisions"],
                                      "seeds_whose_decisions_differ": differ,
                                      "max_abs_diff_d5_second_smallest": [
                                          (None if a is None else round(max(abs(x - y) for x, y in zip(a["d5_second_smallest"], b["d5_second_smallest"])), 6))
                                          for a, b in pairs],
                                      "note": ("context option precision = 0: every dense product on the fp32 MFMA" if other == 0 else
                                               "context option precision = 1: large products from operands split into two fp16 pieces (22 bits)")}
    # ---- extra configs (one timed step each), while the budget lasts
    for cfg in [c for c in args.extra_configs.split(",") if c]:
        if agree(time.perf_counter() + 60 > deadline - reserve):
            break
        r = run_config(cfg, 1, 0, deadline - reserve)
        if rank == 0:
            extra[cfg] = {"N": r["N"], "M": r["M"], "sclens_wall_s": round(r["dt"], 3), "value": round(r["N"] * r["M"] / r["dt"], 1),
                          "search_iters": int(r["res"]["n_search"]), "signals": int(len(r["res"].get("signal_ev", [])))}
    if rank == 0:
        if extra:
            out["extra"] = extra
        n = min(N, M)
        if not args.no_roofline:
            stages = stage_probe(ctx, X, N, M)
            if "sy2sb_dense_to_band" in stages:
                # two-stage solver: ONE search step's eigensolve (n/2 vectors) is the stage with the most wall time (S of them per
                # call, each ~85 % of its step). Algorithmic work: SURVEY 8(d) `F_eig(bottom n/2) = 4/3 n^3 + 2 n^2 (n/2)`; the
                # denominator is the sum of the HIP-event times of every stage that replaces `syevd!` (scLENS.jl:377).
                parts = ["sy2sb_dense_to_band", "sb2st_bulge_chasing", "stebz", "stein", "q2_back_transform", "q1_back_transform"]
                solve_ms = sum(stages[k]["ms"] for k in parts if k in stages)
                flop = 4.0 / 3.0 * float(n) ** 3 + 2.0 * float(n) ** 2 * (n // 2)
                ach = flop / (solve_ms * 1e-3) / 1e12
                pmc = PMC_EIG_TRAFFIC[1 if precision != 0 else 0]
                pmc = pmc if abs(n - pmc["n"]) <= 64 else None
                out["roofline"] = {"bound": "mfma", "kernel": "two-stage symmetric eigensolver of one search step (sy2sb + sb2st + stebz + "
                                                             "stein + Q2 + Q1, n/2 eigenvectors): the stage with the most wall time",
                                   "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
                                   "frac": round(ach / MFMA_F32_PEAK_TFS, 4),
                                   "traffic": (pmc["bytes"] if pmc else None),
                                   "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE doubled on the 16-byte-per-lane "
                                                      f"streams as MI355X_MICROARCH.md prescribes) over one eigensolve of order {pmc['n']} with {pmc['vectors']} "
                                                      f"vectors: {pmc['bytes']:.3g} B = {pmc['bytes'] / pmc['algorithmic_bytes']:.2f} x the algorithmic "
                                                      f"{pmc['algorithmic_bytes']:.3g} B ({pmc['source']}); a constant of the build, not measured in this run"
                                                      if pmc else None),
                                   "n": n, "vectors": n // 2, "launch_ms": round(solve_ms, 2), "algorithmic_flop_per_launch": flop,
                                   "stage_ms": {k: stages[k]["ms"] for k in parts if k in stages},
                                   "note": ("time-weighted rate of the whole stage -- six kernel families, two of them latency chains -- against the fp32 "
                                            "MFMA peak its products run on" if precision == 0 else
                                            "time-weighted fp32-EQUIVALENT rate of the whole stage against the fp32 MFMA peak (with precision = 1 "
                                            "its large products issue 3x their fp32-equivalent flop on the fp16 MFMA: see dtype_note and the "
                                            "per-stage fractions against the fp16 peak in the detail file)")}
            else:
                out["roofline"] = symv_probe(ctx, n)
            out["roofline"]["stages"] = stages
        if not args.no_cpu_baseline:
            left = deadline - time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(N, M, int(res["n_search"]), args.n_perturb, max(20.0, min(90.0, left - 10)))
        out["bench_wall_s"] = round(time.perf_counter() - T_PROCESS_START, 1)
        detail_path = os.environ.get("SCLENS_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
        except OSError as e:
            print(f"[bench] detail file not written ({e})", file=sys.stderr)
            detail_path = None
        print("[bench] full record:", json.dumps(out), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(out, detail_path)), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
