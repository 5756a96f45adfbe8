"""-m gpu: building blocks of the two-stage tridiagonalisation (sbr.hip, work in progress; the one-stage solver is what
eig_values uses). Stage 1: dense symmetric -> band of half-width 64."""
import ctypes as C
import os

import numpy as np
import pytest

from devutil import DevArray, pad_rows, rup

pytestmark = pytest.mark.gpu
SB = 64


def _sym_psd(n, seed, K=None):
    rng = np.random.default_rng(seed)
    K = K or 2 * n
    B = rng.standard_normal((n, K)).astype(np.float32)
    B -= B.mean(axis=0, keepdims=True)  # one structurally zero eigenvalue, like the centred data
    return (B @ B.T / K).astype(np.float32)


def _run_sy2sb(ctx, A):
    n = A.shape[0]
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    npan = n // SB - 1
    dT = DevArray(ctx, nbytes=4 * max(1, npan) * SB * SB)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    out = dA.get((n, lda), np.float32)[:, :n].astype(np.float64)
    T = dT.get((max(1, npan), SB, SB), np.float32).astype(np.float64)
    dA.free()
    dT.free()
    return out, T, bd.value


def _band_of(out):
    n = out.shape[0]
    i, j = np.indices((n, n))
    low = np.where((i - j >= 0) & (i - j <= SB), out, 0.0)
    return low + np.tril(low, -1).T


@pytest.fixture(params=["default", "delayed", "split", "delayed+split"])
def delayed_update(request, opt):
    """sy2sb applies the trailing updates of two panels as one rank-256 update while the trailing matrix is large (from order
    18 432 by default), and runs trailing updates of at least 4 096 rows on the fp16 matrix cores from split operands; "delayed"
    turns the first on from order 321, "split" the second from 512 rows, so that the small cases here go through them."""
    if "delayed" in request.param:
        opt(sy2sb_delay_min=321)
    on = "split" in request.param
    # trailing updates, W = A22 V and the first back-transformation from fp16 pieces, or all of them on the fp32 matrix cores
    opt(sy2sb_split_min=512 if on else 0, sy2sb_wsplit_min=128 if on else 0, q1_split_min=32 if on else 0)
    return request.param


@pytest.mark.parametrize("n", [128, 320, 1024, 2048])
def test_sy2sb_band_has_the_same_spectrum(ctx, n, delayed_update):
    A = _sym_psd(n, n)
    out, T, bd = _run_sy2sb(ctx, A)
    assert bd == 0
    Bm = _band_of(out)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    got = np.linalg.eigvalsh(Bm)
    assert np.abs(got - ref).max() < 4e-7 * np.sqrt(n) * ref.max() + 1e-7


@pytest.mark.parametrize("log2_norm", [0, 14, 20])
def test_sy2sb_split_update_with_separate_scales(ctx, log2_norm, opt):
    """a matrix of large norm: the reflector columns of the split update's operands get their own scale, so the band keeps the
    spectrum to the fp32 path's tolerance at every norm (the shared scale does inside its range: DESIGN.md section 4)"""
    n = 2048
    A = (_sym_psd(n, 3) * np.float32(2.0 ** log2_norm)).astype(np.float32)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    opt(sy2sb_delay_min=321, sy2sb_split_min=512)
    err = {}
    for scales in ("2", "1", "2-pass"):
        opt(sy2sb_split_scales=int(scales[0]))
        # "2": the Z columns' largest entry comes from the kernel that writes Z and the V columns take the fixed scale 2^13 (default);
        # "2-pass": both maxima by a pass over the operands (until the end of round 4)
        opt(sy2sb_zmax=0 if scales == "2-pass" else 1)
        out, T, bd = _run_sy2sb(ctx, A)
        assert bd == 0
        err[scales] = np.abs(np.linalg.eigvalsh(_band_of(out)) - ref).max() / ref.max()
    assert err["2"] < 4e-7 * np.sqrt(n) + 1e-7 and err["2-pass"] < 4e-7 * np.sqrt(n) + 1e-7, err
    if log2_norm <= 14:
        assert err["1"] < 4e-7 * np.sqrt(n) + 1e-7, err


@pytest.mark.parametrize("n", [1024, 2368])
def test_sy2sb_split_update_started_from_c(ctx, n, opt):
    """the split-fp16 trailing update with its accumulators started from C / alpha (all of the C tile requested up front through
    unpredicated buffer loads; default) against C added in the epilogue (context option split_acc_init = 0): partial edge tiles, diagonal
    tiles, rank-128 and rank-256 (delayed) updates; both keep the spectrum to the fp32 path's tolerance"""
    A = _sym_psd(n, 11)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    opt(sy2sb_delay_min=321, sy2sb_split_min=512)
    err, band = {}, {}
    for mode in ("1", "0"):
        opt(split_acc_init=int(mode))
        out, T, bd = _run_sy2sb(ctx, A)
        assert bd == 0
        band[mode] = _band_of(out)
        err[mode] = np.abs(np.linalg.eigvalsh(band[mode]) - ref).max() / ref.max()
    assert err["1"] < 4e-7 * np.sqrt(n) + 1e-7 and err["0"] < 4e-7 * np.sqrt(n) + 1e-7, err
    assert err["1"] < 3 * err["0"] + 2e-7, err


@pytest.mark.parametrize("shape", ["gram", "dominant", "graded"])
@pytest.mark.parametrize("log2_norm", [-20, 0, 14, 20])
def test_sy2sb_w_product_from_fp16_pieces(ctx, log2_norm, shape, opt):
    """W = A22 V with the trailing matrix split into fp16 pieces in registers (sbr_w_split; scale from the largest absolute row sum,
    which bounds every entry of every trailing matrix): the band keeps the spectrum to the tolerance of the fp32 product, whatever
    the norm, with one dominant eigenvalue (entries of the trailing matrices far above those of A) and with graded rows"""
    n = 1536
    A = _sym_psd(n, 5).astype(np.float64)
    if shape == "dominant":
        u = np.ones(n) / np.sqrt(n)
        A = A + 3000.0 * np.outer(u, u)  # lambda_max 3000 x the bulk, entries of A ~ 2: the band holds an entry ~ 3000
    elif shape == "graded":
        d = np.logspace(0, -3, n)
        A = A * d[:, None] * d[None, :]
    A = (A * 2.0 ** log2_norm).astype(np.float32)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    opt(sy2sb_split_min=0)  # trailing updates on the fp32 matrix cores: the W product is what differs
    err = {}
    for w in ("128", "0"):
        opt(sy2sb_wsplit_min=int(w))
        out, T, bd = _run_sy2sb(ctx, A)
        assert bd == 0
        err[w] = np.abs(np.linalg.eigvalsh(_band_of(out)) - ref).max() / np.abs(ref).max()
    assert err["128"] < 4e-7 * np.sqrt(n) + 1e-7, err
    assert err["128"] < 3 * err["0"] + 2e-7, err


@pytest.mark.parametrize("n", [128, 256, 448, 832])
def test_sy2sb_reflectors_reproduce_the_band(ctx, n, delayed_update):
    """Q1 = H_0 H_1 ... with H_p = I - V_p T_p V_p' (V_p from the upper part of the output) satisfies Q1' A Q1 = band."""
    A = _sym_psd(n, 7 * n)
    out, T, bd = _run_sy2sb(ctx, A)
    assert bd == 0
    Q = np.eye(n)
    for p in range(n // SB - 1):
        c0, r0 = p * SB, (p + 1) * SB
        V = out[c0:c0 + SB, r0:].T  # (n - r0) x SB, unit lower trapezoidal
        assert np.allclose(np.diag(V[:SB]), 1.0) and np.abs(np.triu(V[:SB], 1)).max() == 0
        H = np.eye(n)
        H[r0:, r0:] -= V @ T[p] @ V.T
        assert np.abs(H.T @ H - np.eye(n)).max() < 5e-6  # orthogonal to fp32 accuracy
        Q = Q @ H
    Bm = _band_of(out)
    A64 = A.astype(np.float64)
    assert np.abs(Q.T @ A64 @ Q - Bm).max() < 2e-5 * np.abs(A64).max() * np.sqrt(n / 64)


def test_sy2sb_rank_deficient_panel_raises_the_flag(ctx):
    n = 256
    A = np.zeros((n, n), dtype=np.float32)
    A[:8, :8] = _sym_psd(8, 1)
    _, _, bd = _run_sy2sb(ctx, A)
    assert bd == 1
    rc = ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, None, 100, 128, None, None)
    assert rc == 1  # order not a multiple of 64


@pytest.mark.parametrize("n", [128, 192, 576, 2048])
def test_two_stage_tridiagonal_has_the_same_spectrum(ctx, n):
    """sy2sb + sb2st (bulge chasing in one persistent kernel, sweeps pipelined through progress counters): the tridiagonal
    matrix has the spectrum of A; the off-band part of the packed band is annihilated."""
    import scipy.linalg as sla

    A = _sym_psd(n, 3 * n + 1)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
    dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    assert bd.value == 0
    ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
    ctx.sync()
    d, e = dd.get((n,), np.float64), de.get((n,), np.float64)
    for x in (dA, dT, dd, de):
        x.free()
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    got = sla.eigvalsh_tridiagonal(d, e[: n - 1])
    assert np.abs(got - ref).max() < 6e-7 * np.sqrt(n) * ref.max() + 1e-7


@pytest.mark.parametrize("n", [128, 192, 576, 2048, 4160])
def test_bulge_chase_kernels_agree_bitwise(ctx, n, opt):
    """sbr_chase_mb (row hand-off between sweeps by tagged messages, blocks prefetched one task ahead) does the arithmetic of
    sbr_chase (counter + loads per task) in the same order: the tridiagonal matrix and the stored reflectors (seen through the
    second back-transformation of a random block) have the same bits."""
    A = _sym_psd(n, 7 * n + 5)
    lda = rup(n, 32)
    rng = np.random.default_rng(n)
    m = 48
    Z0 = np.zeros((m, lda), dtype=np.float32)
    Z0[:, :n] = rng.standard_normal((m, n)).astype(np.float32)
    got = {}
    for mb in ("0", "1"):
        opt(chase_mb=int(mb))
        dA = DevArray(ctx, pad_rows(A, lda))
        dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
        dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
        bd = C.c_int(-1)
        ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
        assert bd.value == 0
        ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
        dZ = DevArray(ctx, Z0)
        ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q2_f32(ctx.h, n, dZ.p, m, lda))
        ctx.sync()
        got[mb] = (dd.get((n,), np.float64), de.get((n,), np.float64), dZ.get((m, lda), np.float32))
        for x in (dA, dT, dd, de, dZ):
            x.free()
    for a, b in zip(got["0"], got["1"]):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(got["1"][2]))


@pytest.mark.parametrize("n,m", [(256, 256), (448, 100), (1024, 37)])
def test_first_back_transformation(ctx, n, m, delayed_update):
    """Eigenvectors of the band matrix (host, float64) multiplied by Q1 on the device are eigenvectors of A."""
    A = _sym_psd(n, 5 * n + 2)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    npan = n // SB - 1
    dT = DevArray(ctx, nbytes=4 * max(1, npan) * SB * SB)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    out = dA.get((n, lda), np.float32)[:, :n].astype(np.float64)
    wB, ZB = np.linalg.eigh(_band_of(out))
    sel = np.linspace(0, n - 1, m).astype(int)
    Zt = np.zeros((m, lda), dtype=np.float32)
    Zt[:, :n] = ZB[:, sel].T
    dZ = DevArray(ctx, Zt)
    ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q1_f32(ctx.h, dA.p, n, lda, dT.p, dZ.p, m, lda))
    ctx.sync()
    Z = dZ.get((m, lda), np.float32)[:, :n].astype(np.float64)
    for x in (dA, dT, dZ):
        x.free()
    A64 = A.astype(np.float64)
    nrm = np.abs(wB).max()
    assert np.abs(Z @ A64 - wB[sel, None] * Z).max() < 3e-5 * nrm
    assert np.abs(np.linalg.norm(Z, axis=1) - 1).max() < 1e-4


@pytest.mark.parametrize("n,m", [(128, 128), (320, 64), (1024, 50)])
def test_two_stage_eigenvectors(ctx, n, m, delayed_update):
    """Eigenvectors of the tridiagonal matrix (host, float64) through both back-transformations are eigenvectors of A."""
    import scipy.linalg as sla

    A = _sym_psd(n, 11 * n + 3)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
    dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
    ctx.sync()
    d, e = dd.get((n,), np.float64), de.get((n,), np.float64)
    w, ZT = sla.eigh_tridiagonal(d, e[: n - 1])
    sel = np.linspace(0, n - 1, m).astype(int)
    Zt = np.zeros((m, lda), dtype=np.float32)
    Zt[:, :n] = ZT[:, sel].T
    dZ = DevArray(ctx, Zt)
    ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q2_f32(ctx.h, n, dZ.p, m, lda))
    ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q1_f32(ctx.h, dA.p, n, lda, dT.p, dZ.p, m, lda))
    ctx.sync()
    Z = dZ.get((m, lda), np.float32)[:, :n].astype(np.float64)
    for x in (dA, dT, dd, de, dZ):
        x.free()
    A64 = A.astype(np.float64)
    nrm = np.abs(w).max()
    assert np.abs(Z @ A64 - w[sel, None] * Z).max() < 5e-5 * nrm
    assert np.abs(Z @ Z.T - np.eye(m)).max() < 2e-4


@pytest.mark.parametrize("n,lo,hi", [(300, 0, 300), (515, 100, 360), (1000, 990, 1000), (2048, 0, 1025), (100, 0, 100),
                                     (8192, 4000, 4100)])
def test_two_stage_solver_behind_eigh(ctx, n, lo, hi):
    """sclens_hip_dev_eigh_f32 with the context option "two_stage": orders that are not multiples of 64 go through the
    padded copy (decoupled sentinel block), n = 100 is below the threshold and silently takes the one-stage path."""
    from sclens_amd._lib import Context

    c2 = Context(ctx.device)
    c2.set_option("two_stage", 1)
    try:
        A = _sym_psd(n, 1000 + n)
        lda = rup(n, 32)
        m = hi - lo
        dA = DevArray(c2, pad_rows(A, lda))
        dw = DevArray(c2, nbytes=8 * n)
        dZ = DevArray(c2, nbytes=4 * m * lda)
        c2.check(c2.lib.sclens_hip_dev_eigh_f32(c2.h, dA.p, n, lda, dw.p, lo, hi, dZ.p, lda))
        c2.sync()
        w = dw.get((n,), np.float64)
        Z = dZ.get((m, lda), np.float32)[:, :n].astype(np.float64)
        if n >= 128:  # the two-stage path works on a copy: A is untouched
            assert np.array_equal(dA.get((n, lda), np.float32)[:, :n], A)
        for x in (dA, dw, dZ):
            x.free()
        A64 = A.astype(np.float64)
        ref = np.linalg.eigvalsh(A64)
        assert np.abs(w - ref).max() < 6e-7 * np.sqrt(n) * ref.max() + 1e-7
        assert np.abs(Z @ A64 - w[lo:hi, None] * Z).max() < 5e-5 * ref.max() * np.sqrt(n / 64 + 1)
        assert np.abs(Z @ Z.T - np.eye(m)).max() < 3e-4
    finally:
        c2.close()


@pytest.mark.parametrize("n,lo,hi", [(700, 0, 350), (2112, 2000, 2112)])
def test_switches_that_only_move_work_give_the_same_bits(ctx, n, lo, hi):
    """The T factors of the second back-transformation built on the auxiliary stream behind the bisection (default) or in front of
    the apply kernel (context option q2_tg_early = 0), and the inverse iteration with 4 / 16 / 32 steps of loads in flight: the same
    arithmetic on the same data, so eigenvalues and eigenvectors have the same bits."""
    from sclens_amd._lib import Context

    A = _sym_psd(n, 77 + n)
    lda = rup(n, 32)
    m = hi - lo
    got = []
    for options in ({}, {"q2_tg_early": 0}, {"stein_pf": 4}, {"stein_pf": 32}):
        c2 = Context(ctx.device)
        c2.set_option("two_stage", 1)
        for key, val in options.items():
            c2.set_option(key, val)
        try:
            dA, dw, dZ = DevArray(c2, pad_rows(A, lda)), DevArray(c2, nbytes=8 * n), DevArray(c2, nbytes=4 * m * lda)
            for _ in range(2):  # twice on one context: the second call finds the first call's T factors and must not use them
                c2.h2d(dA.p, pad_rows(A, lda))
                c2.check(c2.lib.sclens_hip_dev_eigh_f32(c2.h, dA.p, n, lda, dw.p, lo, hi, dZ.p, lda))
            c2.sync()
            got.append((dw.get((n,), np.float64), dZ.get((m, lda), np.float32)[:, :n]))
            for x in (dA, dw, dZ):
                x.free()
        finally:
            c2.close()
    for w, Z in got[1:]:
        assert np.array_equal(w, got[0][0]) and np.array_equal(Z, got[0][1])
    A64 = A.astype(np.float64)
    Z = got[0][1].astype(np.float64)
    assert np.abs(Z @ A64 - got[0][0][lo:hi, None] * Z).max() < 5e-5 * np.abs(got[0][0]).max() * np.sqrt(n / 64 + 1)


def test_first_back_transformation_group_data_prepared_ahead_gives_the_same_bits(ctx):
    """Round 4: the block reflectors of the first back-transformation (clean reflector blocks, merged T factors, split images) are
    built for all groups on the auxiliary stream right after the band reduction (sbr_q1_prepare) instead of group by group inside
    the apply loop (context option q1_prep = 0): the same kernels on the same data -- the same bits. Order 4 288 = 66 panels (groups of
    8 from 64 panels), 2 112 vectors (from 2 048), the split products forced on as at the bench's order."""
    from sclens_amd._lib import Context

    n, lo, hi = 4288, 1000, 3112
    A = _sym_psd(n, 31 + n)
    lda = rup(n, 32)
    m = hi - lo
    got = []
    for prep, w1 in (("1", "1"), ("0", "1"), ("1", "0"), ("1", "2")):
        c2 = Context(ctx.device)
        c2.set_option("two_stage", 1)
        c2.set_option("q1_prep", int(prep))
        c2.set_option("q1_w1_split", int(w1))  # 0: the first product of every group on the fp32 matrix cores (round 3)
        try:
            dA, dw, dZ = DevArray(c2, pad_rows(A, lda)), DevArray(c2, nbytes=8 * n), DevArray(c2, nbytes=4 * m * lda)
            for _ in range(2):  # twice: the second decomposition must not pick up the first one's group data
                c2.h2d(dA.p, pad_rows(A, lda))
                c2.check(c2.lib.sclens_hip_dev_eigh_f32(c2.h, dA.p, n, lda, dw.p, lo, hi, dZ.p, lda))
            c2.sync()
            got.append((dw.get((n,), np.float64), dZ.get((m, lda), np.float32)[:, :n]))
            for x in (dA, dw, dZ):
                x.free()
        finally:
            c2.close()
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    # "2": Z through a split image instead of split in registers by the product's kernel (the default): the same pieces, the same bits
    assert np.array_equal(got[0][0], got[3][0]) and np.array_equal(got[0][1], got[3][1])
    A64 = A.astype(np.float64)
    for w, Zf in (got[0], got[2]):  # W1 = Z Vm' from split images (22-bit operands, fixed scale 2^13) / in fp32
        Z = Zf.astype(np.float64)
        assert np.abs(Z @ A64 - w[lo:hi, None] * Z).max() < 5e-5 * np.abs(w).max() * np.sqrt(n / 64 + 1)
        assert np.abs(Z @ Z.T - np.eye(m)).max() < 3e-4
    # the two arithmetic variants of that product agree to fp32 rounding of unit vectors (the eigenvalues come before it: same bits)
    assert np.array_equal(got[0][0], got[2][0])
    d = np.abs(got[0][1].astype(np.float64) - got[2][1].astype(np.float64)).max()
    print(f"[q1 W1 split vs fp32] max abs difference of eigenvector entries {d:.2e} (entries ~ {1 / np.sqrt(n):.1e})")
    assert d < 5e-6


@pytest.mark.parametrize("variant", [14, 15, 16, 3])
@pytest.mark.parametrize("n,m", [(192, 192), (1088, 100), (2560, 70), (640, 641 - 1)])
def test_second_back_transformation_matches_the_unblocked_reference(ctx, n, m, variant, opt):
    """The register-resident MFMA version of Q2 (16-vector wave tiles, QJ sweep blocks per pass) against the one-reflector-at-
    a-time reference kernel on the same reflectors: vector counts that are not multiples of 16 / 64, several super-blocks."""
    A = _sym_psd(n, 13 * n + 5)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
    dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
    rng = np.random.default_rng(n + m)
    Z0 = np.zeros((m, lda), dtype=np.float32)
    Z0[:, :n] = rng.standard_normal((m, n)) / np.sqrt(n)
    outs = []
    # 16 (default) / 15 / 14: pre-built 16 KB group images moved by LDS-DMA (passes of 8 / 4 blocks; one group ahead), split-fp16 products;
    # 3: fp32 products, reflectors staged by every workgroup (what `precision = 0` selects)
    opt(q2_variant=variant)
    for ref in (True, False):
        opt(q2_reference=1 if ref else 0)
        dZ = DevArray(ctx, Z0)
        ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q2_f32(ctx.h, n, dZ.p, m, lda))
        ctx.sync()
        outs.append(dZ.get((m, lda), np.float32)[:, :n].astype(np.float64))
        dZ.free()
    for x in (dA, dT, dd, de):
        x.free()
    assert np.abs(outs[0] - outs[1]).max() < 2e-5 * np.abs(outs[0]).max() * np.sqrt(n / 64)
    # orthogonal transformation: norms are preserved
    assert np.abs(np.linalg.norm(outs[1], axis=1) - np.linalg.norm(Z0[:, :n].astype(np.float64), axis=1)).max() < 1e-4


def test_second_back_transformation_variants(ctx, opt):
    """Variants that differ only in WHEN the group images and the window are fetched give the same bits (14: one group ahead, 15: two,
    16: two and passes of eight blocks); the split-fp16
    products (three fp16 matrix instructions with fp32 accumulation per product, 22-bit operands) stay within 4e-6 of the fp32 ones
    (variant 3, also what `precision = 0` selects) on unit vectors and keep orthonormal rows orthonormal to 2e-6."""
    n, m = 1344, 130
    A = _sym_psd(n, 5 * n + 1)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dT = DevArray(ctx, nbytes=4 * max(1, n // SB - 1) * SB * SB)
    dd, de = DevArray(ctx, nbytes=8 * n), DevArray(ctx, nbytes=8 * n)
    bd = C.c_int(-1)
    ctx.check(ctx.lib.sclens_hip_dev_sy2sb_f32(ctx.h, dA.p, n, lda, dT.p, C.byref(bd)))
    ctx.check(ctx.lib.sclens_hip_dev_sb2st_f32(ctx.h, dA.p, n, lda, dd.p, de.p))
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    Z0 = np.zeros((m, lda), dtype=np.float32)
    Z0[:, :n] = Q.T.astype(np.float32)
    out = {}
    for v in ("3", "14", "15", "16", "strict"):
        if v == "strict":
            opt(q2_variant=15, precision=0)
        else:
            opt(q2_variant=int(v))
        dZ = DevArray(ctx, Z0)
        ctx.check(ctx.lib.sclens_hip_dev_sbr_apply_q2_f32(ctx.h, n, dZ.p, m, lda))
        ctx.sync()
        out[v] = dZ.get((m, lda), np.float32)[:, :n]
        dZ.free()
    for x in (dA, dT, dd, de):
        x.free()
    assert np.array_equal(out["3"], out["strict"])  # precision = 0 IS the fp32 kernel, whatever q2_variant says
    # 14 / 15: one copy of the reflectors + T in a 16 KB image, the third product's operand by transposing LDS reads: three products
    assert np.array_equal(out["14"], out["15"]) and np.array_equal(out["16"], out["15"]) and np.abs(out["14"].astype(np.float64) - out["3"]).max() < 4e-6
    for v in ("3", "14"):
        Z = out[v].astype(np.float64)
        assert np.abs(Z @ Z.T - np.eye(m)).max() < 2e-6, v


def test_sclens_with_the_two_stage_solver(ctx):
    """The whole path with the context option two_stage = 1 (every worker context created inside sclens() inherits it): same decisions as
    the default solver, spectra and scores within the fp32 tolerances."""
    from sclens_amd import api
    from sclens_amd._lib import Context
    from sclens_amd.synth import synth_counts

    X = synth_counts(300, 500, seed=1, C=5, marker_frac=0.2, marker_sd=1.5)
    d = api.make_draws_native(X, seed=19)
    ref = api.sclens(X, draws=d, n_perturb=5, ctx=ctx, streams=2)
    c2 = Context(ctx.device)
    c2.set_option("two_stage", 1)
    try:
        res = api.sclens(X, draws=d, n_perturb=5, ctx=c2, streams=2)
    finally:
        c2.close()
    assert np.abs(res["L"] - ref["L"]).max() < 2e-5 * ref["L"].max()
    assert len(res["signal_ev"]) == len(ref["signal_ev"]) and res["p_"] == ref["p_"] and res["n_search"] == ref["n_search"]
    for (p1, t1), (p2, t2) in zip(res["search_trace"], ref["search_trace"]):
        assert p1 == p2 and np.abs(t1 - t2).max() < 2e-3
    assert np.array_equal(res["sig_id"], ref["sig_id"])
    assert np.abs(res["robustness_scores"]["rob_score"] - ref["robustness_scores"]["rob_score"]).max() < 3e-3
