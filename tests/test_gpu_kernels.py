"""-m gpu: kernel-level parity of the HIP building blocks against NumPy/SciPy float64, through the C ABI."""
import numpy as np
import pytest
import scipy.linalg as sla

from devutil import DevArray, pad_rows, rup

pytestmark = pytest.mark.gpu


def _gemm(ctx, P, Q, C0, alpha, beta, q_kcontig, lower=0, absmax=False):
    M, K = P.shape
    N = Q.shape[0] if q_kcontig else Q.shape[1]
    ldp, ldq, ldc = rup(P.shape[1], 4), rup(Q.shape[1], 4), rup(N, 4)
    dP, dQ = DevArray(ctx, pad_rows(P, ldp)), DevArray(ctx, pad_rows(Q, ldq))
    dC = DevArray(ctx, pad_rows(C0, ldc))
    dmax = DevArray(ctx, nbytes=4 * N) if absmax else None
    ctx.check(ctx.lib.sclens_hip_dev_gemm_f32(ctx.h, dP.p, dQ.p, dC.p, M, N, K, ldp, ldq, ldc, alpha, beta, q_kcontig, lower,
                                              dmax.p if absmax else None))
    ctx.sync()
    out = dmax.get((N,), np.float32) if absmax else dC.get((M, ldc), np.float32)[:, :N]
    for d in (dP, dQ, dC, dmax):
        if d is not None:
            d.free()
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (130, 257, 33), (64, 300, 1000), (1000, 77, 515), (5, 5, 3),
                                   (300, 520, 1000), (512, 256, 64), (700, 161, 37), (161, 769, 4100), (700, 64, 515), (300, 40, 96)])
def test_gemm_nt_nn(ctx, M, N, K, opt):
    opt(gemm_force=1)  # NT shapes that fit go through the 256x256 kernel as well
    rng = np.random.default_rng(M * 7 + N)
    P = rng.standard_normal((M, K)).astype(np.float32)
    Qn = rng.standard_normal((N, K)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = 0.5 * (P.astype(np.float64) @ Qn.T.astype(np.float64)) + 2.0 * C0
    out = _gemm(ctx, P, Qn, C0, 0.5, 2.0, 1)
    tol = 2e-6 * np.sqrt(K) * np.abs(ref).max() + 1e-5
    assert np.abs(out - ref).max() < tol
    out2 = _gemm(ctx, P, np.ascontiguousarray(Qn.T), C0, 0.5, 2.0, 0)
    assert np.abs(out2 - ref).max() < tol
    # C += P Q' (alpha = beta = 1): the large-tile kernels start their accumulators from C
    ref1 = P.astype(np.float64) @ Qn.T.astype(np.float64) + C0
    out3 = _gemm(ctx, P, Qn, C0, 1.0, 1.0, 1)
    assert np.abs(out3 - ref1).max() < 2e-6 * np.sqrt(K) * np.abs(ref1).max() + 1e-5


def test_gemm_asymmetric_identity(ctx):
    # A = I with an asymmetric B catches a transposed C write
    n = 96
    B = np.arange(n * n, dtype=np.float32).reshape(n, n) / 100.0
    out = _gemm(ctx, np.eye(n, dtype=np.float32), np.ascontiguousarray(B.T), np.zeros((n, n), np.float32), 1.0, 0.0, 1)
    assert np.array_equal(out, B)


@pytest.mark.parametrize("big", [0, 1])
@pytest.mark.parametrize("n,K", [(200, 64), (333, 257), (1300, 100), (520, 16)])
def test_gemm_lower_mirror_and_absmax(ctx, n, K, big, opt):
    if big:
        opt(gemm_force=1)
    rng = np.random.default_rng(n)
    P = rng.standard_normal((n, K)).astype(np.float32)
    Q = rng.standard_normal((n, K)).astype(np.float32)
    C0 = rng.standard_normal((n, n)).astype(np.float32)
    C0 = (C0 + C0.T) / 2
    # symmetric rank-2K style update: P Q^T + Q P^T via [P|Q][Q|P]^T
    PQ, QP = np.hstack([P, Q]), np.hstack([Q, P])
    ref = C0 - (PQ.astype(np.float64) @ QP.T.astype(np.float64))
    out = _gemm(ctx, PQ, QP, C0, -1.0, 1.0, 1, lower=1)
    assert np.array_equal(out, out.T), "lower+mirror must give an exactly symmetric matrix"
    assert np.abs(out - ref).max() < 1e-4 * np.abs(ref).max()
    out1 = _gemm(ctx, PQ, -QP, C0, 1.0, 1.0, 1, lower=1)  # the form the band reduction uses (negated operand, alpha = beta = 1)
    assert np.array_equal(out1, out1.T)
    assert np.abs(out1 - ref).max() < 1e-4 * np.abs(ref).max()
    am = _gemm(ctx, P, Q, np.zeros((n, n), np.float32), 1.0, 0.0, 1, absmax=True)
    refm = np.abs(P.astype(np.float64) @ Q.T.astype(np.float64)).max(axis=0)
    assert np.abs(am - refm).max() < 1e-4 * refm.max()


def _sym(n, seed, K=None):
    rng = np.random.default_rng(seed)
    K = K or 2 * n
    B = rng.standard_normal((n, K))
    B -= B.mean(axis=0, keepdims=True)  # exact null vector like the centred matrices of the path
    return (B @ B.T / K).astype(np.float32)


@pytest.mark.parametrize("n", [3, 64, 129, 300, 1000])
def test_sytrd_stebz_eigenvalues(ctx, n):
    A = _sym(n, n)
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dd, de, dt, dw = (DevArray(ctx, nbytes=8 * n) for _ in range(4))
    ctx.check(ctx.lib.sclens_hip_dev_sytrd_f32(ctx.h, dA.p, n, lda, dd.p, de.p, dt.p))
    ctx.check(ctx.lib.sclens_hip_dev_stebz_f64(ctx.h, dd.p, de.p, n, dw.p))
    ctx.sync()
    d, e, w = dd.get((n,), np.float64), de.get((n,), np.float64), dw.get((n,), np.float64)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    # bisection is exact (fp64) for the tridiagonal it was given
    wt = sla.eigvalsh_tridiagonal(d, e[: n - 1]) if n > 1 else d
    assert np.abs(w - wt).max() < 1e-12 * max(1.0, np.abs(wt).max())
    # the fp32 reduction is backward stable: eigenvalue error ~ sqrt(n) eps32 ||A||
    assert np.abs(w - ref).max() < 4e-7 * np.sqrt(n) * np.abs(ref).max() + 1e-7
    for x in (dA, dd, de, dt, dw):
        x.free()


@pytest.mark.parametrize("n,K", [(300, 1000), (1030, 4100), (515, 257)])
def test_gram_on_split_fp16_operands_matches_float64(ctx, n, K, opt):
    """sclens_hip_dev_gram_f32 from the split image of the operand (context option gram_split_min_n: every entry as two fp16 pieces after a
    power-of-two scaling, three fp16 matrix instructions per product, fp32 accumulation) on a matrix shaped like a scaled count
    matrix -- a small negative background, a few per cent of entries up to ~80 -- against the float64 product; the fp32 path on the
    same input for comparison. Exactly symmetric either way."""
    rng = np.random.default_rng(n + K)
    B = (-0.02 * rng.random((n, K))).astype(np.float32)
    nz = rng.random((n, K)) < 0.05
    B[nz] = (rng.gamma(0.7, 4.0, size=int(nz.sum())) + 0.1).astype(np.float32)
    B[rng.integers(0, n, 20), rng.integers(0, K, 20)] = 80.0
    B -= B.mean(axis=0, keepdims=True)
    ldb, lda = rup(K, 32), rup(n, 32)
    ref = B.astype(np.float64) @ B.astype(np.float64).T / K
    err, got = {}, {}
    for mode in ("64", "64/two-buffer", "0"):
        opt(gram_split_min_n=int(mode.split("/")[0]), split_pipe=0 if "two-buffer" in mode else 1)
        Bp = np.zeros((n, ldb), np.float32)
        Bp[:, :K] = B
        dB, dA = DevArray(ctx, Bp), DevArray(ctx, nbytes=4 * n * lda)
        ctx.check(ctx.lib.sclens_hip_dev_gram_f32(ctx.h, dB.p, n, K, ldb, float(K), dA.p, lda))
        ctx.sync()
        A = dA.get((n, lda), np.float32)[:, :n]
        dB.free()
        dA.free()
        assert np.array_equal(A, A.T)
        err[mode], got[mode] = np.abs(A - ref).max() / np.abs(ref).max(), A.copy()
    assert np.array_equal(got["64"], got["64/two-buffer"])  # the pipelined stage loop (round 5): the same products in the same order
    assert err["64"] < 2e-6, err
    assert err["64"] < 2 * err["0"] + 3e-7, err


def _tridiag_cases():
    rng = np.random.default_rng(4)
    n = 777
    cases = {}
    cases["random"] = (rng.standard_normal(n), rng.standard_normal(n - 1))
    d = rng.standard_normal(n)
    e = rng.standard_normal(n - 1)
    e[[100, 101, 400, 776 - 64]] = 0.0  # decoupled blocks, one of them a 1 x 1 block
    cases["decoupled"] = (d, e)
    cases["wilkinson"] = (np.abs(np.arange(n) - n // 2).astype(np.float64), np.ones(n - 1))  # pairs agreeing to 1e-14
    cases["toeplitz_2_-1"] = (np.full(n, 2.0), np.full(n - 1, -1.0))  # probes that hit eigenvalues of leading blocks
    cases["graded"] = (10.0 ** np.linspace(6, -6, n), 10.0 ** np.linspace(2.5, -9, n - 1))
    cases["tiny_offdiagonals"] = (rng.standard_normal(n) * 1e3, rng.standard_normal(n - 1) * 1e-18)
    cases["large_scale"] = (rng.standard_normal(n) * 1e150, rng.standard_normal(n - 1) * 1e150)
    cases["all_zero"] = (np.zeros(n), np.zeros(n - 1))
    cases["padded_like_two_stage"] = (np.concatenate([rng.standard_normal(n - 60) + 5.0, np.zeros(60)]),
                                      np.concatenate([rng.standard_normal(n - 61), np.zeros(60)]))
    return cases


@pytest.mark.parametrize("name", list(_tridiag_cases()))
@pytest.mark.parametrize("form", ["product", "ratio"])
def test_stebz_tridiagonal_cases(ctx, name, form, opt):
    """sclens_hip_dev_stebz_f64 on given tridiagonal matrices: the division-free Sturm count (three-term recurrence of the leading
    minors on a copy scaled by a power of two, rescaled every eight steps, exact zeros replaced) and the ratio form
    (context option bisect_div = 1) against LAPACK, to a few ulps of the norm."""
    d, e = _tridiag_cases()[name]
    n = len(d)
    if form == "ratio":
        opt(bisect_div=1)
    dd, de, dw = DevArray(ctx, np.ascontiguousarray(d)), DevArray(ctx, np.concatenate([e, [0.0]])), DevArray(ctx, nbytes=8 * n)
    ctx.check(ctx.lib.sclens_hip_dev_stebz_f64(ctx.h, dd.p, de.p, n, dw.p))
    ctx.sync()
    w = dw.get((n,), np.float64)
    for x in (dd, de, dw):
        x.free()
    ref = sla.eigvalsh_tridiagonal(d, e)
    nrm = max(np.abs(d).max() + 2 * np.abs(e).max(), 1e-300)
    assert np.all(np.isfinite(w)) and np.all(np.diff(w) >= -1e-15 * nrm)  # independent bisections: ordered up to their own width
    assert np.abs(w - ref).max() <= 8e-15 * nrm * np.sqrt(n) + 1e-300, (name, form, np.abs(w - ref).max() / nrm)


@pytest.mark.parametrize("n,lo,hi", [(64, 0, 64), (300, 0, 300), (300, 290, 300), (515, 100, 360), (1000, 0, 1000)])
def test_eigh_vectors(ctx, n, lo, hi):
    A = _sym(n, 1000 + n)
    lda = rup(n, 32)
    m = hi - lo
    dA = DevArray(ctx, pad_rows(A, lda))
    dw = DevArray(ctx, nbytes=8 * n)
    dZ = DevArray(ctx, nbytes=4 * m * lda)
    ctx.check(ctx.lib.sclens_hip_dev_eigh_f32(ctx.h, dA.p, n, lda, dw.p, lo, hi, dZ.p, lda))
    ctx.sync()
    w = dw.get((n,), np.float64)
    Z = dZ.get((m, lda), np.float32)[:, :n].astype(np.float64)  # rows = eigenvectors
    A64 = A.astype(np.float64)
    nrm = np.abs(np.linalg.eigvalsh(A64)).max()
    resid = np.abs(Z @ A64 - w[lo:hi, None] * Z).max()
    orth = np.abs(Z @ Z.T - np.eye(m)).max()
    assert resid < 2e-5 * nrm * np.sqrt(n / 64 + 1), resid
    assert orth < 2e-5 * np.sqrt(n / 64 + 1), orth
    for x in (dA, dw, dZ):
        x.free()


def test_eigh_repeated_eigenvalues_are_orthonormalised(ctx):
    """Exactly repeated eigenvalues (two identical diagonal blocks): inverse iteration alone would return non-orthogonal
    vectors inside each 2-dimensional eigenspace; the cluster Gram-Schmidt pass must restore orthonormality."""
    h = 150
    B = _sym(h, 5)
    A = np.zeros((2 * h, 2 * h), np.float32)
    A[:h, :h] = B
    A[h:, h:] = B
    n = 2 * h
    lda = rup(n, 32)
    dA = DevArray(ctx, pad_rows(A, lda))
    dw = DevArray(ctx, nbytes=8 * n)
    dZ = DevArray(ctx, nbytes=4 * n * lda)
    ctx.check(ctx.lib.sclens_hip_dev_eigh_f32(ctx.h, dA.p, n, lda, dw.p, 0, n, dZ.p, lda))
    ctx.sync()
    w = dw.get((n,), np.float64)
    Z = dZ.get((n, lda), np.float32)[:, :n].astype(np.float64)
    assert np.abs(w[0::2] - w[1::2]).max() < 1e-5 * w.max()  # every eigenvalue twice
    assert np.abs(Z @ Z.T - np.eye(n)).max() < 5e-5
    assert np.abs(Z @ A.astype(np.float64) - w[:, None] * Z).max() < 5e-5 * w.max()
    for x in (dA, dw, dZ):
        x.free()


def _sytrd_on(c, A, n, lda):
    dA = DevArray(c, pad_rows(A, lda))
    dd, de, dt = (DevArray(c, nbytes=8 * n) for _ in range(3))
    c.check(c.lib.sclens_hip_dev_sytrd_f32(c.h, dA.p, n, lda, dd.p, de.p, dt.p))
    c.sync()
    out = (dA.get((n, lda), np.float32), dd.get((n,), np.float64), de.get((n,), np.float64), dt.get((n,), np.float32))
    for x in (dA, dd, de, dt):
        x.free()
    return out


def test_release_scratch_hands_the_eigensolver_workspaces_back_and_the_next_call_rebuilds_them(ctx):
    """sclens_hip_release_scratch (round 5): api.sclens() returns the idle scratch families of its contexts to the pool at the phase
    boundaries. After a release the pool holds the bytes, the context's next decomposition allocates again and gives the same bits."""
    import ctypes as C

    from sclens_amd._lib import Context

    n, lda = 1100, rup(1100, 32)
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, 300)).astype(np.float32)
    A = np.zeros((n, lda), np.float32)
    A[:, :n] = B @ B.T / 300
    c2 = Context(ctx.device)
    c2.set_option("two_stage", 1)
    try:
        out = []
        for rep in range(2):
            dA, dw, dZ = DevArray(c2, A), DevArray(c2, nbytes=8 * n), DevArray(c2, nbytes=4 * 200 * lda)
            c2.check(c2.lib.sclens_hip_dev_eigh_f32(c2.h, dA.p, n, lda, dw.p, n - 200, n, dZ.p, lda))
            c2.sync()
            out.append((dw.get((n,), np.float64), dZ.get((200, lda), np.float32)))
            for x in (dA, dw, dZ):
                x.free()
            cached0 = C.c_int64(0)
            c2.lib.sclens_hip_pool_stats(c2.device, C.byref(cached0), None, None, None)
            c2.release_scratch("eigensolver")
            cached1 = C.c_int64(0)
            c2.lib.sclens_hip_pool_stats(c2.device, C.byref(cached1), None, None, None)
            assert cached1.value > cached0.value  # the workspaces went to the pool's idle list
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        with pytest.raises(Exception):
            c2.release_scratch("no such family")
    finally:
        c2.close()
