"""CPU models of two device algorithms whose correctness is a numerical claim rather than an identity, checked against
float64 LAPACK / numpy. They restate what the KERNELS do step by step (same guards, same rescaling, same operand split), so a
flaw in the scheme shows here without a GPU; the kernels themselves are compared with float64 in the `-m gpu` tests
(`test_stebz_tridiagonal_cases`, `test_gram_on_split_fp16_operands_matches_float64`, `test_corr_colmax_split_fp16`).

  * `sturm_count_prod`  -- csrc/trieig.hip `tri_scale` + `tri_bisect_prod`: the Sturm count of a symmetric tridiagonal matrix from
    the three-term recurrence of its leading principal minors (no division), scaled to norm < 1, both running values rescaled
    by a power of two every eight steps, an exact zero replaced by a tiny value of the sign that makes it a sign change.
  * `split_product`     -- csrc/gram_bits.hip `k_split_image_scaled` + `gemm_split_kernel`: a b ~ ah bh + ah bl + al bh on fp16
    pieces of power-of-two-scaled operands with fp32 accumulation.
"""
import numpy as np
import pytest
from scipy.linalg import eigvalsh_tridiagonal

TINY = 1.0e-290


def sturm_count_prod(d, e, x):
    """number of eigenvalues of tridiag(d, e) below each probe point of `x` (vectorised over x), the kernel's way"""
    d = np.asarray(d, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    n = len(d)
    e2 = np.concatenate([e * e, [0.0]])
    # tri_bounds: Gershgorin interval; tri_scale: a power of two that brings the bound below 1 (exact)
    r = np.abs(np.concatenate([[0.0], e])) + np.abs(np.concatenate([e, [0.0]]))
    bound = max(abs((d - r).min()), abs((d + r).max()))
    sc = 1.0
    if 0.0 < bound < 1e300:
        sc = np.ldexp(1.0, -int(np.frexp(bound)[1]))
    ds, e2s, xs = d * sc, e2 * sc * sc, x * sc
    p2 = np.ones_like(xs)
    p1 = ds[0] - xs
    p1 = np.where(p1 == 0.0, -TINY, p1)
    cnt = (p1 < 0.0).astype(np.int64)
    i = 1
    while i < n:
        blk = 8 if i + 8 <= n else 1  # the kernel's unrolled groups of eight, then the remainder one by one
        for u in range(blk):
            pn = (ds[i + u] - xs) * p1 - e2s[i - 1 + u] * p2  # the kernel fuses the first product (fma); sign-equivalent here
            pn = np.where(pn == 0.0, np.copysign(TINY, -p1), pn)
            cnt += (np.signbit(pn) != np.signbit(p1)).astype(np.int64)
            p2, p1 = p1, pn
        if blk == 8:
            ex = np.frexp(np.maximum(np.abs(p1), np.abs(p2)))[1]
            p1, p2 = np.ldexp(p1, -ex), np.ldexp(p2, -ex)
        i += blk
    return cnt


def _cases():
    rng = np.random.default_rng(11)
    out = {}
    n = 301
    out["random"] = (rng.standard_normal(n), rng.standard_normal(n - 1))
    m = 10
    out["wilkinson21"] = (np.abs(np.arange(-m, m + 1)).astype(float), np.ones(2 * m))
    out["toeplitz"] = (np.full(200, 2.0), np.full(199, -1.0))
    e = rng.standard_normal(n - 1)
    e[[40, 41, 150]] = 0.0  # decoupled blocks, one of size one
    out["zero_couplings"] = (rng.standard_normal(n), e)
    out["graded"] = (np.logspace(0, -12, 120), 0.3 * np.logspace(0, -12, 119))
    out["tiny_couplings"] = (rng.standard_normal(64), np.full(63, 1e-170))  # e^2 underflows to zero
    out["huge_norm"] = (1e150 * rng.standard_normal(50), 1e150 * rng.standard_normal(49))
    out["small_norm"] = (1e-150 * rng.standard_normal(50), 1e-150 * rng.standard_normal(49))
    out["identical_diagonal_no_coupling"] = (np.full(33, 0.75), np.zeros(32))
    # the spectrum shape of the path: a Marchenko-Pastur bulk with a few large outliers (Lanczos-like tridiagonal of a Gram matrix)
    A = rng.standard_normal((400, 160))
    A[:, :3] *= 6.0
    G = A.T @ A / 400.0
    from scipy.linalg import hessenberg

    H = hessenberg(G)
    out["gram_like"] = (np.diag(H).copy(), np.diag(H, -1).copy())
    return out


@pytest.mark.parametrize("name", sorted(_cases()))
def test_division_free_sturm_count_matches_lapack(name):
    d, e = _cases()[name]
    lam = eigvalsh_tridiagonal(d, e) if len(d) > 1 else d.copy()
    span = max(np.abs(lam).max(), 1e-300)
    # probe points: between neighbouring eigenvalues that are clearly separated, outside the spectrum, and AT diagonal entries
    # (where a leading minor is exactly zero at the first step)
    gaps = np.flatnonzero(np.diff(lam) > 1e-9 * span)
    mids = 0.5 * (lam[gaps] + lam[gaps + 1])
    want_mid = gaps + 1
    got = sturm_count_prod(d, e, mids)
    assert np.array_equal(got, want_mid), (name, np.flatnonzero(got != want_mid)[:5])
    outside = np.array([lam[0] - 0.1 * span - 1e-300, lam[-1] + 0.1 * span + 1e-300])
    assert list(sturm_count_prod(d, e, outside)) == [0, len(d)]
    # a count taken AT a point is allowed to place an eigenvalue within rounding of it on either side, nothing more
    at = np.unique(d)
    c = sturm_count_prod(d, e, at)
    tol = 1e-12 * span
    lo = np.searchsorted(lam, at - tol, side="left")
    hi = np.searchsorted(lam, at + tol, side="right")
    assert np.all((c >= lo) & (c <= hi)), name


def test_nine_section_on_the_model_count_resolves_every_eigenvalue():
    """the kernel's outer loop (nine-section until the interval is at rounding level) on the model count"""
    d, e = _cases()["gram_like"]
    lam = eigvalsh_tridiagonal(d, e)
    n = len(d)
    r = np.abs(np.concatenate([[0.0], e])) + np.abs(np.concatenate([e, [0.0]]))
    glo, ghi = (d - r).min(), (d + r).max()
    eps = 2.220446049250313e-16
    atol = eps * max(abs(glo), abs(ghi))
    for k in (0, 1, n // 2, n - 4, n - 1):
        lo, hi = glo, ghi
        for _ in range(40):
            if hi - lo <= 2 * eps * max(abs(lo), abs(hi)) + atol:
                break
            step = (hi - lo) / 9.0
            xq = lo + np.arange(1, 9) * step
            m = int(np.sum(sturm_count_prod(d, e, xq) <= k))
            lo, hi = (lo if m == 0 else lo + m * step), (hi if m == 8 else lo + (m + 1) * step)
        assert abs(0.5 * (lo + hi) - lam[k]) <= 8 * eps * max(abs(glo), abs(ghi)), k


# ------------------------------------------------------------------------------------------------ split-fp16 products
def split_image_scaled(x):
    """(hi, lo, scale): fp16 pieces of x * scale, scale = the power of two that puts max|x| into [2^13, 2^14) (k_pick_scale)"""
    amax = float(np.abs(x).max())
    ex = int(np.frexp(amax)[1])  # amax = f 2^ex, 1/2 <= f < 1
    scale = np.ldexp(1.0, 14 - ex)
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)  # exact: a power of two
    hi = xs.astype(np.float16)  # round to nearest even, as v_cvt_f16_f32
    lo = (xs - hi.astype(np.float32)).astype(np.float16)  # the residual is exact in fp32
    return hi, lo, scale


def split_product(A, B):
    """A B' from split images: three products of fp16 pieces with fp32 accumulation, unscaled at the end"""
    ah, al, sa = split_image_scaled(A)
    bh, bl, sb = split_image_scaled(B)
    f = np.float32
    acc = ah.astype(f) @ bh.astype(f).T  # every product of two fp16 values is exact in fp32; the sums round in fp32
    acc = acc + ah.astype(f) @ bl.astype(f).T
    acc = acc + al.astype(f) @ bh.astype(f).T
    return acc * f(1.0 / (sa * sb))


def _scaled_count_like(rng, n, k):
    """entries spread like a scaled count matrix: a background near -1e-2 .. -1e-3 and sparse positives up to ~80"""
    X = -np.abs(rng.normal(5e-3, 2e-3, size=(n, k)))
    hit = rng.random((n, k)) < 0.06
    X[hit] = rng.lognormal(0.0, 1.2, size=int(hit.sum()))
    X[0, 0] = 80.0
    return X.astype(np.float32)


def test_split_fp16_product_is_as_accurate_as_the_fp32_product_it_replaces():
    rng = np.random.default_rng(5)
    A = _scaled_count_like(rng, 96, 4096)
    B = _scaled_count_like(rng, 80, 4096)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    fp32 = A @ B.T
    got = split_product(A, B)
    top = np.abs(ref).max()
    err_split = np.abs(got - ref).max() / top
    err_fp32 = np.abs(fp32 - ref).max() / top
    assert err_split < 2e-6
    assert err_split < 2 * err_fp32 + 1e-6  # the bound the device tests assert for the kernels
    # the pieces keep 22 significant bits of every entry above the fp16 floor: the dropped al*bl term is below 2^-22 of a product
    ah, al, sa = split_image_scaled(A)
    back = (ah.astype(np.float64) + al.astype(np.float64)) / sa
    big = np.abs(A) * sa >= 2.0 ** -3  # hi's ulp is then >= 2^-13, its residual >= the fp16 subnormal spacing 2^-24
    assert np.all(np.abs(back[big] - A[big]) <= np.abs(A[big]) * 2.0 ** -21)


def test_split_scale_keeps_the_largest_entry_in_range_and_is_exact():
    rng = np.random.default_rng(6)
    for mag in (1e-6, 3e-2, 1.0, 77.0, 6.5e4, 1e9):
        x = (rng.standard_normal((8, 64)) * mag).astype(np.float32)
        hi, lo, scale = split_image_scaled(x)
        assert np.isfinite(hi.astype(np.float32)).all()
        top = np.abs(x).max() * scale
        assert 2.0 ** 13 <= top < 2.0 ** 14
        assert np.log2(scale) == np.round(np.log2(scale))


# ------------------------------------------------------------------- the rank-2k update of the band reduction from split operands
def _split(xs):
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _pow2_scale(amax):
    return np.float32(np.ldexp(1.0, 14 - int(np.frexp(float(amax))[1])))


def update_from_split(V, Z, two_scales):
    """-(V Z' + Z V') as sbr.hip forms it: P = [V | Z], Q = [-Z | -V] split into fp16 pieces, three products, fp32 sums.
    One shared scale (split_image_pair_scaled) or one per kind of column (split_image_pair_scaled2)."""
    f = np.float32
    if two_scales:
        a, b = _pow2_scale(np.abs(V).max()), _pow2_scale(np.abs(Z).max())
    else:
        a = b = _pow2_scale(max(np.abs(V).max(), np.abs(Z).max()))
    P = np.hstack([V * a, Z * b]).astype(f)
    Q = np.hstack([-Z * b, -V * a]).astype(f)
    ph, pl = _split(P)
    qh, ql = _split(Q)
    acc = ph @ qh.T + ph @ ql.T + pl @ qh.T
    return acc * f(1.0 / (float(a) * float(b)))


@pytest.mark.parametrize("log2_norm", [0, 8, 12, 16, 20])
def test_shared_and_separate_scales_of_the_update_operands(log2_norm):
    """What DESIGN.md section 4 states about the accuracy regime of the split trailing update: with ONE scale for [V | Z] the
    error stays at the fp32 product's level while the entries of Z (~ the norm of the matrix) are below ~2^12 times those of
    the reflectors, and grows linearly beyond; with one scale per kind it does not depend on the norm."""
    rng = np.random.default_rng(17)
    n, k = 512, 64
    V = np.tril(rng.standard_normal((n, k)) / np.sqrt(n), -1).astype(np.float32)
    V[np.arange(k), np.arange(k)] = 1.0  # unit lower trapezoidal reflector block
    Z = (rng.standard_normal((n, k)) * 2.0 ** log2_norm / np.sqrt(n)).astype(np.float32)
    Z[3, 5] = 2.0 ** log2_norm  # the largest entry of Z sets the shared scale
    ref = -(V.astype(np.float64) @ Z.astype(np.float64).T + Z.astype(np.float64) @ V.astype(np.float64).T)
    fp32 = -(V @ Z.T + Z @ V.T)
    top = np.abs(ref).max()
    e32 = np.abs(fp32 - ref).max() / top
    e1 = np.abs(update_from_split(V, Z, False) - ref).max() / top
    e2 = np.abs(update_from_split(V, Z, True) - ref).max() / top
    assert e2 < 4 * e32 + 2e-7, (e2, e32)  # separate scales: always at the fp32 product's level
    if log2_norm <= 12:
        assert e1 < 4 * e32 + 2e-7, (e1, e32)  # the default is as good inside its regime
    if log2_norm >= 20:
        assert e1 > 8 * e2  # ... and visibly worse far outside it, which is why the second form exists


def test_split_gram_of_a_scaled_count_matrix_keeps_the_spectrum():
    """the product bench.py's data / null decompositions start from (session.hip gram_f32 -> split_image_scaled +
    gemm_split_update): a synthetic count matrix through the ORACLE's normalisation, its Gram matrix from split fp16 operands
    and from fp32 operands, spectra against the float64 product"""
    from oracle import sclens_oracle as O
    from sclens_amd.synth import synth_counts

    N, M = 3000, 800
    S = np.asarray(O.logn_scale(O.pre_scale(np.asarray(synth_counts(N, M, seed=5, C=6).todense(), dtype=np.float64))))
    ref = np.linalg.eigvalsh(S.T @ S / N)
    B = S.astype(np.float32)
    hi, lo, sc = split_image_scaled(B)
    H, L = hi.astype(np.float32), lo.astype(np.float32)
    acc = H.T @ H
    acc = acc + H.T @ L
    acc = acc + L.T @ H
    Gs = (acc * np.float32(1.0 / (sc * sc)) / np.float32(N)).astype(np.float64)
    G32 = ((B.T @ B) / np.float32(N)).astype(np.float64)
    e_split = np.abs(np.linalg.eigvalsh(0.5 * (Gs + Gs.T)) - ref).max() / ref[-1]
    e_fp32 = np.abs(np.linalg.eigvalsh(0.5 * (G32 + G32.T)) - ref).max() / ref[-1]
    assert e_split < 2e-7 and e_split < 10 * e_fp32 + 1e-7, (e_split, e_fp32)
