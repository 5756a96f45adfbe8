"""CPU: the algebra behind csrc/gram_bits.hip, restated in numpy and checked against the oracle's scaled matrix.

(a) For a binarised matrix P the Gram matrix of logn_scale(pre_scale(P)) is a weighted co-occurrence product plus rank-one
    terms (the identity in the header of gram_bits.hip), with weights that split exactly into fp16 pieces.
(b) The split-fp16 product a b ~ ah bh + ah bl + al bh of the search statistic keeps 2^-21 relative accuracy per product.
The device kernels are tested against the same references in tests/test_gpu_gram_bits.py."""
import numpy as np

from oracle import sclens_oracle as O
from sclens_amd.synth import synth_counts


def _stats(P):
    """statistics of scale.hip for a 0/1 matrix P (float64): s_i, l_i, std_j, mu_j, cent_j"""
    N = P.shape[0]
    l = np.log1p(1.0 / P.sum(axis=1))
    lg = P * l[:, None]
    std = lg.std(axis=0, ddof=1)
    Z = lg / std
    mu = Z.mean(axis=0)
    l2 = np.sqrt(((Z - mu) ** 2).sum(axis=1))
    s = l2.mean() / l2
    cent = (s[:, None] * (Z - mu)).mean(axis=0)
    return s, l, std, mu, cent, N


def test_binary_gram_identity_and_fp16_weight_split():
    N, M = 700, 300
    X = synth_counts(N, M, seed=N + M, C=5, marker_frac=0.2, marker_sd=1.5).tocsc()
    X.data[:] = 1.0
    S = np.asarray(O.logn_scale(O.pre_scale(X)), dtype=np.float64)
    want = S.T @ S
    P = np.asarray(X.todense(), dtype=np.float64)
    s, l, std, mu, cent, _ = _stats(P)
    a = s * l
    w = a * a
    d = 1.0 / std
    u = d * (P.T @ (a * s))
    S2 = float((s * s).sum())
    # the weights as the kernel holds them: scaled by a power of two into [2^14, 2^15), two fp16 pieces
    scale = 2.0 ** (14 - int(np.floor(np.log2(w.max()))))
    x = w * scale
    w1 = x.astype(np.float16).astype(np.float64)
    w2 = (x - w1).astype(np.float16).astype(np.float64)
    assert np.all(np.isfinite(w1)) and x.max() < 2.0 ** 15
    assert np.abs(x - w1 - w2).max() <= 2.0 ** -21 * x.max()  # 22 significant bits at the top of the range
    # P is 0/1, so P * w1 and P * w2 are exact in fp16: the two MFMA products reproduce C up to fp32 accumulation
    C = (P.T @ (w1[:, None] * P) + P.T @ (w2[:, None] * P)) / scale
    G = np.outer(d, d) * C - np.outer(u, mu) - np.outer(mu, u) + S2 * np.outer(mu, mu) - N * np.outer(cent, cent)
    assert np.abs(G - want).max() < 5e-7 * np.abs(want).max()  # the oracle's closure path rounds l_i and std_j to Float32


def test_split_fp16_product_accuracy():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(20000) * np.exp(rng.normal(0, 3, 20000))).clip(-1, 1).astype(np.float32)
    b = (rng.standard_normal(20000) * np.exp(rng.normal(0, 3, 20000))).clip(-1, 1).astype(np.float32)

    def split(v):
        x = (v * np.float32(4096.0)).astype(np.float32)
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)  # residual exact in fp32
        return hi.astype(np.float64), lo.astype(np.float64)

    ah, al = split(a)
    bh, bl = split(b)
    got = (ah * bh + ah * bl + al * bh) / 4096.0 ** 2
    exact = a.astype(np.float64) * b.astype(np.float64)
    big = np.abs(a) > 1e-3  # entries whose low piece is still a normal fp16 number
    big &= np.abs(b) > 1e-3
    assert np.abs(got - exact)[big].max() / np.abs(exact[big]).max() < 2.0 ** -20
    assert np.all(np.abs(got - exact)[big] <= 2.0 ** -20 * np.abs(exact[big]) + 1e-12)
    # everywhere (small entries lose low bits to fp16 subnormals): 2^-21 |a b| + 2^-36, so a dot product of unit vectors
    # (sum |a_k b_k| <= 1) is off by less than 5e-7 before the fp32 accumulation
    assert np.all(np.abs(got - exact) <= 2.0 ** -21 * np.abs(exact) + 2.0 ** -36)
