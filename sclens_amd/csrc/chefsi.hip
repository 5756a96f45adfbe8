// Leading eigenpairs by Chebyshev-filtered subspace iteration (CheFSI) with Rayleigh-Ritz, device-resident.
//
// Used for the members of the perturbation ensemble (scLENS.jl:771-778): the reference runs a FULL
// eigendecomposition per member (get_eigvec, :489-524) and keeps the first min_pc = ceil(1.5 k) columns (:776).
// Here only those are computed: the block is seeded with the leading eigenvectors of the unperturbed data matrix
// (already known from get_sigev), every sweep is `degree` block products with the Gram matrix (fp32 MFMA GEMM,
// split-K so the 400 MB - 3.6 GB matrix is streamed exactly once per product by >= 256 blocks) and one b x b
// Rayleigh-Ritz problem solved on the host in fp64 (Cholesky + cyclic Jacobi). If the residuals do not reach the
// tolerance the caller falls back to the full solver, so correctness never depends on convergence here.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

namespace scl {

// out = a1 * (sum_s part[s] - cc * cur) - a2 * prev      (elementwise over rows x cols; prev may be null when a2 == 0)
__global__ __launch_bounds__(256) void k_cheb_step(const float* __restrict__ part, int S, int64_t slab,
                                                   const float* __restrict__ cur, const float* __restrict__ prev,
                                                   float* __restrict__ out, int64_t rows, int64_t cols, int64_t ld,
                                                   float a1, float cc, float a2) {
  const int64_t r = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t o = r * ld + c;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += part[(int64_t)q * slab + o];  // fixed order
  float v = a1 * (s - cc * cur[o]);
  if (a2 != 0.f) v -= a2 * prev[o];
  out[o] = v;
}

// out[i] = sum_s part[s*slab + i]
__global__ void k_sum_slabs(const float* __restrict__ part, int S, int64_t slab, float* __restrict__ out, int64_t cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += part[(int64_t)q * slab + i];
  out[i] = s;
}

// part[z][r][c] = 0 for the slabs of a split-K sum that a product with fewer slices does not write
__global__ __launch_bounds__(256) void k_zero_slabs(float* __restrict__ part, int64_t slab, int64_t cols, int64_t ld) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < cols) part[(int64_t)blockIdx.z * slab + (int64_t)blockIdx.y * ld + c] = 0.f;
}

// res[r] = || Y[r,:] - theta[r] X[r,:] ||_2
__global__ __launch_bounds__(256) void k_resid_rows(const float* __restrict__ Y, const float* __restrict__ X,
                                                    const float* __restrict__ theta, int64_t cols, int64_t ld,
                                                    float* __restrict__ res) {
  __shared__ double sw[4];
  const int64_t r = blockIdx.x;
  const float th = theta[r];
  double s = 0.0;
  for (int64_t c = threadIdx.x; c < cols; c += 256) {
    const double dlt = (double)Y[r * ld + c] - (double)th * (double)X[r * ld + c];
    s += dlt * dlt;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) res[r] = (float)sqrt(sw[0] + sw[1] + sw[2] + sw[3]);
}

namespace {

// ---- small dense fp64 helpers on the host (b <= 128) ---------------------------------------------------------
bool cholesky_lower(std::vector<double>& G, int b) {  // in place, row-major; returns false if not SPD
  for (int j = 0; j < b; ++j) {
    double dsum = G[j * b + j];
    for (int k = 0; k < j; ++k) dsum -= G[j * b + k] * G[j * b + k];
    if (!(dsum > 0.0)) return false;
    const double l = std::sqrt(dsum);
    G[j * b + j] = l;
    for (int i = j + 1; i < b; ++i) {
      double s = G[i * b + j];
      for (int k = 0; k < j; ++k) s -= G[i * b + k] * G[j * b + k];
      G[i * b + j] = s / l;
    }
    for (int i = 0; i < j; ++i) G[i * b + j] = 0.0;
  }
  return true;
}
void invert_lower(const std::vector<double>& L, std::vector<double>& Li, int b) {  // Li = L^-1 (lower)
  Li.assign((size_t)b * b, 0.0);
  for (int j = 0; j < b; ++j) {
    Li[j * b + j] = 1.0 / L[j * b + j];
    for (int i = j + 1; i < b; ++i) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s += L[i * b + k] * Li[k * b + j];
      Li[i * b + j] = -s / L[i * b + i];
    }
  }
}
// cyclic Jacobi: C (symmetric, row-major) -> eigenvalues ev, eigenvectors as the COLUMNS of U
void jacobi_eig(std::vector<double> C, int b, std::vector<double>& ev, std::vector<double>& U) {
  U.assign((size_t)b * b, 0.0);
  for (int i = 0; i < b; ++i) U[i * b + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = 0; i < b; ++i)
      for (int j = 0; j < b; ++j) (i == j ? dia : off) += C[i * b + j] * C[i * b + j];
    if (off <= 1e-30 * dia) break;
    for (int p = 0; p < b - 1; ++p)
      for (int q = p + 1; q < b; ++q) {
        const double apq = C[p * b + q];
        // threshold Jacobi: after the first Rayleigh-Ritz step the projected matrix is nearly diagonal (the block rows
        // are Ritz vectors), so almost every rotation is skipped
        if (std::fabs(apq) <= 1e-17 * std::sqrt(std::fabs(C[p * b + p] * C[q * b + q])) || std::fabs(apq) < 1e-300) continue;
        const double tau = (C[q * b + q] - C[p * b + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < b; ++k) {  // columns p, q
          const double ckp = C[k * b + p], ckq = C[k * b + q];
          C[k * b + p] = cs * ckp - sn * ckq;
          C[k * b + q] = sn * ckp + cs * ckq;
        }
        for (int k = 0; k < b; ++k) {  // rows p, q
          const double cpk = C[p * b + k], cqk = C[q * b + k];
          C[p * b + k] = cs * cpk - sn * cqk;
          C[q * b + k] = sn * cpk + cs * cqk;
        }
        for (int k = 0; k < b; ++k) {
          const double ukp = U[k * b + p], ukq = U[k * b + q];
          U[k * b + p] = cs * ukp - sn * ukq;
          U[k * b + q] = sn * ukp + cs * ukq;
        }
      }
  }
  ev.resize(b);
  for (int i = 0; i < b; ++i) ev[i] = C[i * b + i];
}

}  // namespace

int topk_chefsi(Ctx* ctx, const float* A, int64_t n, int64_t lda, int m, int m_strict, int b, const float* X0t, int64_t ldx,
                const double* theta0, double* w_desc, float* Zt, int64_t ldz, int* converged, int* iters, const float* Bop,
                int64_t Kop, int64_t ldb, float div, double tail_gap, int tail_free) {
  *converged = 0;
  if (iters) *iters = 0;
  if (m <= 0 || b < m || b > 128 || b > n) return ctx->fail(SCLENS_ERR_ARG, "topk_chefsi: bad block sizes");
  StageTimer tm(ctx, "chefsi");
  hipStream_t st = ctx->stream;
  const int64_t ld = round_up(n, 32);
  // split-K so that one block product launches >= ~320 blocks (implicit operator: the contraction over Kop is much longer
  // than a tile row is wide, so at least four slices)
  const int64_t tiles_n = (n + 127) / 128;
  int S = (int)((320 + tiles_n - 1) / tiles_n);
  S = std::max(Bop ? 4 : 1, std::min(S, 8));
  // implicit operator: both products stream Bop once and are HBM-bound; S (second product) and S1 (first) cut the contractions
  // (measured at cfg4, 20 perturbations: S/S1 = 4/1 7.7 s, 9/3 7.5 s, 8/2 8.1 s, 16/6 25.8 s -- the products already run at
  // about half of the HBM roofline and more slices do not help.)
  const int S1 = 1;
  const int64_t ldt = Bop ? round_up(Kop, 32) : 0;
  float *Tb = nullptr, *Tpart = nullptr;
  // Round 5: the implicit operator on the fp16 matrix cores. Both products contract b = 64 rows against the whole 12 GB scaled matrix:
  // 2 b n Kop flop per pass -- on the fp32 matrix cores that takes as long as the HBM pass itself and the two did not overlap (4.0-4.4 ms
  // per pass where the bytes take 2.4). From split images (the same 4 bytes per entry, three matrix instructions at 16x the rate) on a
  // 64 x 256 tile (gemm_split_skinny: the 256 x 256 split kernel was measured first and is NOT faster for 64 rows, it pays its tile's
  // operand delivery: profiles/r05_pipe_masks_chefsi_split.log) the products stream Bop at HBM rate. Needs the image of Bop AND of its
  // transpose (the first product contracts over the rows of Bop), built once per call; the block rows (V, then T = V Bop) get ONE
  // POWER-OF-TWO SCALE PER ROW -- a Chebyshev sweep spreads the row norms by up to 1e5 -- multiplied back into the rows of the result
  // (exact). `precision = 0` / option chefsi_split = 0: fp32 products.
  const bool split_op = Bop && ctx->opt.split() && ctx->opt.chefsi_split != 0;
  int SS1 = 1, SS2 = 1;  // split-K slices of the two split products
  void *imgB = nullptr, *imgBt = nullptr, *imgV = nullptr, *imgT = nullptr;
  float *scB = nullptr, *one = nullptr, *rsV = nullptr, *rsT = nullptr;
  if (Bop) {
    Tb = static_cast<float*>(ctx->workspace("che.T", sizeof(float) * (size_t)b * ldt));
    if (!Tb) return SCLENS_ERR_OOM;
    if (split_op) {
      // output tiles of 256 columns, two workgroups per CU: slices so that the grid fills whole rounds of the 512 slots
      auto pick = [](int64_t tiles, int64_t K, int max_s) {
        int best = 1;
        double best_cost = 1e300;
        for (int q = 1; q <= max_s; ++q) {
          if (q > 1 && K / q < 2048) break;
          const double rounds = (double)((tiles * q + 511) / 512);
          const double cost = rounds / q + 0.01 * q;
          if (cost < best_cost) { best_cost = cost; best = q; }
        }
        return best;
      };
      SS1 = ctx->opt.chefsi_split_s1 > 0 ? (int)ctx->opt.chefsi_split_s1 : pick((Kop + 255) / 256, n, 8);
      SS2 = ctx->opt.chefsi_split_s2 > 0 ? (int)ctx->opt.chefsi_split_s2 : pick((n + 255) / 256, Kop, 16);
      if (SS2 > S) SS2 = S;  // `part` holds S slabs
      imgB = ctx->workspace("che.imgB", split_image_bytes(n, Kop));
      imgBt = ctx->workspace("che.imgBt", split_image_bytes(Kop, n));
      imgV = ctx->workspace("che.imgV", split_image_bytes(b, n));
      imgT = ctx->workspace("che.imgT", split_image_bytes(b, Kop));
      scB = static_cast<float*>(ctx->workspace("che.scB", 8 * sizeof(float)));
      rsV = static_cast<float*>(ctx->workspace("che.rsV", 2 * b * sizeof(float)));
      if (!imgB || !imgBt || !imgV || !imgT || !scB || !rsV) return SCLENS_ERR_OOM;
      one = scB + 4;
      rsT = rsV + b;
      SCL_TRY(split_image_scaled(ctx, Bop, n, Kop, ldb, imgB, scB));
      SCL_TRY(split_image_transposed(ctx, Bop, n, Kop, ldb, imgBt, scB));
      const float h1 = 1.f;
      SCL_HIP(ctx, hipMemcpyAsync(one, &h1, sizeof(float), hipMemcpyHostToDevice, st));
      SCL_HIP(ctx, hipStreamSynchronize(st));  // `h1` is a local
    }
    const int nslab1 = split_op ? SS1 : S1;
    if (nslab1 > 1) {
      Tpart = static_cast<float*>(ctx->workspace("che.Tp", sizeof(float) * (size_t)nslab1 * b * ldt));
      if (!Tpart) return SCLENS_ERR_OOM;
    }
  }
  const int64_t slab = (int64_t)b * ld;
  SCL_WS(ctx, X, float, "che.X", slab);
  SCL_WS(ctx, Y1, float, "che.Y1", slab);
  SCL_WS(ctx, Y2, float, "che.Y2", slab);
  SCL_WS(ctx, AX, float, "che.AX", slab);
  SCL_WS(ctx, part, float, "che.part", (int64_t)(S + 1) * slab);  // + one slab for the deflation term of locked pairs
  const int SG = 32;  // split-K of the two b x b products
  constexpr int NLMAX = 32;  // at most this many leading pairs are locked
  SCL_WS(ctx, gpart, float, "che.gpart", (int64_t)SG * b * std::max(b, NLMAX));
  SCL_WS(ctx, Lk, float, "che.Lk", (int64_t)NLMAX * ld);    // locked vectors (rows)
  SCL_WS(ctx, LkT, float, "che.LkT", (int64_t)NLMAX * ld);  // - theta_i * locked vector i
  SCL_WS(ctx, C1, float, "che.C1", (int64_t)b * NLMAX);
  SCL_WS(ctx, nthd, float, "che.nth", NLMAX);
  SCL_WS(ctx, GH, float, "che.GH", 2 * b * b);
  SCL_WS(ctx, Wd, float, "che.W", b * b);
  SCL_WS(ctx, thd, float, "che.theta", b);
  SCL_WS(ctx, resd, float, "che.res", b);
  SCL_HIP(ctx, hipMemsetAsync(X, 0, sizeof(float) * slab, st));
  SCL_HIP(ctx, hipMemcpy2DAsync(X, sizeof(float) * ld, X0t, sizeof(float) * ldx, sizeof(float) * n, b,
                                hipMemcpyDeviceToDevice, st));
  SCL_HIP(ctx, hipMemsetAsync(Y1, 0, sizeof(float) * slab, st));
  SCL_HIP(ctx, hipMemsetAsync(Y2, 0, sizeof(float) * slab, st));

  // part[s] rows r0.. = V[r0..] * A (slice s of the contraction); rows below r0 (locked pairs) are left alone
  auto block_product = [&](const float* V0, int r0 = 0) -> int {
    const float* V = V0 + (int64_t)r0 * ld;
    float* Pr = part + (int64_t)r0 * ld;
    const int rows = b - r0;
    if (split_op) {
      // T' = (D V) Bop from the images of V (row scales D) and of Bop'; then (E T') Bop' from the images of T' (row scales E) and of Bop,
      // 64 block rows at a time; the row scales and 1 / div are multiplied back by the second product's epilogue
      SCL_TRY(split_image_rows(ctx, V, rows, n, ld, imgV, rsV));
      const size_t rbV = split_image_bytes(1, n), rbT = split_image_bytes(1, Kop);
      for (int q0 = 0; q0 < rows; q0 += 64) {
        const int qr = std::min(64, rows - q0);
        SCL_TRY(gemm_split_skinny(ctx, static_cast<const char*>(imgV) + (size_t)q0 * rbV, one, rsV + q0, qr, imgBt, scB, Kop, n,
                                  (SS1 > 1 ? Tpart : Tb) + (int64_t)q0 * ldt, ldt, SS1, round_up((n + SS1 - 1) / SS1, 32), (int64_t)b * ldt, 1.f));
      }
      if (SS1 > 1)
        hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)(((int64_t)rows * ldt + 255) / 256)), dim3(256), 0, st, Tpart, SS1, (int64_t)b * ldt, Tb,
                           (int64_t)rows * ldt);
      SCL_TRY(split_image_rows(ctx, Tb, rows, Kop, ldt, imgT, rsT));
      for (int q0 = 0; q0 < rows; q0 += 64) {
        const int qr = std::min(64, rows - q0);
        SCL_TRY(gemm_split_skinny(ctx, static_cast<const char*>(imgT) + (size_t)q0 * rbT, one, rsT + q0, qr, imgB, scB, n, Kop,
                                  Pr + (int64_t)q0 * ld, ld, SS2, round_up((Kop + SS2 - 1) / SS2, 32), slab, 1.f / div));
      }
      if (SS2 < S)  // the slabs SS2 .. S - 1 of `part` are not written by this product: zero for the consumers, which sum S of them
        hipLaunchKernelGGL(k_zero_slabs, dim3((unsigned)((n + 255) / 256), (unsigned)rows, (unsigned)(S - SS2)), dim3(256), 0, st,
                           Pr + (int64_t)SS2 * slab, slab, n, ld);
      SCL_HIP(ctx, hipGetLastError());
      return SCLENS_OK;
    }
    if (Bop) {
      // A = Bop Bop' / div is never formed: V A = (V Bop) Bop' / div, two products that stream Bop once each
      // (2 * 2 b n Kop flop against n^2 Kop for the Gram matrix: cheaper below ~n / (4 b) applications)
      GemmArgs g1{};
      g1.P = V; g1.Q = Bop; g1.C = S1 > 1 ? Tpart : Tb;
      g1.M = rows; g1.N = Kop; g1.K = n;
      g1.ldp = ld; g1.ldq = ldb; g1.ldc = ldt;
      g1.alpha = 1.f; g1.beta = 0.f; g1.q_kcontig = 0; g1.lower = 0; g1.colabsmax = nullptr;
      if (S1 > 1) {
        g1.splits = S1; g1.k_chunk = round_up((n + S1 - 1) / S1, 16); g1.c_split_off = (int64_t)b * ldt;
      }
      SCL_TRY(gemm_f32(ctx, g1));
      if (S1 > 1)
        hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)(((int64_t)b * ldt + 255) / 256)), dim3(256), 0, st, Tpart, S1, (int64_t)b * ldt, Tb,
                           (int64_t)b * ldt);
      GemmArgs g2{};
      g2.P = Tb; g2.Q = Bop; g2.C = Pr;
      g2.M = rows; g2.N = n; g2.K = Kop;
      g2.ldp = ldt; g2.ldq = ldb; g2.ldc = ld;
      g2.alpha = 1.f / div; g2.beta = 0.f; g2.q_kcontig = 1; g2.lower = 0; g2.colabsmax = nullptr;
      g2.splits = S; g2.k_chunk = round_up((Kop + S - 1) / S, 32); g2.c_split_off = slab;
      return gemm_f32(ctx, g2);
    }
    GemmArgs g{};
    g.P = V; g.Q = A; g.C = Pr;
    g.M = rows; g.N = n; g.K = n;
    g.ldp = ld; g.ldq = lda; g.ldc = ld;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;  // A symmetric: rows are K-contiguous
    g.splits = S; g.k_chunk = round_up((n + S - 1) / S, 16); g.c_split_off = slab;
    return gemm_f32(ctx, g);
  };
  auto small_product = [&](const float* P, const float* Q, float* out) -> int {  // out[b x b] = P Q^T
    GemmArgs g{};
    g.P = P; g.Q = Q; g.C = gpart;
    g.M = b; g.N = b; g.K = n;
    g.ldp = ld; g.ldq = ld; g.ldc = b;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
    g.splits = SG; g.k_chunk = round_up((n + SG - 1) / SG, 16); g.c_split_off = (int64_t)b * b;
    SCL_TRY(gemm_f32(ctx, g));
    hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((b * b + 255) / 256)), dim3(256), 0, st, gpart, SG, (int64_t)b * b,
                       out, (int64_t)b * b);
    return SCLENS_OK;
  };
  // Locking. Once the leading `nlock` pairs have reached their targets they stay in the basis but are no longer filtered, and
  // the filter runs on the deflated operator A - sum_i theta_i v_i v_i' (their eigenvalues move to 0, inside the damped
  // interval): the degree is then set by the largest UNLOCKED Ritz value. Without it the spread limit below pins the degree at
  // 2 as long as a signal 20x above the bulk edge is in the block, and the pairs at the edge need a dozen sweeps
  // (cfg4: 13 sweeps of 3 products per member -> 2 sweeps + one of degree 16).
  // slab S of `part` rows r0.. = -(V[r0..] Lk') diag(theta) Lk
  int nlock = 0;
  auto deflate = [&](const float* V0, int r0) -> int {
    const int rows = b - r0;
    GemmArgs g{};
    g.P = V0 + (int64_t)r0 * ld; g.Q = Lk; g.C = gpart;
    g.M = rows; g.N = nlock; g.K = n;
    g.ldp = ld; g.ldq = ld; g.ldc = nlock;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
    g.splits = SG; g.k_chunk = round_up((n + SG - 1) / SG, 16); g.c_split_off = (int64_t)b * NLMAX;
    SCL_TRY(gemm_f32(ctx, g));
    hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((rows * nlock + 255) / 256)), dim3(256), 0, st, gpart, SG, (int64_t)b * NLMAX, C1,
                       (int64_t)rows * nlock);
    GemmArgs h{};
    h.P = C1; h.Q = LkT; h.C = part + (int64_t)S * slab + (int64_t)r0 * ld;
    h.M = rows; h.N = n; h.K = nlock;
    h.ldp = nlock; h.ldq = ld; h.ldc = ld;
    h.alpha = 1.f; h.beta = 0.f; h.q_kcontig = 0; h.lower = 0; h.colabsmax = nullptr;
    return gemm_f32(ctx, h);
  };

  float *B0 = X, *B1 = Y1, *B2 = Y2;  // B0 = current basis, B1/B2 = scratch
  std::vector<double> theta(theta0, theta0 + b);
  std::vector<float> hGH(2 * (size_t)b * b), hW((size_t)b * b), hth(b), hres(b);
  std::vector<double> G((size_t)b * b), H((size_t)b * b), Li, Cm((size_t)b * b), T1((size_t)b * b), ev, U;
  const int max_degree = 16, max_outer = 32;
  // residual targets: the first m_strict pairs (the signals, whose eigenvectors are consumed) 1e-3 * theta; the
  // remaining ones up to m only feed eigenvalues and the matching argmax: 3e-3 * theta (Ritz value error ~ res^2 / gap
  // to the spectrum outside the block, well below the 3e-4 relative tolerance of the parity tests)
  const double tol_rel = 1e-3, tol_rel_tail = 5e-3, tol_gap = 2e-3;
  // tail pairs: gap-aware target only on request (tail_gap > 0; the caller asks for it when a tail vector is CONSUMED, see
  // api.sclens / session option "chefsi_tail_gap_milli"); the context option chefsi_tail_gap_micro overrides
  const double tail_env = ctx->opt.chefsi_tail_gap_micro >= 0 ? 1e-6 * (double)ctx->opt.chefsi_tail_gap_micro : -1.0;
  const double tol_gap_tail = tail_env >= 0.0 ? tail_env : (tail_gap > 0.0 ? tail_gap : 1e30);
  // tail_free: only the strict pairs decide convergence, locking and the filter's degree; the pairs m_strict .. m-1 come back as
  // whatever Ritz pairs the block holds when the strict ones are done (orthonormal to them by the Rayleigh-Ritz step, eigenvalues
  // to a few per cent). For callers that can PROVE they do not consume them (session_robustness' matching certificate).
  const int m_conv = tail_free ? std::min(m, std::max(1, m_strict)) : m;
  for (int outer = 0; outer < max_outer; ++outer) {
    // ---- Chebyshev filter of `degree`: damp [0, theta_b], normalise at theta_1
    const double lo = 0.0, cut = std::max(theta[b - 1], 1e-12 * theta[0]);
    const double e = 0.5 * (cut - lo), c = 0.5 * (cut + lo);
    const int r0 = nlock;  // rows r0 .. b-1 are filtered
    const double sigma1 = e / (theta[r0] - c);
    double sigma = sigma1;
    // The filter scales Ritz direction q by T_d(x_q), x_q = (theta_q - c)/e. Between two re-orthogonalisations the
    // spread T_d(x_1) / T_d(1) must stay far below 1/eps32, or the rows collapse onto the leading eigenvectors.
    int degree = 2;
    {
      const double ach = std::acosh(std::max(1.0 + 1e-9, (theta[r0] - c) / e));
      while (degree < max_degree && std::cosh((degree + 1) * ach) <= 1e5) ++degree;
    }
    if (r0 > 0 && outer >= 1) {
      // no more than the sweep needs: the wanted pair q is amplified by cosh(d acosh(x_q)) against the edge of the block;
      // ask for 4x the largest remaining residual / target ratio, at the pair that the filter separates least
      double need = 1.0, ach_min = 1e300;
      for (int q = r0; q < m_conv; ++q) {
        double lim = (q < m_strict ? tol_rel : tol_rel_tail) * std::fabs(theta[q]);
        if (q >= m_strict && b > m) lim = std::min(lim, std::max(tol_gap_tail * (theta[q] - theta[b - 1]), 4e-6 * std::fabs(theta[0])));
        need = std::max(need, (double)hres[q] / std::max(lim, 1e-300));
        ach_min = std::min(ach_min, std::acosh(std::max(1.0 + 1e-9, (theta[q] - c) / e)));
      }
      if (ach_min < 1e299 && ach_min > 0.0) {
        const int want = (int)std::ceil(std::acosh(std::max(1.0, 4.0 * need)) / ach_min);
        degree = std::max(2, std::min(degree, want));
      }
    }
    const dim3 fgrid((unsigned)((n + 255) / 256), (unsigned)(b - r0));
    const int64_t roff = (int64_t)r0 * ld;
    const int Sx = S + (r0 > 0 ? 1 : 0);
    if (r0 > 0) {  // this sweep's copies of the locked vectors: Lk = rows 0 .. r0-1 of the basis, LkT = -theta_i Lk_i
      SCL_HIP(ctx, hipMemcpyAsync(Lk, B0, sizeof(float) * (size_t)r0 * ld, hipMemcpyDeviceToDevice, st));
      SCL_HIP(ctx, hipMemcpyAsync(LkT, B0, sizeof(float) * (size_t)r0 * ld, hipMemcpyDeviceToDevice, st));
      std::vector<float> nth(r0);
      for (int q = 0; q < r0; ++q) nth[q] = (float)(-theta[q]);
      SCL_HIP(ctx, hipMemcpyAsync(nthd, nth.data(), sizeof(float) * r0, hipMemcpyHostToDevice, st));
      SCL_HIP(ctx, hipStreamSynchronize(st));  // `nth` is a local
      SCL_TRY(scale_rows_f32(ctx, LkT, r0, n, ld, nthd));
    }
    float *xprev = B0, *xcur = B1, *xnext = B2;
    SCL_TRY(block_product(B0, r0));
    if (r0 > 0) SCL_TRY(deflate(B0, r0));
    hipLaunchKernelGGL(k_cheb_step, fgrid, dim3(256), 0, st, part + roff, Sx, slab, B0 + roff, (const float*)nullptr, xcur + roff,
                       (int64_t)(b - r0), n, ld, (float)(sigma1 / e), (float)c, 0.f);
    for (int i = 2; i <= degree; ++i) {
      const double sn = 1.0 / (2.0 / sigma1 - sigma);
      SCL_TRY(block_product(xcur, r0));
      if (r0 > 0) SCL_TRY(deflate(xcur, r0));
      hipLaunchKernelGGL(k_cheb_step, fgrid, dim3(256), 0, st, part + roff, Sx, slab, xcur + roff, xprev + roff, xnext + roff,
                         (int64_t)(b - r0), n, ld, (float)(2.0 * sn / e), (float)c, (float)(sigma * sn));
      float* t = xprev; xprev = xcur; xcur = xnext; xnext = t;
      sigma = sn;
    }
    if (r0 > 0 && xcur != B0)  // the locked rows re-join the block unfiltered (B0's copies were never written)
      SCL_HIP(ctx, hipMemcpyAsync(xcur, B0, sizeof(float) * (size_t)r0 * ld, hipMemcpyDeviceToDevice, st));
    // ---- Rayleigh-Ritz on span(xcur): rows rescaled to unit length first, G = V V^T, H = V (A V)^T
    SCL_TRY(normalize_rows_f32(ctx, xcur, b, n, ld));
    SCL_TRY(block_product(xcur));
    hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, part, S, slab, AX, slab);
    SCL_TRY(small_product(xcur, xcur, GH));
    SCL_TRY(small_product(xcur, AX, GH + b * b));
    SCL_HIP(ctx, hipMemcpyAsync(hGH.data(), GH, sizeof(float) * 2 * b * b, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < b; ++i)
      for (int j = 0; j < b; ++j) {
        G[i * b + j] = 0.5 * ((double)hGH[i * b + j] + (double)hGH[j * b + i]);
        H[i * b + j] = 0.5 * ((double)hGH[b * b + i * b + j] + (double)hGH[b * b + j * b + i]);
      }
    if (!cholesky_lower(G, b)) {  // numerically rank deficient block -> let the caller fall back
      if (ctx->opt.debug) fprintf(stderr, "[chefsi] outer %d: Gram of the block not SPD\n", outer);
      break;
    }
    invert_lower(G, Li, b);
    // Cm = Li H Li^T
    for (int i = 0; i < b; ++i)
      for (int j = 0; j < b; ++j) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s += Li[i * b + k] * H[k * b + j];
        T1[i * b + j] = s;
      }
    for (int i = 0; i < b; ++i)
      for (int j = 0; j < b; ++j) {
        double s = 0.0;
        for (int k = 0; k <= j; ++k) s += T1[i * b + k] * Li[j * b + k];
        Cm[i * b + j] = s;
      }
    jacobi_eig(Cm, b, ev, U);
    std::vector<int> ord(b);
    for (int i = 0; i < b; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return ev[x] > ev[y]; });
    // new rows = W V with W = U^T Li (rows ordered by descending Ritz value)
    for (int r = 0; r < b; ++r) {
      const int col = ord[r];
      theta[r] = ev[col];
      hth[r] = (float)ev[col];
      for (int j = 0; j < b; ++j) {
        double s = 0.0;
        for (int k = j; k < b; ++k) s += U[k * b + col] * Li[k * b + j];
        hW[r * b + j] = (float)s;
      }
    }
    SCL_HIP(ctx, hipMemcpyAsync(Wd, hW.data(), sizeof(float) * b * b, hipMemcpyHostToDevice, st));
    SCL_HIP(ctx, hipMemcpyAsync(thd, hth.data(), sizeof(float) * b, hipMemcpyHostToDevice, st));
    {  // new basis = W V -> xnext ; W (A V) -> xprev ; residuals
      GemmArgs g{};
      g.P = Wd; g.Q = xcur; g.C = xnext;
      g.M = b; g.N = n; g.K = b;
      g.ldp = b; g.ldq = ld; g.ldc = ld;
      g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
      SCL_TRY(gemm_f32(ctx, g));
      g.Q = AX; g.C = xprev;
      SCL_TRY(gemm_f32(ctx, g));
      hipLaunchKernelGGL(k_resid_rows, dim3((unsigned)b), dim3(256), 0, st, xprev, xnext, thd, n, ld, resd);
      B0 = xnext; B1 = xprev; B2 = xcur;
    }
    SCL_HIP(ctx, hipMemcpyAsync(hres.data(), resd, sizeof(float) * b, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
    if (iters) *iters = outer + 1;
    if (ctx->opt.debug) {
      fprintf(stderr, "[chefsi] outer %d deg %d cut %.4g theta:", outer, degree, cut);
      for (int q = 0; q < std::min(b, 14); ++q) fprintf(stderr, " %.5g", theta[q]);
      fprintf(stderr, " ... %.5g | res/theta:", theta[b - 1]);
      for (int q = 0; q < std::min(m, 14); ++q) fprintf(stderr, " %.2e", hres[q] / std::fabs(theta[q]));
      fprintf(stderr, "\n");
    }
    // eigenvector error ~ residual / gap. The strict pairs (the signals, whose vectors enter the robustness products) are held to
    // tol_gap * the gap to the neighbouring Ritz values. The tail pairs (up to m = ceil(1.5 k): candidates of the `argmax`
    // matching, scLENS.jl:788) sit at the edge of the bulk, where neighbouring gaps are ~1e-3 theta and individual vectors are
    // not determined even in exact arithmetic; what the Rayleigh-Ritz step cannot repair is contamination from OUTSIDE the
    // block, angle ~ residual / (theta_q - theta_out) with theta_out <= the smallest Ritz value of the block, so the tail is
    // held to tol_gap_tail * that gap. Neither goes below the fp32 floor of a residual (~ eps32 theta_1).
    auto target = [&](int q) {
      double lim = (q < m_strict ? tol_rel : tol_rel_tail) * std::fabs(theta[q]);
      const double floor32 = 4e-6 * std::fabs(theta[0]);
      if (q < m_strict) {
        double gap = (q + 1 < b) ? theta[q] - theta[q + 1] : std::fabs(theta[q]);
        if (q > 0) gap = std::min(gap, theta[q - 1] - theta[q]);
        lim = std::min(lim, std::max(tol_gap * gap, floor32));
      } else if (b > m) {
        lim = std::min(lim, std::max(tol_gap_tail * (theta[q] - theta[b - 1]), floor32));
      }
      return lim;
    };
    bool ok = true;
    for (int q = 0; q < m_conv; ++q) ok = ok && ((double)hres[q] <= target(q));
    if (ok && outer >= 1) { *converged = 1; break; }
    // lock the leading run of pairs that have reached their targets (from the second sweep on); a locked pair that drifts
    // above ten times its target unlocks everything
    if (outer >= 1 && ctx->opt.chefsi_lock) {
      bool drift = false;
      for (int q = 0; q < nlock; ++q) drift = drift || (double)hres[q] > 10.0 * target(q);
      if (drift) {
        nlock = 0;
      } else {
        int nl = 0;
        while (nl < m_conv && nl < NLMAX && nl < b - 8 && (double)hres[nl] <= target(nl)) ++nl;
        nlock = std::max(nlock, nl);
      }
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  if (!*converged) return SCLENS_OK;
  for (int q = 0; q < m; ++q) w_desc[q] = theta[q];
  SCL_HIP(ctx, hipMemcpy2DAsync(Zt, sizeof(float) * ldz, B0, sizeof(float) * ld, sizeof(float) * n, m,
                                hipMemcpyDeviceToDevice, st));
  SCL_TRY(normalize_rows_f32(ctx, Zt, m, n, ldz));
  return SCLENS_OK;
}

}  // namespace scl
