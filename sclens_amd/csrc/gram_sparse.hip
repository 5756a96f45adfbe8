// Sparse-structured Gram matrix of a COUNT-VALUED scaled matrix (the identity the reference's own normalisation uses for
// the cells' norms, scLENS.jl:601-603 / :688-690, applied to the Gram product of :332-361).
//
// The scaled matrix of logn_scale / the inline twin is sparse + rank two:
//     B_ij = c d_i (Z_ij - mu_j) - cent_j ,   Z_ij = lg_ij / std_j (non-zero only where a count is stored), d_i = 1 / l_i, c = mean(l)
// hence, with u_ij = Z_ij d_i (fp32, sparse), t_j = sum_i d_i^2 Z_ij, D2 = sum_i d_i^2,
//     (B'B)_jk = c^2 ( S_jk - mu_k t_j - mu_j t_k + D2 mu_j mu_k ) - N cent_j cent_k ,        S = U'U  (sparse x sparse -> dense).
// S needs sum_i r_i^2 / 2 multiply-adds (r_i = stored entries of cell i: 4.5e11 at 100 000 x 30 000, a hundredth of the dense
// product's 4.5e13), but they scatter over a dense n x n result. Here the result is cut into 128 x 128 tiles of gene pairs; one
// workgroup of 16 waves owns one tile as 16 384 accumulators in LDS and walks ALL cells, one cell per lane: the cell's entries inside
// the tile's two gene blocks are two short runs of its CSR row (found through a per-block offset table, boffT[b][cell], coalesced
// across the lanes), the second run is held in registers, and every pair of the two runs is one fp32 product added to its
// accumulator by an LDS atomic. The accumulators are 64-bit FIXED POINT (product x 2^40, rounded once): integer addition is
// associative, so the sum does not depend on the order in which the waves arrive -- the result is bitwise reproducible, which a
// floating-point atomic would not give -- and it is exact to 2^-41 per product, i.e. the contraction itself is more accurate than
// an fp32 GEMM's (the operands are fp32: no operand narrower than the reference's SGEMM). The rank-two terms and the scaling are
// applied per entry in fp64 when the tile is written (lower tiles + mirror: exactly symmetric).
#include "common.h"
#include "pattern.h"

namespace scl {
namespace {

constexpr int SG_TB = 128;                 // genes per block: a tile is SG_TB x SG_TB accumulators of 8 bytes = 128 KB of LDS
constexpr int SG_THREADS = 1024;           // 16 waves share the tile
constexpr int SG_RUN = 16;                 // entries of the second run held in registers at a time
constexpr double SG_SCALE = 1099511627776.0;  // 2^40: |u| <= 1, so a sum over 2^22 cells stays below 2^62

__device__ __forceinline__ double sg_wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// u (CSR slot order) = (float)(Z_ij d_i); one wave per cell. lg is in CSC slot order (gathered through csr2csc), or recomputed from the
// CSR companion copy of the values exactly as k_row_norms does.
__global__ __launch_bounds__(256) void k_sg_u(PatternDev p, const float* __restrict__ val, const double* __restrict__ tgc, int f32path,
                                              const double* __restrict__ lg, const double* __restrict__ stdv, const double* __restrict__ l2,
                                              float* __restrict__ u) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= p.N) return;
  const double d = 1.0 / l2[row];
  if (p.base_val_csr) {
    const float* vc = val + p.nU;
    const double t = tgc[row];
    const float inv = 1.0f / (float)t;
    for (int64_t q = p.rowptr[row] + lane; q < p.rowptr[row + 1]; q += 64) {
      const float v = vc[q];
      float o = 0.f;
      if (v != 0.f) {
        const double l = f32path ? (double)log1pf(inv * v) : log1p((double)v / t);
        o = (float)((l / stdv[p.csrcol[q]]) * d);
      }
      u[q] = o;
    }
  } else {
    for (int64_t q = p.rowptr[row] + lane; q < p.rowptr[row + 1]; q += 64) {
      const int64_t pos = p.csr2csc[q];
      u[q] = (val[pos] != 0.f) ? (float)((lg[pos] / stdv[p.csrcol[q]]) * d) : 0.f;
    }
  }
}

// boffT[b][row] = number of entries of the row with gene < b * SG_TB (b = 0 .. nb): binary search in the row's ascending gene list
__global__ void k_sg_offsets(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ csrcol, int64_t N, int nb,
                             int32_t* __restrict__ boffT) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (row >= N) return;
  const int64_t base = rowptr[row];
  int lo = 0, hi = (int)(rowptr[row + 1] - base);
  const int32_t key = b * SG_TB;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (csrcol[base + mid] < key) lo = mid + 1;
    else hi = mid;
  }
  boffT[(int64_t)b * N + row] = lo;
}

// t_j = sum_i d_i^2 Z_ij (CSC view, one wave per gene); the slots of a union pattern that hold no value contribute lg = 0
__global__ __launch_bounds__(256) void k_sg_t(PatternDev p, const double* __restrict__ lg, const double* __restrict__ stdv,
                                              const double* __restrict__ l2, double* __restrict__ t) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  double s = 0.0;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64) {
    const double d = 1.0 / l2[p.row[q]];
    s += lg[q] * d * d;
  }
  s = sg_wsum(s);
  if (lane == 0) t[col] = s / stdv[col];
}
// out[0] = sum_i d_i^2; out[1] = the mean cell norm c = lsum[0] / n_all (1 when lsum is null: the caller scales)
__global__ __launch_bounds__(1024) void k_sg_d2(const double* __restrict__ l2, int64_t n, const double* __restrict__ lsum, double n_all,
                                                double* __restrict__ out) {
  __shared__ double sw[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double d = 1.0 / l2[i];
    s += d * d;
  }
  s = sg_wsum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double q = 0.0;
    for (int w = 0; w < 16; ++w) q += sw[w];
    out[0] = q;
    out[1] = lsum ? lsum[0] / n_all : 1.0;
  }
}

struct SgArgs {
  const int64_t* rowptr;
  const int32_t* csrcol;
  const float* u;
  const int32_t* boffT;
  int64_t N, M;
  int nb;
  float* A;
  int64_t lda;
  const double *t, *mu, *cent, *d2;  // cent may be null (no cent term); d2[0] = D2, d2[1] = c
  double alpha, beta;                // A (+)= alpha c^2 (S - mu t' - t mu' + D2 mu mu') - beta cent cent'
  int accumulate;
};

__global__ __launch_bounds__(SG_THREADS) void k_sg_tile(SgArgs a) {
  extern __shared__ unsigned long long acc[];
  // tile index -> (J, K), K <= J
  int J = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((int64_t)(J + 1) * (J + 2) / 2 <= (int64_t)blockIdx.x) ++J;
  while ((int64_t)J * (J + 1) / 2 > (int64_t)blockIdx.x) --J;
  const int K = (int)((int64_t)blockIdx.x - (int64_t)J * (J + 1) / 2);
  for (int i = threadIdx.x; i < SG_TB * SG_TB; i += SG_THREADS) acc[i] = 0ull;
  __syncthreads();
  const int32_t* oJ0 = a.boffT + (int64_t)J * a.N;
  const int32_t* oJ1 = oJ0 + a.N;
  const int32_t* oK0 = a.boffT + (int64_t)K * a.N;
  const int32_t* oK1 = oK0 + a.N;
  const bool diag = J == K;
  for (int64_t i = threadIdx.x; i < a.N; i += SG_THREADS) {  // one cell per lane; a wave covers 64 consecutive cells
    const int64_t rp = a.rowptr[i];
    const int64_t a0 = rp + oJ0[i], a1 = rp + oJ1[i];
    int64_t b0 = rp + oK0[i];
    const int64_t b1 = rp + oK1[i];
    if (a0 == a1) continue;
    for (; b0 < b1; b0 += SG_RUN) {  // the K run in chunks held in registers
      int cb[SG_RUN];
      float ub[SG_RUN];
#pragma unroll
      for (int e = 0; e < SG_RUN; ++e) {
        const bool in = b0 + e < b1;
        cb[e] = in ? a.csrcol[b0 + e] - K * SG_TB : -1;
        ub[e] = in ? a.u[b0 + e] : 0.f;
      }
      for (int64_t qa = a0; qa < a1; ++qa) {
        const float ua = a.u[qa];
        if (ua == 0.f) continue;
        const int ca = a.csrcol[qa] - J * SG_TB;
        unsigned long long* row = acc + ca * SG_TB;
#pragma unroll
        for (int e = 0; e < SG_RUN; ++e) {
          // (diagonal tiles: gene pairs with k <= j only -- the genes of a row ascend, so this is a prefix of the run)
          if (cb[e] >= 0 && ub[e] != 0.f && (!diag || cb[e] <= ca)) {
            const long long v = __double2ll_rn((double)(ua * ub[e]) * SG_SCALE);
            atomicAdd(row + cb[e], (unsigned long long)v);
          }
        }
      }
    }
  }
  __syncthreads();
  const double alpha = a.alpha * a.d2[1] * a.d2[1];
  const double d2 = a.d2[0];
  for (int idx = threadIdx.x; idx < SG_TB * SG_TB; idx += SG_THREADS) {
    const int r = idx / SG_TB, c = idx % SG_TB;
    const int64_t j = (int64_t)J * SG_TB + r, k = (int64_t)K * SG_TB + c;
    if (j >= a.M || k >= a.M || (diag && c > r)) continue;
    const double S = (double)(long long)acc[idx] * (1.0 / SG_SCALE);
    double v = alpha * (S - a.mu[k] * a.t[j] - a.mu[j] * a.t[k] + d2 * a.mu[j] * a.mu[k]);
    if (a.cent) v -= a.beta * a.cent[j] * a.cent[k];
    float o = (float)v;
    if (a.accumulate) o += a.A[j * a.lda + k];
    a.A[j * a.lda + k] = o;
    if (j != k) a.A[k * a.lda + j] = o;
  }
}

}  // namespace

// sum_i r_i^2 / 2 over the rows of a CSR view: the multiply-adds of the sparse form (what decides between it and the dense product)
__global__ __launch_bounds__(1024) void k_sg_macs(const int64_t* __restrict__ rowptr, int64_t n, double* __restrict__ out) {
  __shared__ double sw[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double r = (double)(rowptr[i + 1] - rowptr[i]);
    s += 0.5 * r * r;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double q = 0.0;
    for (int w = 0; w < 16; ++w) q += sw[w];
    out[0] = q;
  }
}
int gram_sparse_macs(Ctx* ctx, const PatternDev& p, double* macs) {
  SCL_WS(ctx, d, double, "gs.macs", 2);
  hipLaunchKernelGGL(k_sg_macs, dim3(1), dim3(1024), 0, ctx->stream, p.rowptr, p.N, d);
  SCL_HIP(ctx, hipGetLastError());
  SCL_HIP(ctx, hipMemcpyAsync(macs, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

// A (n = M genes, lda, zero padded outside) (+)= alpha c^2 (U'U - mu t' - t mu' + D2 mu mu') - beta cent cent'
//   p / val: the matrix (CSR view of the pattern + its values); tgc, lg, stdv, mu, l2: its statistics (scale_stats / chunk_dense);
//   lsum != nullptr: c = lsum[0] / n_all (device scalar: the sum of the cells' norms), else c = 1; cent == nullptr: no cent term.
// Exactly symmetric, bitwise reproducible. Needs p.nU < 2^31.
int gram_sparse(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* tgc, const double* lg, const double* stdv,
                const double* mu, const double* l2, const double* cent, const double* lsum, double n_all, double alpha, double beta,
                float* A, int64_t lda, bool accumulate) {
  StageTimer tm(ctx, "gram");
  const int64_t N = p.N, M = p.M;
  if (N <= 0 || M <= 0 || p.nU >= 0x7FFFFFFFll || !p.rowptr || !p.csrcol) return ctx->fail(SCLENS_ERR_ARG, "gram_sparse: bad pattern");
  const int nb = (int)((M + SG_TB - 1) / SG_TB);
  SCL_WS(ctx, u, float, "gs.u", p.nU);
  SCL_WS(ctx, boffT, int32_t, "gs.boff", (int64_t)(nb + 1) * N);
  SCL_WS(ctx, t, double, "gs.t", M);
  SCL_WS(ctx, d2, double, "gs.d2", 2);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_sg_u, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc, f32path, lg, stdv, l2, u);
  hipLaunchKernelGGL(k_sg_offsets, dim3((unsigned)((N + 255) / 256), (unsigned)(nb + 1)), dim3(256), 0, st, p.rowptr, p.csrcol, N, nb, boffT);
  hipLaunchKernelGGL(k_sg_t, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, lg, stdv, l2, t);
  hipLaunchKernelGGL(k_sg_d2, dim3(1), dim3(1024), 0, st, l2, N, lsum, n_all, d2);
  SCL_HIP(ctx, hipGetLastError());
  if (!accumulate) {  // the padding rows / columns beyond M stay zero
    SCL_HIP(ctx, hipMemsetAsync(A, 0, sizeof(float) * (size_t)M * lda, st));
  }
  const int lds = SG_TB * SG_TB * (int)sizeof(unsigned long long);
  SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(k_sg_tile), lds));
  SgArgs a{p.rowptr, p.csrcol, u, boffT, N, M, nb, A, lda, t, mu, cent, d2, alpha, beta, accumulate ? 1 : 0};
  const int64_t tiles = (int64_t)nb * (nb + 1) / 2;
  hipLaunchKernelGGL(k_sg_tile, dim3((unsigned)tiles), dim3(SG_THREADS), lds, st, a);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
