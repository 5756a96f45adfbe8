// Householder tridiagonalisation of a dense symmetric fp32 matrix, device-resident (gfx950).
// Replaces the first phase of the reference's `_get_eigen` (scLENS.jl:375-387), which there is
// cuSOLVER `syevd!` (GPU) / LAPACK `dsyevr` (CPU) on a host matrix.
//
// Blocked (panel width NB) right-looking reduction A = Q T Q^T with the trailing matrix kept in
// FULL symmetric storage. Two kernels per column, no host synchronisation:
//   trd_colA : column update by the panel's reflectors, W-column finalisation of the previous
//              column, partial sums (||x||^2, V^T x, W^T x) per block               (small)
//   trd_colB : Householder scalars, v = x*scale on the fly, u = A_trail * v  (HBM-bound symv, the
//              dominant kernel of the whole path), w' = tau (u - V (W^T v) - W (V^T v)), partial w'^T v
// and per panel one finalize/transposition kernel + one rank-2*NB symmetric MFMA update
// (gemm_f32, lower+mirror so the matrix stays exactly symmetric).
// V^T v of each column is kept (Gst) so the block-reflector T factors need no extra pass over V.
#include "common.h"

namespace scl {

constexpr int NB = 128;       // panel width (also the block-reflector width of the back-transform)
constexpr int RPB_A = 512;    // rows per block in trd_colA
constexpr int ROWS_B = 64;    // rows per block in trd_colB (8 waves x 8 rows)
constexpr int PA_LD = 2 * NB + 1;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TrdArgs {
  float* A;        // n x n, row-major, lda (multiple of 4, padding zero)
  int64_t n, lda;
  float* VWt;      // [2*NB][ldv]: rows 0..NB-1 = V columns, NB..2NB-1 = W columns (each contiguous over matrix rows)
  int64_t ldv;
  float* x;        // [ldv] current column
  double* partA;   // [na][PA_LD]
  double* partB;   // [nblkB]
  double* d;       // [n]
  double* e;       // [n]
  float* tau;      // [n]
  float* Gst;      // [n][NB]: Gst[j][cc] = V[:,cc]^T v_j for cc < c(j)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_colA(TrdArgs a, int64_t j, int c, int nbB_prev) {
  __shared__ float Vj[NB], Wj[NB];
  __shared__ float a_s[RPB_A];
  __shared__ double red[4];
  __shared__ float alpha2_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n, ldv = a.ldv;

  // alpha2 = -1/2 tau_{j-1} (w'^T v) from the previous column's kernel-B partials
  if (c > 0) {
    double s = 0.0;
    for (int b = tid; b < nbB_prev; b += 256) s += a.partB[b];
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) alpha2_s = (float)(-0.5 * (double)a.tau[j - 1] * (red[0] + red[1] + red[2] + red[3]));
    __syncthreads();
  }
  const float alpha2 = (c > 0) ? alpha2_s : 0.f;
  // row j of V and W (W[j][c-1] needs its finalisation: w' + alpha2 * v, with v_j = V[j][c-1])
  for (int cc = tid; cc < c; cc += 256) {
    const float vj = a.VWt[(int64_t)cc * ldv + j];
    float wj = a.VWt[(int64_t)(NB + cc) * ldv + j];
    if (cc == c - 1) wj += alpha2 * vj;
    Vj[cc] = vj;
    Wj[cc] = wj;
  }
  __syncthreads();

  const int64_t i_lo = j + (int64_t)blockIdx.x * RPB_A;
  const int64_t i_hi = (i_lo + RPB_A < n) ? i_lo + RPB_A : n;
  // ---- phase 1: one thread per row
  for (int64_t i = i_lo + tid; i < i_hi; i += 256) {
    float av = a.A[j * a.lda + i];
    if (c > 0) {
      float* wlast = &a.VWt[(int64_t)(NB + c - 1) * ldv + i];
      const float wf = *wlast + alpha2 * a.VWt[(int64_t)(c - 1) * ldv + i];
      if (i != j) *wlast = wf;  // entry j stays raw: every block's prologue re-derives W[j][c-1] from it
      for (int cc = 0; cc < c - 1; ++cc)
        av -= a.VWt[(int64_t)cc * ldv + i] * Wj[cc] + a.VWt[(int64_t)(NB + cc) * ldv + i] * Vj[cc];
      av -= a.VWt[(int64_t)(c - 1) * ldv + i] * Wj[c - 1] + wf * Vj[c - 1];
    }
    a.x[i] = av;
    if (i == j && j == n - 1) a.d[j] = (double)av;  // last diagonal entry (no kernel B for it)
    a_s[i - i_lo] = (i >= j + 2) ? av : 0.f;
  }
  for (int64_t r = (i_hi - i_lo) + tid; r < RPB_A; r += 256) a_s[r] = 0.f;
  __syncthreads();
  // ---- phase 2: one wave per panel column: partial V^T x, W^T x over this block's rows (i >= j+2)
  double* pa = a.partA + (int64_t)blockIdx.x * PA_LD;
  const int nrow = (int)(i_hi - i_lo);
  for (int s = wid; s < 2 * c + 1; s += 4) {
    double acc = 0.0;
    if (s == 2 * c) {
      for (int r = lane; r < nrow; r += 64) acc += (double)a_s[r] * (double)a_s[r];
      acc = wave_sum(acc);
      if (lane == 0) pa[2 * NB] = acc;
    } else {
      const int row = (s < c) ? s : NB + (s - c);
      const float* vp = a.VWt + (int64_t)row * ldv + i_lo;
      for (int r = lane; r < nrow; r += 64) acc += (double)vp[r] * (double)a_s[r];
      acc = wave_sum(acc);
      if (lane == 0) pa[row] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void trd_colB(TrdArgs a, int64_t j, int c, int na) {
  __shared__ double sums[2 * NB + 1];
  __shared__ float tVv[NB], tWv[NB];
  __shared__ float sc_tau, sc_scale;
  __shared__ float us[ROWS_B];
  __shared__ float corr_s[8][ROWS_B];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n, ldv = a.ldv, lda = a.lda;

  // ---- prologue: reduce kernel-A partials (fixed order -> deterministic), Householder scalars
  if (tid < 2 * c + 1) {
    const int row = (tid < c) ? tid : (tid < 2 * c ? NB + (tid - c) : 2 * NB);
    double s = 0.0;
    for (int b = 0; b < na; ++b) s += a.partA[(int64_t)b * PA_LD + row];
    sums[row] = s;
  }
  __syncthreads();
  if (tid == 0) {
    const double alpha = (double)a.x[j + 1];
    const double xn2 = sums[2 * NB];
    double beta, tau, scale;
    if (xn2 == 0.0) {
      beta = alpha; tau = 0.0; scale = 0.0;
    } else {
      beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    sc_tau = (float)tau;
    sc_scale = (float)scale;
    if (blockIdx.x == 0) {
      a.d[j] = (double)a.x[j];
      a.e[j] = beta;
      a.tau[j] = (float)tau;
    }
  }
  __syncthreads();
  const float tau = sc_tau, scale = sc_scale;
  if (tid < c) {
    const float g = (float)(sums[tid] * (double)scale) + a.VWt[(int64_t)tid * ldv + (j + 1)];
    tVv[tid] = g;
    tWv[tid] = (float)(sums[NB + tid] * (double)scale) + a.VWt[(int64_t)(NB + tid) * ldv + (j + 1)];
    if (blockIdx.x == 0) a.Gst[j * NB + tid] = g;
  }
  __syncthreads();

  // ---- symv: 8 rows per wave, all columns >= j+1 (aligned down to 4; v = 0 left of j+1)
  const int64_t r0 = (j + 1) + (int64_t)blockIdx.x * ROWS_B + wid * 8;
  const int64_t c_al = (j + 1) & ~(int64_t)3;
  const float* rp[8];
  bool ok[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    ok[r] = (r0 + r) < n;
    rp[r] = a.A + (ok[r] ? (r0 + r) : (n - 1)) * lda;
  }
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  for (int64_t col = c_al + 4 * lane; col < n; col += 256) {
    f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + col);  // x is padded to a multiple of 4 (zeros)
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t idx = col + e;
      v[e] = (idx == j + 1) ? 1.f : ((idx > j + 1 && idx < n) ? xv[e] * scale : 0.f);
    }
    f32x4 av[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) av[r] = *reinterpret_cast<const f32x4*>(rp[r] + col);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      acc[r] += av[r][0] * v[0] + av[r][1] * v[1] + av[r][2] * v[2] + av[r][3] * v[3];
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float s = wave_sumf(acc[r]);
    if (lane == 0) us[wid * 8 + r] = ok[r] ? s : 0.f;
  }
  // ---- corrections  u_i -= V[i,:] . (W^T v) + W[i,:] . (V^T v): 64 rows x 8 column groups
  {
    const int rr = tid & 63, g = tid >> 6;
    const int64_t i = (j + 1) + (int64_t)blockIdx.x * ROWS_B + rr;
    float p = 0.f;
    if (i < n)
      for (int cc = g; cc < c; cc += 8)
        p += a.VWt[(int64_t)cc * ldv + i] * tWv[cc] + a.VWt[(int64_t)(NB + cc) * ldv + i] * tVv[cc];
    corr_s[g][rr] = p;
  }
  __syncthreads();
  if (tid < 64) {
    const int64_t i = (j + 1) + (int64_t)blockIdx.x * ROWS_B + tid;
    double wv = 0.0;
    if (i < n) {
      float u = us[tid];
#pragma unroll
      for (int g = 0; g < 8; ++g) u -= corr_s[g][tid];
      const float vi = (i == j + 1) ? 1.f : a.x[i] * scale;
      const float w = tau * u;
      a.VWt[(int64_t)(NB + c) * ldv + i] = w;
      a.VWt[(int64_t)c * ldv + i] = vi;
      a.A[j * lda + i] = vi;  // reflector j lives in row j, right of the diagonal
      wv = (double)w * (double)vi;
    }
    wv = wave_sum(wv);
    if (tid == 0) a.partB[blockIdx.x] = wv;
  }
}

// ------------------------------------------------------------------------------------------------
// Panel end: finalise the last W column and write row-major VW = [V|W], WV = [W|V] for rows >= pe.
__global__ __launch_bounds__(256) void trd_panel_finish(TrdArgs a, int64_t pe, int nbB_prev, float* VW,
                                                       float* WV) {
  __shared__ float tile[2 * NB][33];
  __shared__ double red[4];
  __shared__ float alpha2_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double s = 0.0;
  for (int b = tid; b < nbB_prev; b += 256) s += a.partB[b];
  s = wave_sum(s);
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) alpha2_s = (float)(-0.5 * (double)a.tau[pe - 1] * (red[0] + red[1] + red[2] + red[3]));
  __syncthreads();
  const float alpha2 = alpha2_s;
  const int64_t i0 = pe + (int64_t)blockIdx.x * 32;
  // read 2NB x 32 (coalesced over matrix rows)
  for (int idx = tid; idx < 2 * NB * 32; idx += 256) {
    const int row = idx >> 5, r = idx & 31;
    const int64_t i = i0 + r;
    float v = 0.f;
    if (i < a.n) {
      v = a.VWt[(int64_t)row * a.ldv + i];
      if (row == 2 * NB - 1) v += alpha2 * a.VWt[(int64_t)(NB - 1) * a.ldv + i];
    }
    tile[row][r] = v;
  }
  __syncthreads();
  // write 32 x 2NB (coalesced over panel columns); VW/WV row index is relative to pe
  for (int idx = tid; idx < 32 * 2 * NB; idx += 256) {
    const int r = idx / (2 * NB), col = idx % (2 * NB);
    const int64_t i = i0 + r;
    if (i < a.n) {
      const float v = tile[col][r];
      VW[(i - pe) * (2 * NB) + col] = v;
      WV[(i - pe) * (2 * NB) + (col < NB ? col + NB : col - NB)] = v;
    }
  }
}

int sytrd_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev, float* tau_dev) {
  if (n <= 0) return SCLENS_OK;
  if (lda % 4 != 0 || lda < n || (reinterpret_cast<uintptr_t>(A) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "sytrd_f32: A must be 16-byte aligned with lda a multiple of 4");
  StageTimer tm(ctx, "sytrd");
  const int64_t ldv = round_up(n, 64) + 64;
  SCL_WS(ctx, VWt, float, "trd.VWt", 2 * NB * ldv);
  SCL_WS(ctx, x, float, "trd.x", ldv);
  const int64_t naMax = (n + RPB_A - 1) / RPB_A + 1, nbMax = (n + ROWS_B - 1) / ROWS_B + 1;
  SCL_WS(ctx, partA, double, "trd.partA", naMax * PA_LD);
  SCL_WS(ctx, partB, double, "trd.partB", nbMax);
  SCL_WS(ctx, Gst, float, "trd.Gst", n * NB);
  SCL_WS(ctx, VW, float, "trd.VW", n * 2 * NB);
  SCL_WS(ctx, WV, float, "trd.WV", n * 2 * NB);
  SCL_HIP(ctx, hipMemsetAsync(x, 0, sizeof(float) * ldv, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(Gst, 0, sizeof(float) * n * NB, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(tau_dev, 0, sizeof(float) * n, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(e_dev, 0, sizeof(double) * n, ctx->stream));
  TrdArgs a{A, n, lda, VWt, ldv, x, partA, partB, d_dev, e_dev, tau_dev, Gst};
  int nbB_prev = 0;
  for (int64_t p = 0; p < n; p += NB) {
    const int64_t pe = (p + NB < n) ? p + NB : n;
    SCL_HIP(ctx, hipMemsetAsync(VWt, 0, sizeof(float) * 2 * NB * ldv, ctx->stream));
    for (int64_t j = p; j < pe; ++j) {
      const int c = (int)(j - p);
      const int na = (int)((n - j + RPB_A - 1) / RPB_A);
      hipLaunchKernelGGL(trd_colA, dim3(na), dim3(256), 0, ctx->stream, a, j, c, nbB_prev);
      if (j == n - 1) break;
      const int nbB = (int)((n - (j + 1) + ROWS_B - 1) / ROWS_B);
      if (ctx->prof_symv) {
        if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
          for (int q = 0; q < 2; ++q) {
            hipEvent_t ev;
            SCL_HIP(ctx, hipEventCreate(&ev));
            ctx->prof_ev.push_back(ev);
          }
        }
        SCL_HIP(ctx, hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
      }
      hipLaunchKernelGGL(trd_colB, dim3(nbB), dim3(512), 0, ctx->stream, a, j, c, na);
      if (ctx->prof_symv) {
        SCL_HIP(ctx, hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], ctx->stream));
        ctx->prof_used += 2;
        const double nt = (double)(n - j - 1);
        ctx->prof_bytes += 4.0 * nt * nt;  // the trailing matrix, once (sum_j 4 (n-j)^2)
      }
      nbB_prev = nbB;
    }
    if (pe < n) {
      const int64_t nt = n - pe;
      hipLaunchKernelGGL(trd_panel_finish, dim3((unsigned)((nt + 31) / 32)), dim3(256), 0, ctx->stream, a,
                         pe, nbB_prev, VW, WV);
      GemmArgs g{};
      g.P = VW; g.Q = WV; g.C = A + pe * lda + pe;
      g.M = nt; g.N = nt; g.K = 2 * NB;
      g.ldp = 2 * NB; g.ldq = 2 * NB; g.ldc = lda;
      g.alpha = -1.f; g.beta = 1.f; g.q_kcontig = 1; g.lower = 1; g.colabsmax = nullptr;
      SCL_TRY(gemm_f32(ctx, g));
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
