// Householder tridiagonalisation of a dense symmetric fp32 matrix, device-resident (gfx950).
// Replaces the first phase of the reference's `_get_eigen` (scLENS.jl:375-387), which there is
// cuSOLVER `syevd!` (GPU) / LAPACK `dsyevr` (CPU) on a host matrix.
//
// Blocked (panel width NB) right-looking reduction A = Q T Q^T with the trailing matrix kept in
// FULL symmetric storage. Three kernels per column, no host synchronisation:
//   trd_colA : x = column j updated by the panel's reflectors (thread = row x column-group, all loads
//              issued at once), finalisation of the previous W column, per-block partial sums of
//              ||x||^2, V^T x, W^T x                                                   (small, many blocks)
//   trd_colR : one block: fixed-order reduction of those partials, Householder scalars (fp64),
//              V^T v and W^T v                                                         (tiny)
//   trd_colB : v = x*scale on the fly, u = A_trail * v -- the HBM-bound symmetric matrix-vector product,
//              the dominant kernel of the whole sclens() path -- 4 rows per wave, 16 rows per block so a
//              launch has n'/16 >> 256 blocks; then w' = tau (u - V (W^T v) - W (V^T v)) and partial w'^T v
// and per panel one finalize/transposition kernel + one rank-2*NB symmetric MFMA update
// (gemm_f32, lower+mirror so the matrix stays exactly symmetric).
// All reductions run in a fixed order (no float atomics): results are bitwise reproducible.
// V^T v of each column is kept (Gst) so the block-reflector T factors need no extra pass over V.
#include "common.h"

namespace scl {

constexpr int NB = 128;       // panel width (also the block-reflector width of the back-transform)
constexpr int RPB_A = 256;    // rows per block in trd_colA (x 4 column groups = 1024 threads)
constexpr int ROWS_B = 16;    // rows per block in trd_colB (4 waves x 4 rows)
constexpr int PA_LD = 2 * NB + 1;
constexpr int CI_LD = 2 * NB + 4;  // colinfo: [0]=tau [1]=scale, [4..4+NB) = V^T v, [4+NB..4+2NB) = W^T v

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TrdArgs {
  float* A;        // n x n, row-major, lda (multiple of 4, padding finite/zero)
  int64_t n, lda;
  float* VWt;      // [2*NB][ldv]: rows 0..NB-1 = V columns, NB..2NB-1 = W columns (each contiguous over matrix rows)
  int64_t ldv;
  float* x;        // [ldv] current column
  double* partA;   // [na][PA_LD]
  double* partB;   // [nblkB]
  float* colinfo;  // [CI_LD]
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
__global__ __launch_bounds__(1024) void trd_colA(TrdArgs a, int64_t j, int c, int nbB_prev) {
  __shared__ float Vj[NB], Wj[NB];
  __shared__ float a_s[RPB_A];
  __shared__ float part_s[4][RPB_A];
  __shared__ double red[16];
  __shared__ float alpha2_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & (RPB_A - 1), g = tid >> 8;  // row within the block, column group (cc = g mod 4)
  const int64_t n = a.n, ldv = a.ldv;

  // alpha2 = -1/2 tau_{j-1} (w'^T v) from the previous column's kernel-B partials (fixed order)
  if (c > 0) {
    double s = 0.0;
    for (int b = tid; b < nbB_prev; b += 1024) s += a.partB[b];
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < 16; ++w) t += red[w];
      alpha2_s = (float)(-0.5 * (double)a.tau[j - 1] * t);
    }
  }
  // row j of V and W (W[j][c-1] needs its finalisation: w' + alpha2 * v, with v_j = V[j][c-1])
  __syncthreads();
  const float alpha2 = (c > 0) ? alpha2_s : 0.f;
  if (tid < c) {
    const float vj = a.VWt[(int64_t)tid * ldv + j];
    float wj = a.VWt[(int64_t)(NB + tid) * ldv + j];
    if (tid == c - 1) wj += alpha2 * vj;
    Vj[tid] = vj;
    Wj[tid] = wj;
  }
  __syncthreads();

  const int64_t i_lo = j + (int64_t)blockIdx.x * RPB_A;
  const int64_t i = i_lo + r;
  const bool live = i < n;
  // ---- phase 1: thread (row r, group g) accumulates its columns cc = g, g+4, ... ; all loads up front
  float p = 0.f;
  if (live && c > 0) {
    float vv[NB / 4], ww[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      const int cc = g + 4 * q;
      const bool ok = cc < c;
      vv[q] = ok ? a.VWt[(int64_t)cc * ldv + i] : 0.f;
      ww[q] = ok ? a.VWt[(int64_t)(NB + cc) * ldv + i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      const int cc = g + 4 * q;
      if (cc < c) {
        float wv = ww[q];
        if (cc == c - 1) {  // finalise W[:, c-1]; entry j stays raw (every block re-derives W[j][c-1] from it)
          wv += alpha2 * vv[q];
          if (i != j) a.VWt[(int64_t)(NB + cc) * ldv + i] = wv;
        }
        p += vv[q] * Wj[cc] + wv * Vj[cc];
      }
    }
  }
  part_s[g][r] = p;
  __syncthreads();
  if (g == 0) {
    float av = 0.f;
    if (live) {
      av = a.A[j * a.lda + i] - (part_s[0][r] + part_s[1][r] + part_s[2][r] + part_s[3][r]);
      a.x[i] = av;
      if (i == j && j == n - 1) a.d[j] = (double)av;  // last diagonal entry (no kernel B for it)
    }
    a_s[r] = (live && i >= j + 2) ? av : 0.f;
  }
  __syncthreads();
  // ---- phase 2: one wave per panel column: partial V^T x, W^T x, ||x||^2 over this block's rows (i >= j+2)
  double* pa = a.partA + (int64_t)blockIdx.x * PA_LD;
  const int64_t nrow = (i_lo + RPB_A <= n) ? RPB_A : (n - i_lo);
  for (int s = wid; s < 2 * c + 1; s += 16) {
    double acc = 0.0;
    if (s == 2 * c) {
#pragma unroll
      for (int q = 0; q < RPB_A / 64; ++q) acc += (double)a_s[lane + 64 * q] * (double)a_s[lane + 64 * q];
      acc = wave_sum(acc);
      if (lane == 0) pa[2 * NB] = acc;
    } else {
      const int row = (s < c) ? s : NB + (s - c);
      const float* vp = a.VWt + (int64_t)row * ldv + i_lo;
      float t[RPB_A / 64];
#pragma unroll
      for (int q = 0; q < RPB_A / 64; ++q) t[q] = (lane + 64 * q < nrow) ? vp[lane + 64 * q] : 0.f;
#pragma unroll
      for (int q = 0; q < RPB_A / 64; ++q) acc += (double)t[q] * (double)a_s[lane + 64 * q];
      acc = wave_sum(acc);
      if (lane == 0) pa[row] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void trd_colR(TrdArgs a, int64_t j, int c, int na) {
  __shared__ double sums[2 * NB + 1];
  __shared__ float sc_scale;
  const int tid = threadIdx.x;
  const int64_t ldv = a.ldv;
  if (tid < 2 * c + 1) {
    const int row = (tid < c) ? tid : (tid < 2 * c ? NB + (tid - c) : 2 * NB);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 4 <= na; b += 4) {  // fixed order; 4 independent loads in flight
      s0 += a.partA[(int64_t)(b + 0) * PA_LD + row];
      s1 += a.partA[(int64_t)(b + 1) * PA_LD + row];
      s2 += a.partA[(int64_t)(b + 2) * PA_LD + row];
      s3 += a.partA[(int64_t)(b + 3) * PA_LD + row];
    }
    for (; b < na; ++b) s0 += a.partA[(int64_t)b * PA_LD + row];
    sums[row] = (s0 + s1) + (s2 + s3);
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
    sc_scale = (float)scale;
    a.colinfo[0] = (float)tau;
    a.colinfo[1] = (float)scale;
    a.d[j] = (double)a.x[j];
    a.e[j] = beta;
    a.tau[j] = (float)tau;
  }
  __syncthreads();
  if (tid < c) {
    const float scale = sc_scale;
    const float g = (float)(sums[tid] * (double)scale) + a.VWt[(int64_t)tid * ldv + (j + 1)];
    a.colinfo[4 + tid] = g;
    a.colinfo[4 + NB + tid] = (float)(sums[NB + tid] * (double)scale) + a.VWt[(int64_t)(NB + tid) * ldv + (j + 1)];
    a.Gst[j * NB + tid] = g;
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_colB(TrdArgs a, int64_t j, int c) {
  __shared__ float tVv[NB], tWv[NB];
  __shared__ float us[ROWS_B];
  __shared__ float corr_s[16][ROWS_B + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n, ldv = a.ldv, lda = a.lda;
  const float tau = a.colinfo[0], scale = a.colinfo[1];
  if (tid < c) {
    tVv[tid] = a.colinfo[4 + tid];
    tWv[tid] = a.colinfo[4 + NB + tid];
  }

  // ---- symv: 4 rows per wave, all columns >= j+1 (aligned down to 4; v = 0 left of j+1), 2 chunks in flight
  const int64_t rb = (j + 1) + (int64_t)blockIdx.x * ROWS_B;
  const int64_t r0 = rb + wid * 4;
  const int64_t c_al = (j + 1) & ~(int64_t)3;
  const float* rp[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ok[r] = (r0 + r) < n;
    rp[r] = a.A + (ok[r] ? (r0 + r) : (n - 1)) * lda;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto vfrom = [&](const f32x4& xv, int64_t col) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t idx = col + e;
      v[e] = (idx == j + 1) ? 1.f : ((idx > j + 1 && idx < n) ? xv[e] * scale : 0.f);
    }
    return v;
  };
  int64_t col = c_al + 4 * lane;
  for (; col + 256 < n; col += 512) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.x + col);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x + col + 256);
    f32x4 a0[4], a1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0[r] = *reinterpret_cast<const f32x4*>(rp[r] + col);
      a1[r] = *reinterpret_cast<const f32x4*>(rp[r] + col + 256);
    }
    const f32x4 v0 = vfrom(x0, col), v1 = vfrom(x1, col + 256);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r] += a0[r][0] * v0[0] + a0[r][1] * v0[1] + a0[r][2] * v0[2] + a0[r][3] * v0[3];
      acc[r] += a1[r][0] * v1[0] + a1[r][1] * v1[1] + a1[r][2] * v1[2] + a1[r][3] * v1[3];
    }
  }
  if (col < n) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.x + col);  // x is zero-padded past n
    f32x4 a0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = *reinterpret_cast<const f32x4*>(rp[r] + col);
    const f32x4 v0 = vfrom(x0, col);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += a0[r][0] * v0[0] + a0[r][1] * v0[1] + a0[r][2] * v0[2] + a0[r][3] * v0[3];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = wave_sumf(acc[r]);
    if (lane == 0) us[wid * 4 + r] = ok[r] ? s : 0.f;
  }
  __syncthreads();  // us, tVv, tWv visible
  // ---- corrections  u_i -= V[i,:] . (W^T v) + W[i,:] . (V^T v): 16 rows x 16 column groups
  {
    const int rr = tid & 15, g = tid >> 4;
    const int64_t i = rb + rr;
    float p = 0.f;
    if (i < n) {
      float vv[NB / 16], ww[NB / 16];
#pragma unroll
      for (int q = 0; q < NB / 16; ++q) {
        const int cc = g + 16 * q;
        const bool in = cc < c;
        vv[q] = in ? a.VWt[(int64_t)cc * ldv + i] : 0.f;
        ww[q] = in ? a.VWt[(int64_t)(NB + cc) * ldv + i] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < NB / 16; ++q) {
        const int cc = g + 16 * q;
        if (cc < c) p += vv[q] * tWv[cc] + ww[q] * tVv[cc];
      }
    }
    corr_s[g][rr] = p;
  }
  __syncthreads();
  if (tid < 64) {  // wave 0
    const int64_t i = rb + tid;
    double wv = 0.0;
    if (tid < ROWS_B && i < n) {
      float u = us[tid];
#pragma unroll
      for (int g = 0; g < 16; ++g) u -= corr_s[g][tid];
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
  const int64_t ldv = round_up(n, 64) + 512;
  SCL_WS(ctx, VWt, float, "trd.VWt", 2 * NB * ldv);
  SCL_WS(ctx, x, float, "trd.x", ldv);
  const int64_t naMax = (n + RPB_A - 1) / RPB_A + 1, nbMax = (n + ROWS_B - 1) / ROWS_B + 1;
  SCL_WS(ctx, partA, double, "trd.partA", naMax * PA_LD);
  SCL_WS(ctx, partB, double, "trd.partB", nbMax);
  SCL_WS(ctx, colinfo, float, "trd.colinfo", CI_LD);
  SCL_WS(ctx, Gst, float, "trd.Gst", n * NB);
  SCL_WS(ctx, VW, float, "trd.VW", n * 2 * NB);
  SCL_WS(ctx, WV, float, "trd.WV", n * 2 * NB);
  SCL_HIP(ctx, hipMemsetAsync(x, 0, sizeof(float) * ldv, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(Gst, 0, sizeof(float) * n * NB, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(tau_dev, 0, sizeof(float) * n, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(e_dev, 0, sizeof(double) * n, ctx->stream));
  TrdArgs a{A, n, lda, VWt, ldv, x, partA, partB, colinfo, d_dev, e_dev, tau_dev, Gst};
  int nbB_prev = 0;
  for (int64_t p = 0; p < n; p += NB) {
    const int64_t pe = (p + NB < n) ? p + NB : n;
    SCL_HIP(ctx, hipMemsetAsync(VWt, 0, sizeof(float) * 2 * NB * ldv, ctx->stream));
    for (int64_t j = p; j < pe; ++j) {
      const int c = (int)(j - p);
      const int na = (int)((n - j + RPB_A - 1) / RPB_A);
      hipLaunchKernelGGL(trd_colA, dim3(na), dim3(1024), 0, ctx->stream, a, j, c, nbB_prev);
      if (j == n - 1) break;
      hipLaunchKernelGGL(trd_colR, dim3(1), dim3(512), 0, ctx->stream, a, j, c, na);
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
      hipLaunchKernelGGL(trd_colB, dim3(nbB), dim3(256), 0, ctx->stream, a, j, c);
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
