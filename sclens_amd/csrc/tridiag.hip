// Householder tridiagonalisation of a dense symmetric fp32 matrix, device-resident (gfx950).
// Replaces the first phase of the reference's `_get_eigen` (scLENS.jl:375-387), which there is
// cuSOLVER `syevd!` (GPU) / LAPACK `dsyevr` (CPU) on a host matrix.
//
// Blocked (panel width NB) right-looking reduction A = Q T Q^T; the trailing matrix is kept in full
// symmetric storage but the matrix-vector product reads only its LOWER half. Two kernels per column,
// no host synchronisation, every reduction in a fixed order (no float atomics: bitwise reproducible):
//   trd_colA : (1) finishes the previous column: u = A v is assembled from the partial buffers of trd_colB,
//              w = tau (u - V (W^T v) - W (V^T v)) - 1/2 tau (w^T v) v is written as W[:, c-1];
//              (2) x = column j of A updated by the panel's reflectors; (3) per-block partial sums of
//              ||x||^2, V^T x, W^T x. Thread = (row, column group); all global loads are issued up front.
//   (the Householder scalars need ||x||^2 = a grid-wide reduction; the matrix-vector product does not wait for it: it
//    runs on the UN-normalised column, u = a1 + scale * A x~, and the next trd_colA applies `scale`. One spare block of
//    trd_colB's grid reduces the norm partials and writes beta / tau / scale; V^T v, W^T v are reduced in the prologue of
//    the next trd_colA. Two launches per column.)
//   trd_colB : the HBM-bound symmetric matrix-vector product S = A_trail x~ (x~ = the column without its first entry) --
//              the dominant kernel of the whole sclens() path. Work unit = (strip of 32 rows) x (segment of
//              1024 columns) of the lower trapezoid; each of the 4 waves owns one 256-column chunk for all 32
//              rows, so its transposed contribution (u_c += A[r][c] v_r) is complete in registers and is stored
//              once (colpart[strip][c]); row dot products are reduced across the waves in LDS (rowpart[seg][r]).
//              Each matrix element of the lower half is read exactly once: 2 n'^2 B per column instead of 4 n'^2.
// Per panel: one transposition kernel + one rank-2*NB symmetric MFMA update (gemm_f32, lower+mirror).
// V^T v of each column is kept (Gst) so the block-reflector T factors need no extra pass over V.
#include "common.h"

#include <chrono>
#include <condition_variable>
#include <mutex>

namespace scl {

constexpr int NB = 128;        // panel width (also the block-reflector width of the back-transform)
constexpr int RPB_A = 128;     // rows per block in trd_colA (x 8 column groups = 1024 threads)
constexpr int NG_A = 8;        // column groups in trd_colA
constexpr int RS = 32;         // strip height of trd_colB
constexpr int CH = 1;          // 256-column chunks per wave in trd_colB
constexpr int SEG = 1024 * CH; // segment width of trd_colB (4 waves x CH x 256 columns)
constexpr int PR = 16;         // rows loaded per pass in trd_colB (registers vs. loads in flight)
constexpr int PA_LD = 2 * NB + 1;
// colinfo: [0]=tau [1]=scale [2]=(V^T v).(W^T v), [4..4+NB) = V^T v, [4+NB..4+2NB) = W^T v
constexpr int CI_LD = 2 * NB + 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TrdArgs {
  float* A;         // n x n, row-major, lda (multiple of 4, padding finite/zero)
  int64_t n, lda;
  float* VWt;       // [2*NB][ldv]: rows 0..NB-1 = V columns, NB..2NB-1 = W columns (each contiguous over matrix rows)
  int64_t ldv;
  float* x;         // [ldv] current column
  double* partA;    // [2][PA_LD][na_ld]  partial sums of trd_colA (one column per block), ping-pong by column parity
  int64_t na_ld;
  double* partB;    // [nstrips * nsegmax] partial u^T v per trd_colB block (0 for idle blocks)
  float* colinfo;   // [CI_LD]
  float* rowpart;   // [nsegmax][ldv]   row-dot partials of trd_colB, indexed by absolute row
  float* colpart;   // [nstripmax][ldv] transposed partials of trd_colB, indexed by absolute column
  double* d;        // [n]
  double* e;        // [n]
  float* tau;       // [n]
  float* Gst;       // [n][NB]: Gst[j][cc] = V[:,cc]^T v_j for cc < c(j)
  float* VW;        // [n][2NB] row-major [V|W] of the finished panel (operands of the rank-2NB update)
  float* WV;        // [n][2NB] row-major [W|V]
};

// The kernels take an array of argument sets and blockIdx.z selects the member; the host launches ONE member. (Round 2 had a
// host-side rendezvous that merged the column steps of concurrent decompositions into shared launches; measured no faster than
// independent streams at the three members the path runs -- 212 vs 190 ms per matrix at n = 10^4 -- and removed in round 3.)
constexpr int TRD_MAXB = 8;
struct TrdBatch {
  TrdArgs a[TRD_MAXB];
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

// geometry of trd_colB for the column that produced the partial buffers (rows/cols >= jj+1)
__device__ __forceinline__ int nseg_of_strip(int64_t jj, int64_t s, int64_t n) {
  const int64_t c_al = (jj + 1) & ~(int64_t)3;
  int64_t cend = (jj + 1) + (s + 1) * RS;  // exclusive end of the strip's diagonal block
  if (cend > n) cend = n;
  return (int)((cend - c_al + SEG - 1) / SEG);
}

// u_i = sum_g rowpart[g][i] + sum_{s > strip(i)} colpart[s][i]  over the share (part, nparts) of the terms
__device__ __forceinline__ float gather_u(const TrdArgs& a, int64_t jj, int64_t i, int part, int nparts) {
  const int64_t n = a.n, ldv = a.ldv;
  const int64_t si = (i - (jj + 1)) / RS;
  const int64_t nstrip = (n - (jj + 1) + RS - 1) / RS;
  const int nseg = nseg_of_strip(jj, si, n);
  float acc = 0.f;
  // both lists are walked in a fixed order; 16 independent loads in flight
  for (int g0 = part; g0 < nseg; g0 += 16 * nparts) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int g = g0 + u * nparts;
      t[u] = (g < nseg) ? a.rowpart[(int64_t)g * ldv + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += t[u];
  }
  for (int64_t s0 = si + 1 + part; s0 < nstrip; s0 += 16 * nparts) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t s = s0 + (int64_t)u * nparts;
      t[u] = (s < nstrip) ? a.colpart[s * ldv + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += t[u];
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// 32 values per lane -> sums over the 64 lanes; lane l ends with the total of value index vidx(l) in v[0].
__device__ __forceinline__ float reduce32_over_wave(float (&v)[32], int lane) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {  // offset 32: lanes < 32 keep 0..15, lanes >= 32 keep 16..31
    const bool hi = lane & 32;
    const float send = hi ? v[k] : v[k + 16];
    const float keep = hi ? v[k + 16] : v[k];
    v[k] = keep + __shfl_xor(send, 32);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool hi = lane & 16;
    const float send = hi ? v[k] : v[k + 8];
    const float keep = hi ? v[k + 8] : v[k];
    v[k] = keep + __shfl_xor(send, 16);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool hi = lane & 8;
    const float send = hi ? v[k] : v[k + 4];
    const float keep = hi ? v[k + 4] : v[k];
    v[k] = keep + __shfl_xor(send, 8);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool hi = lane & 4;
    const float send = hi ? v[k] : v[k + 2];
    const float keep = hi ? v[k + 2] : v[k];
    v[k] = keep + __shfl_xor(send, 4);
  }
  {
    const bool hi = lane & 2;
    const float send = hi ? v[0] : v[1];
    const float keep = hi ? v[1] : v[0];
    v[0] = keep + __shfl_xor(send, 2);
  }
  v[0] += __shfl_xor(v[0], 1);
  return v[0];
}
// value index held by `lane` after reduce32_over_wave
__device__ __forceinline__ int reduce32_index(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

// ------------------------------------------------------------------------------------------------
// mode 0: full column step for column j (panel-local index c).  mode 1: only finish W[:, c-1] for rows >= j
// (panel end: c = NB, j = pe).
__global__ __launch_bounds__(1024) void trd_colA(TrdBatch bt, int64_t j, int c, int nbB_prev, int na_prev, int mode) {
  const TrdArgs a = bt.a[blockIdx.z];
  __shared__ float Vj[NB], Wj[NB], tVp[NB], tWp[NB];
  __shared__ float a_s[RPB_A];
  __shared__ float part_s[NG_A][RPB_A], part2_s[NG_A][RPB_A], upart_s[NG_A][RPB_A];
  __shared__ double red[16];
  __shared__ float redf[16];
  __shared__ float alpha2_s, wj_s;
  __shared__ double psums_s[2 * NB];
  __shared__ float prod_s[NB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & (RPB_A - 1), g = tid >> 7;  // row within the block, column group (cc = g mod NG_A)
  const int64_t n = a.n, ldv = a.ldv;
  const int cp = c - 1;  // previous panel column (the one whose W is finished here); valid when c > 0
  const int64_t i_lo = j + (int64_t)blockIdx.x * RPB_A;
  const int64_t i = i_lo + r;
  const bool live = i < n;

  // ---------------- all independent global loads first
  double pb = 0.0;
  if (c > 0)
    for (int b = tid; b < nbB_prev; b += 1024) pb += a.partB[b];
  float vj = 0.f, wj = 0.f;
  if (tid < cp) {  // columns < c-1 of V and W are final; column c-1 is completed below (V[j][c-1] = v_j = 1)
    vj = a.VWt[(int64_t)tid * ldv + j];
    wj = a.VWt[(int64_t)(NB + tid) * ldv + j];
  }
  if (tid == cp) vj = 1.f;
  const float ajj = (c > 0) ? a.A[j * a.lda + j] : 0.f;  // a1_j: first column of the previous trailing matrix, row j
  const float tau_p = (c > 0) ? a.colinfo[0] : 0.f;
  const float scale_p = (c > 0) ? a.colinfo[1] : 0.f;
  // partial sums V^T x, W^T x of the previous column (its trd_colA wrote them): 4 lanes per sum, fixed-order tree
  double psum = 0.0;
  if (c > 0) {
    const int sidx = tid >> 2, q4 = tid & 3;
    const bool on = sidx < 2 * cp;
    const int row = (sidx < cp) ? sidx : NB + (sidx - cp);
    const double* pr = a.partA + (int64_t)((j - 1) & 1) * PA_LD * a.na_ld + (int64_t)(on ? row : 0) * a.na_ld;
    for (int b0 = q4; b0 < na_prev; b0 += 64) {
      double t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = (on && b0 + 4 * u < na_prev) ? pr[b0 + 4 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) psum += t[u];
    }
    psum += __shfl_xor(psum, 1);
    psum += __shfl_xor(psum, 2);
  }
  float vv[NB / NG_A], ww[NB / NG_A];
#pragma unroll
  for (int q = 0; q < NB / NG_A; ++q) {
    const int cc = g + NG_A * q;
    vv[q] = (live && cc < cp) ? a.VWt[(int64_t)cc * ldv + i] : 0.f;
    ww[q] = (live && cc < cp) ? a.VWt[(int64_t)(NB + cc) * ldv + i] : 0.f;
  }
  // row j of A: the column being reduced now AND a1 = first column of the previous trailing matrix (u = a1 + scale S)
  const float arow = (live && g == 0) ? a.A[j * a.lda + i] : 0.f;
  const float xold = (live && g == 0 && c > 0) ? a.x[i] : 0.f;  // previous column's x (v = x*scale, v_j = 1)
  float up = 0.f, uj = 0.f;
  if (c > 0) {
    if (live) up = gather_u(a, j - 1, i, g, NG_A);
    uj = gather_u(a, j - 1, j, tid, 1024);  // row j's u, spread over the whole block
  }

  // ---------------- alpha2 (fixed-order block reduction), row j of V/W incl. the finished W[j][c-1]
  if (c > 0) {
    pb = wave_sum(pb);
    uj = wave_sumf(uj);
    if (lane == 0) { red[wid] = pb; redf[wid] = uj; }
    if (tid < c) { Vj[tid] = vj; Wj[tid] = wj; }
    {  // publish the reduced sums: sum index sidx -> psums_s
      const int sidx = tid >> 2;
      if ((tid & 3) == 0 && sidx < 2 * cp) psums_s[sidx] = psum;
    }
    __syncthreads();
    if (tid < cp) {  // V^T v and W^T v of the previous column: scale * (V^T x) + V[j][:]  (v = x*scale, v_{j} = 1)
      const float gv = (float)(psums_s[tid] * (double)scale_p) + Vj[tid];
      const float gw = (float)(psums_s[cp + tid] * (double)scale_p) + Wj[tid];
      tVp[tid] = gv;
      tWp[tid] = gw;
      prod_s[tid] = gv * gw;
      if (blockIdx.x == 0) a.Gst[(j - 1) * NB + tid] = gv;
    }
    __syncthreads();
    if (wid == 0) {  // wave 0: p2 of row j = sum_{cc<cp} V[j][cc] tWp[cc] + W[j][cc] tVp[cc]
      float p2 = 0.f;
      for (int cc = lane; cc < cp; cc += 64) p2 += Vj[cc] * tWp[cc] + Wj[cc] * tVp[cc];
      p2 = wave_sumf(p2);
      if (lane == 0) {
        double utv = 0.0;
        float ujs = 0.f;
        for (int w = 0; w < 16; ++w) { utv += red[w]; ujs += redf[w]; }
        float dotvw_p = 0.f;
        for (int q = 0; q < cp; ++q) dotvw_p += prod_s[q];  // fixed order
        // u = a1 + scale S with S = A x~ from trd_colB (x~ = x without its first entry):
        //   u^T v = A[j][j] + 2 scale S_j + scale^2 (x~^T S)          (a1^T x~ = S_j by symmetry)
        const double sp = (double)scale_p;
        const double utv_full = (double)ajj + 2.0 * sp * (double)ujs + sp * sp * utv;
        const double wtv = (double)tau_p * (utv_full - 2.0 * (double)dotvw_p);
        const float alpha2 = (float)(-0.5 * (double)tau_p * wtv);
        alpha2_s = alpha2;
        const float u_j = ajj + scale_p * ujs;
        wj_s = tau_p * (u_j - p2) + alpha2;  // W[j][c-1]  (v_j = 1)
      }
    }
    __syncthreads();
    if (tid == 0) Wj[cp] = wj_s;
    __syncthreads();
  }
  const float alpha2 = (c > 0) ? alpha2_s : 0.f;

  // ---------------- per (row, group): partial p1 (x update) and p2 (W finish) over cc < c-1
  float p1 = 0.f, p2 = 0.f;
  if (live && c > 0) {
#pragma unroll
    for (int q = 0; q < NB / NG_A; ++q) {
      const int cc = g + NG_A * q;
      if (cc < cp) {
        p1 += vv[q] * Wj[cc] + ww[q] * Vj[cc];
        p2 += vv[q] * tWp[cc] + ww[q] * tVp[cc];
      }
    }
  }
  part_s[g][r] = p1;
  part2_s[g][r] = p2;
  upart_s[g][r] = up;
  __syncthreads();
  if (g == 0) {
    float av = 0.f;
    if (live) {
      float p1t = 0.f;
      if (c > 0) {
        float u = 0.f, p2t = 0.f, p1s = 0.f;
#pragma unroll
        for (int gg = 0; gg < NG_A; ++gg) { u += upart_s[gg][r]; p2t += part2_s[gg][r]; p1s += part_s[gg][r]; }
        const float vl = (i == j) ? 1.f : xold * scale_p;  // v of the previous column
        const float w = tau_p * ((arow + scale_p * u) - p2t) + alpha2 * vl;
        a.VWt[(int64_t)(NB + cp) * ldv + i] = w;  // finished W[i][c-1] (row j included: nobody re-reads it raw)
        a.VWt[(int64_t)cp * ldv + i] = vl;        // V[i][c-1]
        a.A[(j - 1) * a.lda + i] = vl;            // reflector j-1 lives in row j-1, right of the diagonal
        p1t = p1s + vl * Wj[cp] + w * Vj[cp];
      }
      if (mode == 0) {
        av = arow - p1t;
        a.x[i] = av;
        if (i == j && j == n - 1) a.d[j] = (double)av;  // last diagonal entry (no kernel B for it)
      }
    }
    a_s[r] = (live && mode == 0 && i >= j + 2) ? av : 0.f;
  }
  if (mode != 0) return;
  __syncthreads();
  // ---------------- partial V^T x, W^T x, ||x||^2 over this block's rows (i >= j+2); sums s = wid + 16*q
  double* pa = a.partA + (int64_t)(j & 1) * PA_LD * a.na_ld + blockIdx.x;  // partA[j&1][row * na_ld + block]
  const int64_t na_ld = a.na_ld;
  const int64_t nrow = (i_lo + RPB_A <= n) ? RPB_A : (n - i_lo);
  // all loads of the wave issued at once, fp32 products over this block's 128 rows, one 32-value halving tree over the
  // wave (32 shuffles instead of 17 x 6 double shuffles), fp64 only across blocks
  const int nsum = 2 * c + 1;
  {
    float vals[32];
    float xs[RPB_A / 64];
#pragma unroll
    for (int q = 0; q < RPB_A / 64; ++q) xs[q] = a_s[lane + 64 * q];
    float t[17][RPB_A / 64];
#pragma unroll
    for (int qq = 0; qq < 17; ++qq) {
      const int sidx = wid + 16 * qq;
      const int row = (sidx < c) ? sidx : NB + (sidx - c);
      const bool in = sidx < 2 * c;
      const float* vp = a.VWt + (int64_t)(in ? row : 0) * ldv + i_lo;
#pragma unroll
      for (int q = 0; q < RPB_A / 64; ++q) t[qq][q] = (in && lane + 64 * q < nrow) ? vp[lane + 64 * q] : 0.f;
    }
#pragma unroll
    for (int qq = 0; qq < 32; ++qq) vals[qq] = 0.f;
#pragma unroll
    for (int qq = 0; qq < 17; ++qq) {
      const int sidx = wid + 16 * qq;
      float acc = 0.f;
      if (sidx == 2 * c) {
#pragma unroll
        for (int q = 0; q < RPB_A / 64; ++q) acc += xs[q] * xs[q];
      } else {
#pragma unroll
        for (int q = 0; q < RPB_A / 64; ++q) acc += t[qq][q] * xs[q];
      }
      vals[qq] = acc;
    }
    const float tot = reduce32_over_wave(vals, lane);
    const int qq = reduce32_index(lane);
    const int sidx = wid + 16 * qq;
    if ((lane & 1) == 0 && qq < 17 && sidx < nsum) {
      const int row = (sidx < c) ? sidx : (sidx < 2 * c ? NB + (sidx - c) : 2 * NB);
      pa[(int64_t)row * na_ld] = (double)tot;
    }
  }
}

__global__ __launch_bounds__(256, 3) void trd_colB(TrdBatch bt, int64_t j, int c, int nsegmax, int na) {
  const TrdArgs a = bt.a[blockIdx.z];
  __shared__ float rowred[4][RS];
  __shared__ double utv_s[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n, ldv = a.ldv, lda = a.lda;
  const int64_t s = blockIdx.x;                 // strip
  const int seg = blockIdx.y;                   // segment
  const int64_t rb = (j + 1) + s * RS;          // first row of the strip
  const int64_t c_al = (j + 1) & ~(int64_t)3;
  const int64_t cend = (rb + RS < n) ? rb + RS : n;  // columns < cend belong to this strip's trapezoid
  if (seg == nsegmax) {  // spare grid row: one wave of block (0, nsegmax) computes the Householder scalars
    if (s == 0 && wid == 0) {
      const double* pn = a.partA + (int64_t)(j & 1) * PA_LD * a.na_ld + (int64_t)(2 * NB) * a.na_ld;
      double sx = 0.0;
      for (int b = lane; b < na; b += 64) sx += pn[b];
      sx = wave_sum(sx);
      if (lane == 0) {
        const double alpha = (double)a.x[j + 1];
        double beta, tau, sc;
        if (sx == 0.0) {
          beta = alpha; tau = 0.0; sc = 0.0;
        } else {
          beta = -copysign(sqrt(alpha * alpha + sx), alpha);
          tau = (beta - alpha) / beta;
          sc = 1.0 / (alpha - beta);
        }
        a.colinfo[0] = (float)tau;
        a.colinfo[1] = (float)sc;
        a.d[j] = (double)a.x[j];
        a.e[j] = beta;
        a.tau[j] = (float)tau;
      }
    }
    return;
  }
  const int bidx = (int)(s * nsegmax + seg);
  const int64_t cseg = c_al + (int64_t)seg * SEG;
  if (cseg >= cend) {                           // idle unit
    if (tid == 0) a.partB[bidx] = 0.0;
    return;
  }
  float racc[RS];
#pragma unroll
  for (int q = 0; q < RS; ++q) racc[q] = 0.f;
  double utv = 0.0;
#pragma unroll 1
  for (int ch = 0; ch < CH; ++ch) {
    const int64_t col = cseg + (int64_t)(ch * 4 + wid) * 256 + 4 * lane;  // this lane's 4 columns of chunk ch
    if (cseg + (int64_t)(ch * 4) * 256 >= cend) break;  // whole chunk row of the block is outside the trapezoid
    const bool colok = col < n && col < cend;
    // v for the 4 columns (row-dot use) and the transposed mask (columns strictly left of the strip)
    f32x4 vc = {0.f, 0.f, 0.f, 0.f};
    float tmask[4] = {0.f, 0.f, 0.f, 0.f};
    if (colok) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + col);  // x is zero-padded past n
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t idx = col + e;
        const float v = (idx > j + 1 && idx < n) ? xv[e] : 0.f;  // x~: the column without its first entry
        vc[e] = (idx < cend) ? v : 0.f;
        tmask[e] = (idx < rb && idx >= j + 1) ? 1.f : 0.f;
      }
    }
    f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < RS / PR; ++pass) {
      f32x4 av[PR];
      float vr[PR];
#pragma unroll
      for (int q = 0; q < PR; ++q) {
        const int64_t row = rb + pass * PR + q;
        const bool rok = row < n;
        av[q] = (rok && colok) ? *reinterpret_cast<const f32x4*>(a.A + row * lda + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        vr[q] = (rok && row > j + 1) ? a.x[row] : 0.f;  // wave-uniform
      }
#pragma unroll
      for (int q = 0; q < PR; ++q) {
        racc[pass * PR + q] += av[q][0] * vc[0] + av[q][1] * vc[1] + av[q][2] * vc[2] + av[q][3] * vc[3];
        cacc[0] += av[q][0] * vr[q];
        cacc[1] += av[q][1] * vr[q];
        cacc[2] += av[q][2] * vr[q];
        cacc[3] += av[q][3] * vr[q];
      }
      // __builtin_amdgcn_sched_barrier(0);  // keep the next pass's loads below this pass's FMAs: registers are reused
    }
    // transposed contribution of this (strip, chunk): complete, stored once; masked outside [j+1, rb)
#pragma unroll
    for (int e = 0; e < 4; ++e) cacc[e] *= tmask[e];
    if (col < n) *reinterpret_cast<f32x4*>(a.colpart + s * ldv + col) = cacc;
    // partial u^T v of this unit: v_col * cacc (transposed part) ...
    utv += (double)(cacc[0] * vc[0]) + (double)(cacc[1] * vc[1]) + (double)(cacc[2] * vc[2]) + (double)(cacc[3] * vc[3]);
  }
  utv = wave_sum(utv);
  // row dots: reduce the 32 per-lane values over the wave, then over the 4 waves
  const float rsum = reduce32_over_wave(racc, lane);
  if ((lane & 1) == 0) rowred[wid][reduce32_index(lane)] = rsum;
  if (lane == 0) utv_s[wid] = utv;
  __syncthreads();
  if (tid < 64) {
    double t = 0.0;
    if (tid < RS) {
      const int64_t row = rb + tid;
      const float rs = (rowred[0][tid] + rowred[1][tid]) + (rowred[2][tid] + rowred[3][tid]);
      if (row < n) {
        a.rowpart[(int64_t)seg * ldv + row] = rs;
        const float xrow = (row > j + 1) ? a.x[row] : 0.f;
        t = (double)rs * (double)xrow;  // ... + x~_row * rowdot (row part)
      }
    }
    t = wave_sum(t);
    if (tid == 0) a.partB[bidx] = t + ((utv_s[0] + utv_s[1]) + (utv_s[2] + utv_s[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// Panel end: write row-major VW = [V|W], WV = [W|V] for rows >= pe (W is already finished by trd_colA mode 1).
__global__ __launch_bounds__(256) void trd_panel_finish(TrdBatch bt, int64_t pe) {
  const TrdArgs a = bt.a[blockIdx.z];
  float* __restrict__ VW = a.VW;
  float* __restrict__ WV = a.WV;
  __shared__ float tile[2 * NB][33];
  const int tid = threadIdx.x;
  const int64_t i0 = pe + (int64_t)blockIdx.x * 32;
  for (int idx = tid; idx < 2 * NB * 32; idx += 256) {  // read 2NB x 32 (coalesced over matrix rows)
    const int row = idx >> 5, r = idx & 31;
    const int64_t i = i0 + r;
    tile[row][r] = (i < a.n) ? a.VWt[(int64_t)row * a.ldv + i] : 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < 32 * 2 * NB; idx += 256) {  // write 32 x 2NB (coalesced over panel columns)
    const int r = idx / (2 * NB), col = idx % (2 * NB);
    const int64_t i = i0 + r;
    if (i < a.n) {
      const float v = tile[col][r];
      VW[(i - pe) * (2 * NB) + col] = v;
      WV[(i - pe) * (2 * NB) + (col < NB ? col + NB : col - NB)] = v;
    }
  }
}

struct TrdJob {
  Ctx* ctx;
  TrdArgs a;
};

// workspaces + initial state of one member, enqueued on its own stream
static int sytrd_setup(Ctx* ctx, float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev, float* tau_dev, TrdJob* job) {
  const int64_t ldv = round_up(n, 64) + 512;
  const int64_t nstripMax = (n + RS - 1) / RS + 1, nsegMax = (n + SEG - 1) / SEG + 2;
  SCL_WS(ctx, VWt, float, "trd.VWt", 2 * NB * ldv);
  SCL_WS(ctx, x, float, "trd.x", ldv);
  const int64_t naMax = round_up((n + RPB_A - 1) / RPB_A + 1, 64);
  SCL_WS(ctx, partA, double, "trd.partA", 2 * naMax * PA_LD);
  SCL_WS(ctx, partB, double, "trd.partB", nstripMax * nsegMax);
  SCL_WS(ctx, colinfo, float, "trd.colinfo", CI_LD);
  SCL_WS(ctx, rowpart, float, "trd.rowpart", nsegMax * ldv);
  SCL_WS(ctx, colpart, float, "trd.colpart", nstripMax * ldv);
  SCL_WS(ctx, Gst, float, "trd.Gst", n * NB);
  SCL_WS(ctx, VW, float, "trd.VW", n * 2 * NB);
  SCL_WS(ctx, WV, float, "trd.WV", n * 2 * NB);
  SCL_HIP(ctx, hipMemsetAsync(x, 0, sizeof(float) * ldv, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(Gst, 0, sizeof(float) * n * NB, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(tau_dev, 0, sizeof(float) * n, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(e_dev, 0, sizeof(double) * n, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(colinfo, 0, sizeof(float) * CI_LD, ctx->stream));
  job->ctx = ctx;
  job->a = TrdArgs{A, n, lda, VWt, ldv, x, partA, naMax, partB, colinfo, rowpart, colpart, d_dev, e_dev, tau_dev, Gst, VW, WV};
  return SCLENS_OK;
}

// the column loop for nb members of equal order, every launch on lead->stream
static int sytrd_run(Ctx* ctx, const TrdJob* jobs, int nb) {
  TrdBatch bt{};
  for (int b = 0; b < nb; ++b) bt.a[b] = jobs[b].a;
  const int64_t n = bt.a[0].n, ldv = bt.a[0].ldv;
  const bool prof = ctx->prof_symv && nb == 1;
  int nbB_prev = 0, na_prev = 0;
  for (int64_t p = 0; p < n; p += NB) {
    const int64_t pe = (p + NB < n) ? p + NB : n;
    for (int b = 0; b < nb; ++b) SCL_HIP(ctx, hipMemsetAsync(bt.a[b].VWt, 0, sizeof(float) * 2 * NB * ldv, ctx->stream));
    for (int64_t j = p; j < pe; ++j) {
      const int c = (int)(j - p);
      const int na = (int)((n - j + RPB_A - 1) / RPB_A);
      hipLaunchKernelGGL(trd_colA, dim3(na, 1, nb), dim3(1024), 0, ctx->stream, bt, j, c, nbB_prev, na_prev, 0);
      na_prev = na;
      if (j == n - 1) break;
      const int64_t nt = n - (j + 1);
      const int nstrip = (int)((nt + RS - 1) / RS);
      const int64_t c_al = (j + 1) & ~(int64_t)3;
      const int nsegmax = (int)((n - c_al + SEG - 1) / SEG);
      if (prof) {
        if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
          for (int q = 0; q < 2; ++q) {
            hipEvent_t ev;
            SCL_HIP(ctx, hipEventCreate(&ev));
            ctx->prof_ev.push_back(ev);
          }
        }
        SCL_HIP(ctx, hipEventRecord(ctx->prof_ev[ctx->prof_used], ctx->stream));
      }
      hipLaunchKernelGGL(trd_colB, dim3(nstrip, nsegmax + 1, nb), dim3(256), 0, ctx->stream, bt, j, c, nsegmax, na);
      if (prof) {
        SCL_HIP(ctx, hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], ctx->stream));
        ctx->prof_used += 2;
        ctx->prof_bytes += 2.0 * (double)nt * (double)(nt + 1);  // unique bytes of the symmetric trailing matrix (lower half)
      }
      nbB_prev = nstrip * nsegmax;
    }
    if (pe < n) {
      const int64_t nt = n - pe;
      const int na = (int)((nt + RPB_A - 1) / RPB_A);
      hipLaunchKernelGGL(trd_colA, dim3(na, 1, nb), dim3(1024), 0, ctx->stream, bt, pe, (int)NB, nbB_prev, na_prev, 1);  // finish W[:, NB-1]
      hipLaunchKernelGGL(trd_panel_finish, dim3((unsigned)((nt + 31) / 32), 1, nb), dim3(256), 0, ctx->stream, bt, pe);
      for (int b = 0; b < nb; ++b) {
        GemmArgs g{};
        g.P = bt.a[b].VW; g.Q = bt.a[b].WV; g.C = bt.a[b].A + pe * bt.a[b].lda + pe;
        g.M = nt; g.N = nt; g.K = 2 * NB;
        g.ldp = 2 * NB; g.ldq = 2 * NB; g.ldc = bt.a[b].lda;
        g.alpha = -1.f; g.beta = 1.f; g.q_kcontig = 1; g.lower = 1; g.colabsmax = nullptr;
        SCL_TRY(gemm_f32(ctx, g));
      }
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int sytrd_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev, float* tau_dev) {
  if (n <= 0) return SCLENS_OK;
  if (lda % 4 != 0 || lda < n || (reinterpret_cast<uintptr_t>(A) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "sytrd_f32: A must be 16-byte aligned with lda a multiple of 4");
  StageTimer tm(ctx, "sytrd");
  TrdJob job;
  SCL_TRY(sytrd_setup(ctx, A, n, lda, d_dev, e_dev, tau_dev, &job));
  return sytrd_run(ctx, &job, 1);
}

// Roofline probe for bench.py: every trd_colB launch of one tridiagonalisation of order n (j = 0 .. n-2, same grids and
// arguments as sytrd_f32), back to back on ctx->stream between ONE pair of HIP events, on synthetic finite data. Reports
// the launches, the elapsed time and the algorithmic bytes (lower triangle of each trailing matrix, 2 n'(n'+1)).
int symv_probe(Ctx* ctx, int64_t n, int64_t* launches, double* total_ms, double* total_bytes) {
  if (n < 2) return ctx->fail(SCLENS_ERR_ARG, "symv_probe: n < 2");
  const int64_t lda = round_up(n, 32), ldv = round_up(n, 64) + 512;
  const int64_t nstripMax = (n + RS - 1) / RS + 1, nsegMax = (n + SEG - 1) / SEG + 2;
  SCL_WS(ctx, A, float, "probe.A", n * lda);
  SCL_WS(ctx, VWt, float, "trd.VWt", 2 * NB * ldv);
  SCL_WS(ctx, x, float, "trd.x", ldv);
  SCL_WS(ctx, partB, double, "trd.partB", nstripMax * nsegMax);
  SCL_WS(ctx, colinfo, float, "trd.colinfo", CI_LD);
  SCL_WS(ctx, rowpart, float, "trd.rowpart", nsegMax * ldv);
  SCL_WS(ctx, colpart, float, "trd.colpart", nstripMax * ldv);
  SCL_HIP(ctx, hipMemsetAsync(A, 0x3c, sizeof(float) * n * lda, ctx->stream));      // 0x3c3c3c3c = 0.0115 (finite)
  SCL_HIP(ctx, hipMemsetAsync(x, 0x3c, sizeof(float) * ldv, ctx->stream));
  SCL_HIP(ctx, hipMemsetAsync(colinfo, 0x3c, sizeof(float) * CI_LD, ctx->stream));
  const int64_t naMax = round_up((n + RPB_A - 1) / RPB_A + 1, 64);
  SCL_WS(ctx, partA, double, "trd.partA", 2 * naMax * PA_LD);
  SCL_WS(ctx, dd, double, "probe.d", n);
  SCL_WS(ctx, de, double, "probe.e", n);
  SCL_WS(ctx, dt, float, "probe.tau", n);
  SCL_HIP(ctx, hipMemsetAsync(partA, 0x3c, sizeof(double) * 2 * naMax * PA_LD, ctx->stream));  // small positive finite doubles
  TrdBatch a{};
  a.a[0] = TrdArgs{A, n, lda, VWt, ldv, x, partA, naMax, partB, colinfo, rowpart, colpart, dd, de, dt, nullptr, nullptr, nullptr};
  hipEvent_t e0, e1;
  SCL_HIP(ctx, hipEventCreate(&e0));
  SCL_HIP(ctx, hipEventCreate(&e1));
  double bytes = 0.0;
  SCL_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int64_t j = 0; j + 1 < n; ++j) {
    const int64_t nt = n - (j + 1);
    const int nstrip = (int)((nt + RS - 1) / RS);
    const int64_t c_al = (j + 1) & ~(int64_t)3;
    const int nsegmax = (int)((n - c_al + SEG - 1) / SEG);
    hipLaunchKernelGGL(trd_colB, dim3(nstrip, nsegmax + 1), dim3(256), 0, ctx->stream, a, j, (int)(j % NB), nsegmax,
                       (int)((n - j + RPB_A - 1) / RPB_A));
    bytes += 2.0 * (double)nt * (double)(nt + 1);
  }
  SCL_HIP(ctx, hipEventRecord(e1, ctx->stream));
  SCL_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  SCL_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  SCL_HIP(ctx, hipGetLastError());
  *launches = n - 1;
  *total_ms = (double)ms;
  *total_bytes = bytes;
  return SCLENS_OK;
}

}  // namespace scl
