// Two-stage symmetric eigensolver for gfx950: dense -> band of half-width SB = 64 on the matrix cores (sy2sb_f32), band ->
// tridiagonal by bulge chasing in one persistent kernel (sb2st_f32), and the two back-transformations (sbr_apply_q2,
// sbr_apply_q1). Selected by order (n >= 16 000) or by the context option "two_stage"; the one-stage reduction of tridiag.hip
// streams the trailing matrix once per column (2/3 n^3 bytes) and is the faster one below that.
//
// Replaces, together with the tridiagonal solver of trieig.hip, `_get_eigen` (scLENS.jl:375-387 -> cuSOLVER ssyevd / LAPACK
// dsyevr in the reference).
//
// Stage 1. Panel p reduces the columns [p SB, (p+1) SB) below the band: P = A[r0:, c0:c0+SB] (n' x SB, r0 = c0 + SB) is
// factored P = Q R by a Cholesky QR whose Gram matrix and triangular algebra are fp64 (fp32 data: orthogonality ~ eps64
// cond(P)^2 + eps32, so one pass is enough up to cond ~ 1e4; a non-positive pivot raises the breakdown flag and the caller
// falls back to the one-stage solver), the Householder representation Q D = (I - V T V')[:, :SB] is reconstructed from the
// thin Q by the sign-modified LU of its top block (Ballard, Demmel, Grigori, Jacquelin, Knight, Nguyen: "Reconstructing
// Householder vectors from tall-skinny QR", 2015): only SB x SB work is sequential. The trailing matrix then gets the
// two-sided update A22 <- A22 - V Z' - Z V' with W = A22 V, Y = W T, Z = Y - 1/2 V (T' V' Y): one skinny MFMA product
// (256 x 64 tiles, split over K inside one launch) and one rank-128 symmetric MFMA update (256 x 256 tiles) per panel; the
// tall-skinny algebra in between runs on 16x16x4 MFMAs (f64 where the Cholesky QR needs it). The next panel is factored on
// a second stream while the bulk of the update runs (look-ahead).
// Storage: V_p is kept in the UPPER part of A (rows c0..c0+SB-1, columns r0..n-1: contiguous over the long dimension, the
// layout every NT product here wants), the band in the LOWER part, T_p in a side array. The matrix order must be a multiple
// of SB (the caller pads with a decoupled diagonal block). All reductions run in a fixed order: bitwise reproducible.
#include <algorithm>

#include "sbr_common.h"

namespace scl {

__global__ void sbr_row_abs_max(const float* __restrict__ A, int64_t n, int64_t lda, unsigned* __restrict__ out);

// Split-K factor of a product whose `tiles` output tiles (one workgroup each, one workgroup per CU) do not fill the 256 CUs:
// the grid runs in ceil(tiles S / 256) rounds of K / S each, so the time goes like rounds(S) / S. (59 row tiles with S = 9 are
// 531 workgroups = 3 rounds of K / 9; S = 13 gives 767 = 3 rounds of K / 13: the same product in 0.69 of the time.)
static inline int sbr_pick_splits(int64_t tiles, int max_s, int64_t K, int cus = 256) {
  int best = 1;
  double best_cost = 1e300;
  for (int S = 1; S <= max_s; ++S) {
    if (S > 1 && K / S < 256) break;  // keep the slices long enough for the staging pipeline
    const double rounds = (double)((tiles * S + cus - 1) / cus);
    const double cost = rounds / S + 0.004 * S;  // the slabs are summed by the consumer: a slight preference for fewer
    if (cost < best_cost) {
      best_cost = cost;
      best = S;
    }
  }
  return best;
}

// sum of the partials in a fixed order: out[idx] = sum_p part[p][idx], idx < SB * SB (16 workgroups). Sixteen loads in flight per
// thread, four interleaved partial sums (p mod 4) combined as (s0 + s1) + (s2 + s3): the ~100 partials of a long panel cost a few
// memory round trips instead of one each (a serial loop took 44 us per call in round 2, two calls per panel).
__global__ __launch_bounds__(256) void sbr_sum_parts(const double* __restrict__ part, int nparts, double* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int p = 0;
  for (; p + 15 < nparts; p += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(int64_t)(p + u) * SB * SB + idx];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u & 3] += v[u];
  }
  for (; p < nparts; ++p) s[p & 3] += part[(int64_t)p * SB * SB + idx];
  out[idx] = (s[0] + s[1]) + (s[2] + s[3]);
}

// fp64 reciprocal / reciprocal square root from the fp32 hardware estimate + Newton steps: the pivots of the 64-step
// eliminations below sit on the critical path of every step, and the IEEE division / sqrt sequences are ~10x longer.
// Three steps from a 23-bit estimate reach full fp64 accuracy (the error squares, resp. cubes, per step) for normal arguments.
__device__ __forceinline__ double sbr_rcp64(double d) {
  double x = (double)__frcp_rn((float)d);
#pragma unroll
  for (int it = 0; it < 3; ++it) x = x + x * (1.0 - d * x);
  return x;
}
__device__ __forceinline__ double sbr_rsqrt64(double d) {
  double x = (double)rsqrtf((float)d);
#pragma unroll
  for (int it = 0; it < 3; ++it) x = x * (1.5 - 0.5 * d * x * x);
  return x;
}

// ---- the SB x SB algebra of one panel, one wave, fp64 in LDS -----------------------------------------------------------
// in : part[nparts][SB][SB] (Gram partials of the panel), Ptop = transposed top block of the panel
//      (Ptop[j * ldp + i] = P[i][j], i, j < SB)
// out: Mout = R^-1 D U'^-1 (V2 = P2 * Mout), V1 (unit lower, row-major fp32), T (upper, row-major fp32),
//      Rh = D R (upper, fp32), flag != 0 on breakdown
struct SbrSmall {
  double* M;     // [SB][SB]
  float* V1;     // [SB][SB]
  float* T;      // [SB][SB]
  float* Rh;     // [SB][SB]
  int* flag;
  unsigned long long* prof;  // context option panel_prof = 1: [10] shader clocks per phase of sbr_panel_small + the call count (else nullptr)
};

// 64 x 64 fp64 matrices in LDS, one workgroup of 256 threads. Thread (ti, tj) = (tid >> 4, tid & 15) owns the entries
// (ti + 16 u, tj + 16 v), u, v < 4: sixteen consecutive columns per 16-lane group (two-way bank conflicts at most).
// C = A * B, optionally with A given in fp32 (AF), B treated as upper triangular (entries below the diagonal are not read as
// zero but skipped: the storage there may hold something else), the result negated, and written to LDS (C) or global (G32 / G64)
template <typename TA, bool B_UPPER>
__device__ __forceinline__ void mm64_acc(double (&acc)[4][4], const TA* A, const double* B) {
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k = 0; k < SB; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = (double)A[(ti + 16 * u) * SB + k];
#pragma unroll
    for (int v = 0; v < 4; ++v) b[v] = (!B_UPPER || k <= tj + 16 * v) ? B[k * SB + tj + 16 * v] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
  }
}

// X = U^-1 for an upper triangular U (element (j, k), k >= j, through `uel`; `inv_diag[j]` = 1 / U[j][j]), computed by ONE wave:
// lane c owns column c of X in registers (back substitution from the diagonal upwards, x_j = (delta_jc - sum_{k > j} U_jk x_k) /
// U_jj; entries below the diagonal are zero, so every lane runs the same fully unrolled recurrence) and no barrier is needed.
// The other waves of the workgroup skip this and meet the caller's barrier.
template <class UEL>
__device__ __forceinline__ void trinv_cols(double* X, UEL uel, const double* inv_diag) {
  const int c = threadIdx.x & 63;
  double x[SB];
#pragma unroll
  for (int j = SB - 1; j >= 0; --j) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = j + 1; k < SB; k += 2) {
      s0 += uel(j, k) * x[k];
      if (k + 1 < SB) s1 += uel(j, k + 1) * x[k + 1];
    }
    const double v = (((j == c) ? 1.0 : 0.0) - (s0 + s1)) * inv_diag[j];
    x[j] = (j <= c) ? v : 0.0;
  }
#pragma unroll
  for (int j = 0; j < SB; ++j) X[j * SB + c] = x[j];
}

// The same inverse by ALL 256 threads, blocked 16 x 16 (round 4). The column recurrence above is 2 016 dependent {broadcast LDS read,
// fp64 fma} pairs on ONE wave: 76 000 clocks per inverse, three inverses per panel = 41 % of sbr_panel_small
// (profiles/r04_panel_small_phase_clocks.log) while three waves wait at the barrier. Here: (1) the four diagonal blocks by the same
// recurrence on 16 columns each, one block per wave, in parallel (120 pairs per lane); (2) the blocks at distance d = 1, 2, 3 above
// the diagonal from X_ij = -X_ii (sum_{i < k <= j} U_ik X_kj): every thread one entry of every block of that distance, the inner
// sums through a 6 KB scratch, two barriers per distance. Entries below the diagonal are written as zeros (the products that consume
// X read its first operand in full). Contains barriers: every thread of the workgroup must call it.
template <class UEL>
__device__ __forceinline__ void trinv_blocked(double* X, UEL uel, const double* inv_diag, double* S) {
  constexpr int NB = 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int idx = tid; idx < SB * SB; idx += 256) X[idx] = 0.0;
  __syncthreads();
  if (lane < NB) {  // diagonal block `wave`, column `lane`
    const int o = NB * wave, c = lane;
    double x[NB];
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = j + 1; k < NB; k += 2) {
        s0 += uel(o + j, o + k) * x[k];
        if (k + 1 < NB) s1 += uel(o + j, o + k + 1) * x[k + 1];
      }
      const double v = (((j == c) ? 1.0 : 0.0) - (s0 + s1)) * inv_diag[o + j];
      x[j] = (j <= c) ? v : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) X[(o + j) * SB + o + c] = x[j];
  }
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
#pragma unroll
  for (int d = 1; d < SB / NB; ++d) {
    // inner sums S_b = sum_{kb = br + 1 .. bc} U[br][kb] X[kb][bc] of the blocks (br, bc = br + d), br = 0 .. 3 - d
#pragma unroll
    for (int br = 0; br + d < SB / NB; ++br) {
      const int bc = br + d;
      double acc = 0.0;
      for (int kb = br + 1; kb <= bc; ++kb)
#pragma unroll
        for (int t = 0; t < NB; ++t) acc += uel(NB * br + r, NB * kb + t) * X[(NB * kb + t) * SB + NB * bc + c];
      S[br * NB * NB + r * NB + c] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int br = 0; br + d < SB / NB; ++br) {
      const int bc = br + d;
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < NB; ++t) acc += X[(NB * br + r) * SB + NB * br + t] * S[br * NB * NB + t * NB + c];
      X[(NB * br + r) * SB + NB * bc + c] = -acc;
    }
    __syncthreads();
  }
}

constexpr int SBR_PANEL_LDS = 4 * SB * SB * (int)sizeof(double) + SB * SB * (int)sizeof(float) + 4 * SB * (int)sizeof(double) +
                              3 * 16 * 16 * (int)sizeof(double) + 4 * SB * (int)sizeof(double);

// The SB x SB algebra of one panel. Every 64-step elimination below runs with ONE barrier per step: the pivot row / column of
// a step is only read during the step and the entries it updates are disjoint from it (scaled rows / columns go to a second
// matrix instead of in place); the triangular inverses are barrier-free (one wave, a column per lane).
// First build (round 2): 3 + 2 barriers per step, inverses row by row with a reduction per row: 357 us per panel, which was the
// critical path of every panel below n' ~ 15 000 (the look-ahead hides it only while the trailing update takes longer).
__global__ __launch_bounds__(256) void sbr_panel_small(const double* __restrict__ G, const float* __restrict__ Ptop, int64_t ldp,
                                                       SbrSmall o) {
  extern __shared__ double lds[];
  double* M0 = lds;                // G (Cholesky work) -> L of the LU      -> (V1')^-1 ... see below
  double* M1 = lds + SB * SB;      // R                                      -> U'
  double* M2 = lds + 2 * SB * SB;  // R^-1                                   -> R^-1 D
  double* M3 = lds + 3 * SB * SB;  // Q_top (LU work; U' raw in the upper part) -> inverse factors
  float* F0 = reinterpret_cast<float*>(lds + 4 * SB * SB);  // P_top [i][j]
  double* invd = reinterpret_cast<double*>(F0 + SB * SB);   // [SB] reciprocal diagonals
  double* dsign = invd + SB;                                 // [SB]
  double* pivs = dsign + SB;                                 // [SB]
  double* unit_diag = pivs + SB;                             // [SB] ones
  double* trS = unit_diag + SB;                              // [3][16][16] scratch of the blocked inverses
  double* rowb = trS + 3 * 16 * 16;                          // [2][SB] pivot row of an elimination step (double-buffered)
  double* colb = rowb + 2 * SB;                              // [2][SB] pivot column (LU)
  __shared__ int bad;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  unsigned long long pt = o.prof ? __builtin_amdgcn_s_memtime() : 0ull;
#define SBR_PP(i)                                                             \
  if (o.prof && tid == 0) {                                                   \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
    atomicAdd(o.prof + (i), now_ - pt);                                       \
    pt = now_;                                                                \
  }
  if (tid == 0) bad = 0;
  for (int idx = tid; idx < SB * SB; idx += 256) {
    M0[idx] = G[idx];
    const int j = idx >> 6, i = idx & 63;  // coalesced over i
    F0[i * SB + j] = Ptop[(int64_t)j * ldp + i];
  }
  __syncthreads();
  SBR_PP(0)  // loads
  // ---- Cholesky G = R'R, right-looking on the full symmetric matrix. Round 4: the matrix lives in REGISTERS (thread (ti, tj) owns
  // the entries (ti + 16 u, tj + 16 v)); a step publishes its pivot row through a double-buffered 64-entry LDS line (one barrier per
  // step) instead of reading and rewriting the 32 KB matrix in LDS -- 64 KB of LDS traffic per step at 128 B / clock were the 1 760
  // clocks of a step (profiles/r04_panel_small_phase_clocks.log). Same operations on the same operands: the same bits.
  {
    double a[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) a[u][v] = M0[(ti + 16 * u) * SB + tj + 16 * v];
    if (ti == 0) {  // row 0
#pragma unroll
      for (int v = 0; v < 4; ++v) rowb[tj + 16 * v] = a[0][v];
    }
    __syncthreads();
    for (int j = 0; j < SB; ++j) {
      const double* row = rowb + (j & 1) * SB;
      double d = row[j];
      if (!(d > 0.0)) {
        if (tid == 0) bad = 1;
        d = 1.0;
      }
      const double rs = sbr_rsqrt64(d), rd = rs * rs;
      if (tid < SB) {
        M1[j * SB + tid] = (tid >= j) ? ((tid == j) ? d * rs : row[tid] * rs) : 0.0;
        if (tid == j) invd[j] = rs;  // 1 / R[j][j]
      }
      double rr[4], rc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) rr[u] = row[ti + 16 * u] * rd;
#pragma unroll
      for (int v = 0; v < 4; ++v) rc[v] = row[tj + 16 * v];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ti + 16 * u;
        if (i > j) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int c = tj + 16 * v;
            if (c > j) a[u][v] -= rr[u] * rc[v];
          }
        }
      }
      // the owners of row j + 1 publish it for the next step (the other line: this step's readers may still be at work)
      if (j + 1 < SB && ti == ((j + 1) & 15)) {
        double* nxt = rowb + ((j + 1) & 1) * SB;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u == ((j + 1) >> 4)) {
#pragma unroll
            for (int v = 0; v < 4; ++v) nxt[tj + 16 * v] = a[u][v];
          }
      }
      __syncthreads();
    }
  }
  SBR_PP(1)  // Cholesky
  // ---- R^-1
  trinv_blocked(M2, [&](int j, int k) { return M1[j * SB + k]; }, invd, trS);
  SBR_PP(2)  // R^-1
  // ---- Q_top = P_top R^-1, kept in registers for the elimination below (thread (ti, tj): entries (ti + 16 u, tj + 16 v))
  double q[4][4];
  mm64_acc<float, true>(q, F0, M2);
  if (ti == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) rowb[tj + 16 * v] = q[0][v];
  }
  if (tj == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) colb[ti + 16 * u] = q[u][0];
  }
  __syncthreads();
  SBR_PP(3)  // Q_top = P_top R^-1
  // ---- sign-modified LU of (Q D - E), right-looking: D_j = -sgn(q_jj), pivot = D_j q_jj - 1 = -|q_jj| - 1,
  //      L[i][j] = D_j q_ij / pivot (to M0); the rows of U' are written back to the upper part of M3 after the loop. As in the
  //      Cholesky loop the matrix stays in registers and a step publishes its pivot row and column through LDS lines.
  for (int j = 0; j < SB; ++j) {
    const double* row = rowb + (j & 1) * SB;
    const double* col = colb + (j & 1) * SB;
    const double qjj = row[j];
    const double dj = (qjj >= 0.0) ? -1.0 : 1.0;
    const double piv = dj * qjj - 1.0;
    const double rp = dj * sbr_rcp64(piv);
    if (tid == 0) {
      dsign[j] = dj;
      pivs[j] = piv;
    }
    double lc[4], ur[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) lc[u] = col[ti + 16 * u] * rp;
#pragma unroll
    for (int v = 0; v < 4; ++v) ur[v] = row[tj + 16 * v];
    if (tj == (j & 15)) {  // the threads whose column set contains j keep L's column j
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ti + 16 * u;
        M0[i * SB + j] = (i > j) ? lc[u] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = ti + 16 * u;
      if (i > j) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = tj + 16 * v;
          if (c > j) q[u][v] -= lc[u] * ur[v];
        }
      }
    }
    if (j + 1 < SB) {  // pivot row / column of the next step, into the other pair of lines
      const int jn = j + 1;
      if (ti == (jn & 15)) {
        double* nxt = rowb + (jn & 1) * SB;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u == (jn >> 4)) {
#pragma unroll
            for (int v = 0; v < 4; ++v) nxt[tj + 16 * v] = q[u][v];
          }
      }
      if (tj == (jn & 15)) {
        double* nxt = colb + (jn & 1) * SB;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (v == (jn >> 4)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[ti + 16 * u] = q[u][v];
          }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) M3[(ti + 16 * u) * SB + tj + 16 * v] = q[u][v];
  __syncthreads();
  SBR_PP(4)  // LU
  // ---- outputs that need R and the raw LU: Rh = D R, V1 (unit lower); then U' (with the column signs) replaces R in M1
  for (int idx = tid; idx < SB * SB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    o.Rh[idx] = (c >= r) ? (float)(dsign[r] * M1[idx]) : 0.f;
    o.V1[idx] = (c < r) ? (float)M0[idx] : (c == r ? 1.f : 0.f);
  }
  __syncthreads();
  for (int idx = tid; idx < SB * SB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    M1[idx] = (c > r) ? dsign[c] * M3[idx] : (c == r ? pivs[r] : 0.0);
  }
  if (tid < SB) {
    invd[tid] = 1.0 / pivs[tid];
    unit_diag[tid] = 1.0;
  }
  __syncthreads();
  SBR_PP(5)  // Rh, V1, U'
  // ---- T = -U' (V1')^-1: V1' is unit upper triangular with element (j, k) = L[k][j]
  trinv_blocked(M3, [&](int j, int k) { return M0[k * SB + j]; }, unit_diag, trS);
  SBR_PP(6)  // (V1')^-1
  {
    double acc[4][4];
    mm64_acc<double, true>(acc, M1, M3);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = ti + 16 * u, c = tj + 16 * v;
        o.T[r * SB + c] = (c >= r) ? (float)(-acc[u][v]) : 0.f;
      }
  }
  __syncthreads();
  SBR_PP(7)  // T
  // ---- M = (R^-1 D) U'^-1
  trinv_blocked(M3, [&](int j, int k) { return M1[j * SB + k]; }, invd, trS);
  for (int idx = tid; idx < SB * SB; idx += 256) M2[idx] *= dsign[idx & 63];  // scale the columns of R^-1
  __syncthreads();
  SBR_PP(8)  // U'^-1
  {
    double acc[4][4];
    mm64_acc<double, true>(acc, M2, M3);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) o.M[(ti + 16 * u) * SB + tj + 16 * v] = acc[u][v];
  }
  SBR_PP(9)  // M
  if (o.prof && tid == 0) atomicAdd(o.prof + 10, 1ull);
#undef SBR_PP
  if (tid == 0 && bad) atomicExch(o.flag, 1);
}

// ---- the last panel (n' = SB rows, some of which may be the zero rows of the padding: rank deficient, so no Cholesky QR):
// plain Householder QR of the SB x SB block in LDS, one wave, fp64. Same outputs as sbr_panel_small (M is not needed).
__global__ __launch_bounds__(64) void sbr_panel_house(const float* __restrict__ Ptop, int64_t ldp, SbrSmall o) {
  __shared__ double P[SB][SB + 1], T[SB][SB + 1];
  __shared__ double tau_s[SB];
  const int l = threadIdx.x;  // lane = row
  for (int c = 0; c < SB; ++c) {
    P[l][c] = (double)Ptop[(int64_t)c * ldp + l];
    T[l][c] = 0.0;
  }
  __syncthreads();
  for (int j = 0; j < SB; ++j) {
    // reflector from P[j:, j]
    double xi = (l > j) ? P[l][j] : 0.0;
    double sg = xi * xi;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) sg += __shfl_xor(sg, o2);
    const double alpha = P[j][j];
    double tau = 0.0, beta = alpha, scale = 0.0;
    if (sg > 0.0) {
      const double nrm = sqrt(alpha * alpha + sg);
      beta = (alpha >= 0.0) ? -nrm : nrm;
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    const double vl = (l == j) ? 1.0 : (l > j ? xi * scale : 0.0);
    __syncthreads();
    // apply H_j = I - tau v v' to the columns c > j; column j becomes (.., beta, 0, ..)
    for (int c = j + 1; c < SB; ++c) {
      double t = vl * P[l][c];
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2);
      P[l][c] -= tau * vl * t;
    }
    if (l == j) P[j][j] = beta;
    if (l > j) P[l][j] = vl;  // V below the diagonal
    if (l == 0) tau_s[j] = tau;
    __syncthreads();
  }
  // T (forward, columnwise): T[j][j] = tau_j, T[0:j, j] = -tau_j T[0:j, 0:j] (V[:, 0:j]' v_j)
  for (int j = 0; j < SB; ++j) {
    const double tj = tau_s[j];
    // g_i = V[:, i]' v_j for i < j  (lane i): v_j = (0.., 1 at j, P[r][j] for r > j), V[r][i] = (r == i ? 1 : r > i ? P[r][i] : 0)
    double g = 0.0;
    if (l < j) {
      g = P[j][l];  // r = j: V[j][l] * 1   (j > l)
      for (int r = j + 1; r < SB; ++r) g += P[r][l] * P[r][j];
    }
    __syncthreads();
    if (l < j) T[l][SB] = g;  // scratch column
    __syncthreads();
    if (l < j) {
      double acc = 0.0;
      for (int i = l; i < j; ++i) acc += T[l][i] * T[i][SB];
      T[l][j] = -tj * acc;
    }
    if (l == j) T[j][j] = tj;
    __syncthreads();
  }
  for (int c = 0; c < SB; ++c) {
    o.Rh[l * SB + c] = (c >= l) ? (float)P[l][c] : 0.f;
    o.V1[l * SB + c] = (c < l) ? (float)P[l][c] : (c == l ? 1.f : 0.f);
    o.T[l * SB + c] = (c >= l) ? (float)T[l][c] : 0.f;
    o.M[l * SB + c] = 0.0;
  }
}

// Sh = 1/2 T' (V'Y), V'Y = the summed cross partials (sbr_sum_parts); 256 threads, 4 x 4 outputs each
__global__ __launch_bounds__(256) void sbr_small_s(const double* __restrict__ VtY, const float* __restrict__ T, double* __restrict__ Sh,
                                                   unsigned* __restrict__ zmax) {
  __shared__ double G[SB * SB], Tt[SB * SB];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  if (zmax && tid == 0) *zmax = 0u;  // the next kernel (sbr_rmul_f32<3>) leaves the largest |Z| of this panel here
  for (int idx = tid; idx < SB * SB; idx += 256) {
    G[idx] = VtY[idx];
    Tt[(idx & 63) * SB + (idx >> 6)] = (double)T[idx];  // T' (T is upper: its strictly lower part is stored as 0)
  }
  __syncthreads();
  double acc[4][4];
  mm64_acc<double, false>(acc, Tt, G);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) Sh[(ti + 16 * u) * SB + tj + 16 * v] = 0.5 * acc[u][v];
}

// ---- tall-skinny algebra of a panel on the matrix cores ------------------------------------------------------------------
// All "n' x 64 times 64 x 64" and "64 x n' times n' x 64" products of a panel run on `v_mfma_*_16x16x4` (f64 where the
// Cholesky QR needs it, f32 otherwise), one wave per tile of 16 positions of the long dimension, the 64 x 64 factor held in
// registers as MFMA operands. Layouts: Pt = the panel in the transposed storage of A ([64][lda], long dimension contiguous:
// what the NT GEMMs want as their 64-row operand), everything else row-major [n'][64] (what an MFMA operand with the long
// dimension on the lanes of a quad reads in 64-byte runs).
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SBR_RT = 16;          // positions per wave tile
constexpr int SBR_GCH = 256;        // positions per workgroup of the Gram kernels (four waves of 64)
constexpr int SBR_GRAM_LDS = 2 * SB * SB * (int)sizeof(double);  // dynamic LDS of sbr_gram64 (64 KB: at the default limit)

// part[wg][i][j] = sum over the workgroup's positions r of X[r][i] Y[r][j] (fp64), X / Y given as
//   TRANSPOSED = true : Xt[i * ldx + r]   (the panel in A; X == Y: its Gram matrix)
//   TRANSPOSED = false: X[r * 64 + i], Y[r * 64 + j]
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void sbr_gram64(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y, int64_t len,
                                                  double* __restrict__ part) {
  extern __shared__ double red[];  // 2 x [SB * SB]: waves 2, 3 -> waves 0, 1, then wave 1 -> wave 0 (two barriers, fixed order)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t r_begin = (int64_t)blockIdx.x * SBR_GCH + wv * (SBR_GCH / 4);
  int64_t r_end = r_begin + SBR_GCH / 4;
  if (r_end > len) r_end = len;
  for (int64_t rb = r_begin; rb < r_end; rb += 16) {
    // k-step e of this 16-position chunk: lane quad kg supplies position rb + 4 kg + e
    float xa[4][4], yb[4][4];  // [tile][e]
    if (TRANSPOSED) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* px = X + (int64_t)(16 * t + l15) * ldx + rb + 4 * kg;
        if (rb + 4 * kg + 3 < r_end) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(px);
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[t][e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[t][e] = (rb + 4 * kg + e < r_end) ? px[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) yb[t][e] = xa[t][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t r = rb + 4 * kg + e;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xa[t][e] = (r < r_end) ? X[r * SB + 16 * t + l15] : 0.f;
          yb[t][e] = (r < r_end) ? Y[r * SB + 16 * t + l15] : 0.f;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xa[a][e], (double)yb[b][e], acc[a][b], 0, 0, 0);
  }
  // D layout (f64): column = lane & 15, row = (lane >> 4) + 4 * reg. Sum of the four waves as (w0 + w2) + (w1 + w3):
  double* out = part + (int64_t)blockIdx.x * SB * SB;
  if (wv >= 2) {
    double* dst = red + (wv - 2) * SB * SB;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[((a * 4 + b) * 4 + g) * 64 + lane] = acc[a][b][g];
  }
  __syncthreads();
  if (wv < 2) {
    const double* src = red + wv * SB * SB;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[a][b][g] += src[((a * 4 + b) * 4 + g) * 64 + lane];
  }
  __syncthreads();
  if (wv == 1) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((a * 4 + b) * 4 + g) * 64 + lane] = acc[a][b][g];
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[a][b][g] += red[((a * 4 + b) * 4 + g) * 64 + lane];
  }
  if (wv == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) out[(16 * a + kg + 4 * g) * SB + 16 * b + l15] = acc[a][b][g];
  }
}

// V2 = P2 * Mat (fp64 factor, fp64 accumulation): rows r >= SB of the panel. In place in the transposed storage
// (Pt[j][r] <- sum_i Mat[i][j] Pt[i][r]) and as a row-major copy Vr[r][j]. One wave per 16 positions.
__global__ __launch_bounds__(256) void sbr_vmul_f64(float* __restrict__ Pt, int64_t lda, const double* __restrict__ Mat, int64_t len,
                                                    float* __restrict__ Vr) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = SB + wave * SBR_RT;  // the top block (r < SB) is written by sbr_top_block
  if (r0 >= len) return;
  // A operand: A[j][k = i] = Mat[i][j], lane (j = l15 within tile jt, k = kg within step ks)
  double am[16][4];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) am[ks][jt] = Mat[(4 * ks + kg) * SB + 16 * jt + l15];
  const int64_t r = r0 + l15;
  double bp[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bp[ks] = (r < len) ? (double)Pt[(int64_t)(4 * ks + kg) * lda + r] : 0.0;
  f64x4 acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) acc[jt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[ks][jt], bp[ks], acc[jt], 0, 0, 0);
  // D[j = 16 jt + kg + 4 g][r = r0 + l15]
  if (r < len) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 16 * jt + kg + 4 * g;
        const float v = (float)acc[jt][g];
        Pt[(int64_t)j * lda + r] = v;
        Vr[r * SB + j] = v;
      }
  }
}

// top block of a panel: V1 (unit lower) into the transposed storage and the row-major copy, Rh into the band block of A
__global__ __launch_bounds__(256) void sbr_top_block(float* __restrict__ Pt, int64_t lda, const float* __restrict__ V1,
                                                     const float* __restrict__ Rh, float* __restrict__ band, float* __restrict__ Vr) {
  for (int idx = threadIdx.x; idx < SB * SB; idx += 256) {
    const int r = idx >> 6, j = idx & 63;
    const float v = V1[r * SB + j];
    Vr[r * SB + j] = v;
    band[(int64_t)r * lda + j] = Rh[r * SB + j];
  }
  for (int idx = threadIdx.x; idx < SB * SB; idx += 256) {
    const int j = idx >> 6, r = idx & 63;
    Pt[(int64_t)j * lda + r] = V1[r * SB + j];
  }
}

// out[r][j] = sum_i in[r][i] F[i][j] (f32 MFMA), in = the sum of `nslab` row-major slabs (fixed order).
//   MODE 2: Yr = W T                      (in = split-K slabs of W)
//   MODE 3: Z = Yr - Vr Sh  and the row-major operands of the rank-128 update: VW[r] = [V | Z], WV[r] = -[Z | V] (row pitch ldo)
template <int MODE>
__global__ __launch_bounds__(256) void sbr_rmul_f32(const float* __restrict__ in, int nslab, int64_t slab, const float* __restrict__ F32,
                                                    const double* __restrict__ F64, int64_t len, float* __restrict__ out,
                                                    const float* __restrict__ Yr, float* __restrict__ VW, float* __restrict__ WV,
                                                    int64_t ldo, unsigned* __restrict__ zmax = nullptr) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = wave * SBR_RT;
  if (r0 >= len) return;
  // k-step (q, e): lane quad kg carries i = 16 q + 4 kg + e (so that a lane's four e are one 16-byte load of `in`)
  float bf[4][4][4];  // [q][e][jt]: B[k = i][j = 16 jt + l15]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int i = 16 * q + 4 * kg + e;
        bf[q][e][jt] = (MODE == 2) ? F32[i * SB + 16 * jt + l15] : (float)F64[i * SB + 16 * jt + l15];
      }
  const int64_t r = r0 + l15;
  f32x4 a4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
    if (r < len)
      for (int sl = 0; sl < nslab; ++sl) sacc += *reinterpret_cast<const f32x4*>(in + (int64_t)sl * slab + r * SB + 16 * q + 4 * kg);
    a4[q] = sacc;
  }
  f32x4 acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q][e], bf[q][e][jt], acc[jt], 0, 0, 0);
  // D[r = r0 + 4 kg + g][j = 16 jt + l15]
  float zm = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int64_t rr = r0 + 4 * kg + g;
    if (rr >= len) continue;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int j = 16 * jt + l15;
      if (MODE == 2) {
        out[rr * SB + j] = acc[jt][g];
      } else {
        const float z = Yr[rr * SB + j] - acc[jt][g];
        const float v = in[rr * SB + j];
        zm = fmaxf(zm, fabsf(z));
        VW[rr * ldo + j] = v;
        VW[rr * ldo + SB + j] = z;
        WV[rr * ldo + j] = -z;  // negated: the update is then C += [V | Z] [-Z | -V]' with the accumulators started from C
        WV[rr * ldo + SB + j] = -v;
      }
    }
  }
  if (MODE == 3 && zmax) {  // largest |Z| of the panel for the scale of the split update's operands (one atomic per wave; a maximum
                            // does not depend on the order of its arguments)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zm = fmaxf(zm, __shfl_xor(zm, o));
    if (lane == 0) atomicMax(zmax, __float_as_uint(zm));
  }
}

// out[idx] = sum_s in[s][idx] in a fixed order (split-K slabs of a small product), idx < count
__global__ __launch_bounds__(256) void sbr_sum_slabs(const float* __restrict__ in, int nslab, int64_t slab, int count, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float acc = in[i];
  for (int q = 1; q < nslab; ++q) acc += in[(int64_t)q * slab + i];
  out[i] = acc;
}

int sbr_ensure_aux(Ctx* ctx) {  // the context's second stream and its events, created on first use
  if (!ctx->aux_stream) {
    SCL_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    SCL_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_ev[0], hipEventDisableTiming));
    SCL_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_ev[1], hipEventDisableTiming));
  }
  if (!ctx->q2_ev) SCL_HIP(ctx, hipEventCreateWithFlags(&ctx->q2_ev, hipEventDisableTiming));
  return SCLENS_OK;
}

// upper half of the leading 64 x 64 block = its lower half (exact symmetry after a full, unsymmetrically rounded update of the block)
__global__ __launch_bounds__(256) void sbr_mirror_diag(float* __restrict__ A, int64_t lda) {
  for (int idx = threadIdx.x; idx < SB * SB; idx += 256) {
    const int i = idx / SB, j = idx % SB;
    if (j > i) A[i * lda + j] = A[j * lda + i];
  }
}

// ---- W = A22 V on the fp16 matrix cores (round 4) ---------------------------------------------------------------------
// The skinny product streams the whole trailing matrix once per panel (4 n'^2 bytes) for 2 * 64 n'^2 flop: on the fp32 matrix pipe
// (256 x 64 tiles of gemm_nt_big) it is bound by that pipe from the second octile on (the trailing matrix then comes out of the
// 256 MB MALL faster than the pipe consumes it). Here the fp32 tile of A22 arrives in LDS by DMA exactly as before, each lane
// splits the eight k-consecutive values of its fragment into fp16 pieces in registers (x = hi + lo of the value scaled by a power
// of two: 2 conversions + 1 mixed fma per element), V comes as a finished image of [hi | lo] fp16 units built once per panel
// (sbr_v_image), and the product is hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation: 24 matrix
// instructions of 16 clocks per wave and 32 of K instead of 32 of 64 clocks -- the kernel is bound by the delivery of A22 alone.
// Scales: every entry of every trailing matrix is bounded by the 2-norm of A, hence by its largest absolute row sum (computed
// once per reduction); A22 is scaled so that this bound sits below 2^15, V (entries at most 1 in magnitude) by 2^13.
struct SbrWArgs {
  const float* A22;
  int64_t lda;
  const float* Vimg;  // per 32 rows of V: 2048 floats = hi units [g 4][col 64][8 halves], then the lo units
  float* Wp;          // slab s = W partial of K-slice s, [n'][64]
  int64_t np, kch, slab;
  const float* sc;    // {scale of A22, 1 / (scale of A22 * 2^13)}
};
constexpr int SBR_W_STAGE = 256 * 32 + 2048;  // floats of one LDS stage: the A22 tile (32 KB) + the V image of the step (8 KB)
constexpr float SBR_W_VSCALE = 8192.f;

__global__ void sbr_w_scale(const unsigned* __restrict__ bound, float mul, float add, float* __restrict__ sc) {
  const float limit = mul * __uint_as_float(*bound) + add;
  float s = 1.f;
  if (limit > 0.f && limit < 3.0e38f) {  // limit * s < 2^15; the exponent clamped so that s, 1 / s and 1 / (2^13 s) stay normal
    int e = 15 - (ilogbf(limit) + 1);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    s = ldexpf(1.f, e);
  }
  sc[0] = s;
  sc[1] = 1.f / (s * SBR_W_VSCALE);
}

// V image of one panel: block kt covers the rows 32 kt .. 32 kt + 31 of V (row-major [n'][64])
__global__ __launch_bounds__(256) void sbr_v_image(const float* __restrict__ Vr, float* __restrict__ img) {
  const int u = threadIdx.x, g = u >> 6, col = u & 63;
  const float* src = Vr + ((int64_t)blockIdx.x * 32 + 8 * g) * SB + col;
  f32x4 x0, x1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x0[e] = src[e * SB] * SBR_W_VSCALE;
    x1[e] = src[(4 + e) * SB] * SBR_W_VSCALE;
  }
  const SbrHL8 o = sbr_cat(sbr_split_pk(x0), sbr_split_pk(x1));
  f32x4 rh, rl;
  __builtin_memcpy(&rh, &o.h, 16);
  __builtin_memcpy(&rl, &o.l, 16);
  float* dst = img + (int64_t)blockIdx.x * 2048;
  *reinterpret_cast<f32x4*>(dst + 4 * u) = rh;
  *reinterpret_cast<f32x4*>(dst + 1024 + 4 * u) = rl;
}

__global__ __launch_bounds__(512, 2) void sbr_w_split(SbrWArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 256;
  const int64_t k0 = (int64_t)blockIdx.y * a.kch;
  const int64_t kleft = a.np - k0;
  const int nkt = (int)((kleft < a.kch ? kleft : a.kch) / 32);
  const float sA = a.sc[0], inv = a.sc[1];
  // staging: wave w moves the 8-row groups 4 w .. 4 w + 3 of the A22 tile (chunk q of row r lands in slot q ^ ((r >> 1) & 7): the
  // layout of gemm_nt_big) and its 1 KB of the V image
  const int srow = lane >> 3, sq = lane & 7;
  const float* srcA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    int64_t ra = m0 + r;
    if (ra > a.np - 1) ra = a.np - 1;
    srcA[i] = a.A22 + ra * a.lda + k0 + 4 * (sq ^ ((r >> 1) & 7));
  }
  const float* srcV = a.Vimg + (k0 / 32) * 2048 + 4 * tid;
  auto stage = [&](int buf, int kt) {
    float* As = wl + buf * SBR_W_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcA[i] + (int64_t)kt * 32), (lds_void*)(As + (wid * 4 + i) * 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(srcV + (int64_t)kt * 2048), (lds_void*)(As + 8192 + wid * 256), 16, 0, 0);
  };
  // wave w owns the rows 32 w .. 32 w + 31 of the tile (two 16-row fragments) and all 64 columns of W (four fragments).
  // The matrix instruction takes V as its first operand: D[m = column of W][n = row of A22], so a lane's four results are four
  // consecutive columns of one row of W (one 16-byte store).
  int offA[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wid * 32 + 16 * i + l15, sw = (r >> 1) & 7;
    offA[i][0] = r * 32 + (((2 * g) ^ sw) << 2);
    offA[i][1] = r * 32 + (((2 * g + 1) ^ sw) << 2);
  }
  const int offV = 8192 + 4 * (g * 64 + l15);
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nkt > 0) stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const float* S = wl + buf * SBR_W_STAGE;
    f32x4 x[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      x[i][0] = *reinterpret_cast<const f32x4*>(S + offA[i][0]);
      x[i][1] = *reinterpret_cast<const f32x4*>(S + offA[i][1]);
    }
    SbrHL8 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 rh = *reinterpret_cast<const f32x4*>(S + offV + 64 * j);
      const f32x4 rl = *reinterpret_cast<const f32x4*>(S + offV + 1024 + 64 * j);
      __builtin_memcpy(&v[j].h, &rh, 16);
      __builtin_memcpy(&v[j].l, &rl, 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const SbrHL8 b = sbr_cat(sbr_split_pk(x[i][0] * sA), sbr_split_pk(x[i][1] * sA));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[j].h, b.h, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[j].l, b.h, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[j].h, b.l, acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* Ws = a.Wp + (int64_t)blockIdx.y * a.slab;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t row = m0 + wid * 32 + 16 * i + l15;
    if (row < a.np) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(Ws + row * SB + 16 * j + 4 * g) = acc[i][j] * inv;
    }
  }
}

// ---- host driver ------------------------------------------------------------------------------------------------------
// A: n x n fp32 row-major, full symmetric storage, n a multiple of SB. On return: lower band (|i - j| <= SB) = the band
// matrix, upper part = the panel reflectors V_p, Tall[p][SB][SB] = their T factors. *breakdown (host) != 0: a panel was
// numerically rank deficient for the Cholesky QR -- the result must not be used.
int sy2sb_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, float* Tall, int* breakdown) {
  if (n % SB != 0 || n < SB) return ctx->fail(SCLENS_ERR_ARG, "sy2sb_f32: the order must be a positive multiple of 64");
  if (lda % 4 != 0 || lda < n || (reinterpret_cast<uintptr_t>(A) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "sy2sb_f32: A must be 16-byte aligned with lda a multiple of 4");
  StageTimer tm(ctx, "sy2sb");
  ctx->q1p_n = -1;  // new reflectors: group data prepared for the previous ones is void
  const int64_t npan = n / SB - 1;  // the last diagonal block needs no reduction
  const int64_t ldw = round_up(n, 64);
  const int64_t maxparts = (n + SBR_GCH - 1) / SBR_GCH + 1;
  const int S = 16;  // at most this many K-slices of the skinny product W = A22 V
  SCL_WS(ctx, part, double, "sbr.part", maxparts * SB * SB);
  SCL_WS(ctx, psum, double, "sbr.psum", SB * SB);
  SCL_WS(ctx, Mat, double, "sbr.M", 2 * SB * SB);   // M | Sh
  SCL_WS(ctx, V1, float, "sbr.V1", 2 * SB * SB);    // V1 | Rh
  SCL_WS(ctx, Wp, float, "sbr.Wp", (int64_t)(S + 1) * SB * ldw);  // + one slab: the pending update's share of W
  SCL_WS(ctx, Vr, float, "sbr.Vr", SB * ldw);
  SCL_WS(ctx, Yr, float, "sbr.Yr", SB * ldw);
  // operands of the trailing update, one row per ABSOLUTE matrix row, two slots of 2 SB columns: [V Z]_even | [V Z]_odd
  constexpr int64_t LDU = 4 * SB;
  constexpr int GSL = 64;  // K-slices of the small product G = WV' V
  SCL_WS(ctx, VW, float, "sbr.VW2", n * LDU);
  SCL_WS(ctx, WV, float, "sbr.WV2", n * LDU);
  SCL_WS(ctx, Gp, float, "sbr.Gp", (int64_t)(GSL + 1) * SB * 2 * SB);
  // Trailing updates of at least `split_min` rows run on the fp16 matrix cores from operands split into two fp16 pieces
  // (gemm_split_update in gram_bits.hip: 22-bit operands, fp32 accumulation started from C -- the rank-128 / rank-256 update is then
  // C traffic only: 27 us of fp32 matrix-pipe time per 128 of K and 256 x 256 tile become 5). Context option sy2sb_split_min = 0 (or precision = 0): fp32 products.
  const int64_t split_min = std::max<int64_t>(512, ctx->opt.eff_sy2sb_split_min());
  // Separate power-of-two scales for the reflector columns (entries up to 1) and the Z columns (entries ~ the norm of the matrix) of
  // the update's operands (default since round 4: first run on hardware there, test_sy2sb_split_update_with_separate_scales at norms
  // 1, 2^14, 2^20). Context option sy2sb_split_scales = 1: one scale for both, as in round 3 -- accurate only while the norm of the matrix
  // stays below ~2^12, which a drop-in for `_get_eigen` cannot assume.
  const int split_scales = ctx->opt.sy2sb_split_scales == 1 ? 1 : 2;
  const bool any_split = n >= split_min;
  void* imgP = any_split ? ctx->workspace("sbr.imgP", split_image_bytes(n, LDU)) : nullptr;
  void* imgQ = any_split ? ctx->workspace("sbr.imgQ", split_image_bytes(n, LDU)) : nullptr;
  float* imgS = any_split ? static_cast<float*>(ctx->workspace("sbr.imgS", 4 * sizeof(float))) : nullptr;
  if (any_split && (!imgP || !imgQ || !imgS)) return SCLENS_ERR_OOM;
  SCL_WS(ctx, flag, int, "sbr.flag", 4);
  // Largest |Z| of a panel (slot 0 / 1 of a delayed pair), left by the kernel that writes Z: the scales of the split update's operands
  // then need no pass over the operands, no memset and no scale kernel (three launches of ~40 us per update on the main stream). The V
  // columns take the fixed scale 2^13. Context option sy2sb_zmax = 0: the largest entries by a pass over the operands (until round 4).
  unsigned* zmax = nullptr;
  if (any_split && split_scales == 2 && ctx->opt.sy2sb_zmax != 0) {
    zmax = static_cast<unsigned*>(ctx->workspace("sbr.zmax", 4 * sizeof(unsigned)));
    if (!zmax) return SCLENS_ERR_OOM;
  }
  hipStream_t st = ctx->stream;
  SCL_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int) * 4, st));
  // W = A22 V from fp16 pieces (sbr_w_split) while the trailing matrix has at least `wsplit_min` rows: follows the switch of the
  // trailing updates; context option sy2sb_wsplit_min = 0: fp32 product, = r: from r rows
  const int64_t wsplit_min = std::max<int64_t>(2 * SB, ctx->opt.eff_sy2sb_wsplit_min());
  const bool any_wsplit = n - SB >= wsplit_min;
  const int ws_slots = 512;  // resident workgroups the split-K slices are chosen for (256 and 1 024 were 7 and 11 ms slower)
  float* Vimg = nullptr;
  float* wsc = nullptr;
  if (any_wsplit) {
    Vimg = static_cast<float*>(ctx->workspace("sbr.Vimg", sizeof(float) * (size_t)(n * SB)));
    wsc = static_cast<float*>(ctx->workspace("sbr.wsc", 4 * sizeof(float)));
    unsigned* wbound = static_cast<unsigned*>(ctx->workspace("sbr.wbound", 4 * sizeof(unsigned)));
    if (!Vimg || !wsc || !wbound) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemsetAsync(wbound, 0, 4 * sizeof(unsigned), st));
    hipLaunchKernelGGL(sbr_row_abs_max, dim3((unsigned)n), dim3(256), 0, st, A, n, lda, wbound);
    hipLaunchKernelGGL(sbr_w_scale, dim3(1), dim3(1), 0, st, wbound, 1.f, 0.f, wsc);
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(sbr_w_split), 2 * SBR_W_STAGE * (int)sizeof(float)));
  }
  SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(sbr_panel_small), SBR_PANEL_LDS));
  // Look-ahead: panel p + 1 only needs the first SB columns of the trailing matrix of step p. Those are updated first
  // (a strip product, its transposed copy, and the diagonal block), then the panel is factored on a second stream while
  // the main stream applies the rest of the rank-128 update: the latency-bound panel algebra (Gram, SB x SB factorisations,
  // V = P M) leaves the critical path.
  SCL_TRY(sbr_ensure_aux(ctx));
  hipStream_t st2 = ctx->aux_stream;
  const bool lookahead = ctx->opt.sy2sb_lookahead != 0;
  // Delayed update (round 3). Per 256 x 256 tile a rank-128 update costs 20 us of C traffic + 17 us of prologue against 27 us
  // of MFMA work (profiles/r03_update_gemm_decomposition.log), so two panels' updates applied as ONE rank-256 update save a third
  // of the update time. An EVEN panel p therefore updates only the columns panel p + 1 is factored from (the look-ahead strip)
  // and leaves the rest of the trailing matrix stale; the ODD panel p + 1 takes the missing share of its W = A22 V from the
  // pending operands, W += [V Z]_p ([-Z -V]_p' V_{p+1}) (two small products, an extra slab of the split-K sum), and applies
  // both panels' operands at once (K = 256). The price: nothing runs beside the factorisation of panel p + 1 on the second
  // stream (~0.4 ms per pair at full size), which is why the net gain is 14 of the 76 ms the update kernels save
  // (profiles/r03_sy2sb_delayed_update.log; a variant that kept ~1000 tiles of the even panel's update immediate to cover the
  // factorisation was slower than this one: its K = 128 launches and the misaligned 192-column remainder cost more).
  const bool delay_ok = lookahead && ctx->opt.sy2sb_delay != 0;
  // pair = 2 max(U, F) before, F + max(1.32 U, F) + c now (U: rank-128 update, F ~ 0.39 ms: factorisation, c ~ 0.09 ms: the two
  // small products): pays from U ~ 0.7 ms, i.e. from a trailing matrix of order ~19 000 (context option sy2sb_delay_min)
  const int64_t delay_min = std::max<int64_t>(4 * SB + 1, ctx->opt.sy2sb_delay_min);  // round 4: the factorisation an even panel exposes got cheaper (0.39 -> 0.25 ms): 457.6 ms at 18 432, 452.5 at 12 288 (r4m)
  unsigned long long* pprof = nullptr;  // context option panel_prof = 1: per-phase shader clocks of sbr_panel_small on stderr
  {
    if (ctx->opt.panel_prof > 0) {
      pprof = static_cast<unsigned long long*>(ctx->workspace("sbr.pprof", 16 * sizeof(unsigned long long)));
      if (!pprof) return SCLENS_ERR_OOM;
      SCL_HIP(ctx, hipMemsetAsync(pprof, 0, 16 * sizeof(unsigned long long), st));
      SCL_HIP(ctx, hipStreamSynchronize(st));
    }
  }
  // The look-ahead strip's diagonal block used to be a launch of its own (lower + mirror, one workgroup: 41 us of latency per panel
  // on the main stream, 17 ms per reduction); context option sy2sb_fold_diag = 0 restores it
  const bool fold_diag = ctx->opt.sy2sb_fold_diag != 0;
  bool pending = false;  // the previous panel's bulk update is outstanding (its operands sit in slot 0)
  auto factor_panel = [&](int64_t p, hipStream_t s_) -> int {
    const int64_t c0 = p * SB, r0 = c0 + SB, np = n - r0;
    float* Pt = A + c0 * lda + r0;  // transposed panel: Pt[j][i] = A[c0 + j][r0 + i] = P[i][j] (symmetric storage)
    float* Tp = Tall + p * SB * SB;
    const int nparts = (int)((np + SBR_GCH - 1) / SBR_GCH);
    const unsigned rtiles = (unsigned)((np + 4 * SBR_RT - 1) / (4 * SBR_RT));  // workgroups of four 16-position wave tiles
    hipLaunchKernelGGL((sbr_gram64<true>), dim3(nparts), dim3(256), SBR_GRAM_LDS, s_, Pt, lda, (const float*)nullptr, np, part);
    hipLaunchKernelGGL(sbr_sum_parts, dim3(SB * SB / 256), dim3(256), 0, s_, part, nparts, psum);
    SbrSmall sm{Mat, V1, Tp, V1 + SB * SB, flag, pprof};
    if (np == SB)  // last panel: may contain the zero rows of the padding
      hipLaunchKernelGGL(sbr_panel_house, dim3(1), dim3(64), 0, s_, Pt, lda, sm);
    else
      hipLaunchKernelGGL(sbr_panel_small, dim3(1), dim3(256), SBR_PANEL_LDS, s_, psum, Pt, lda, sm);
    // V = [V1; P2 M] in the transposed storage (in place) and row-major; Rh into the band block of the panel
    if (np > SB) hipLaunchKernelGGL(sbr_vmul_f64, dim3(rtiles), dim3(256), 0, s_, Pt, lda, Mat, np, Vr);
    hipLaunchKernelGGL(sbr_top_block, dim3(1), dim3(256), 0, s_, Pt, lda, V1, V1 + SB * SB, A + r0 * lda + c0, Vr);
    if (any_wsplit && np >= wsplit_min) hipLaunchKernelGGL(sbr_v_image, dim3((unsigned)(np / 32)), dim3(256), 0, s_, Vr, Vimg);
    return SCLENS_OK;
  };
  SCL_TRY(factor_panel(0, st));
  for (int64_t p = 0; p < npan; ++p) {
    const int64_t c0 = p * SB, r0 = c0 + SB, np = n - r0;
    float* Pt = A + c0 * lda + r0;
    float* Tp = Tall + p * SB * SB;
    const int nparts = (int)((np + SBR_GCH - 1) / SBR_GCH);
    const unsigned rtiles = (unsigned)((np + 4 * SBR_RT - 1) / (4 * SBR_RT));
    // W (n' x SB) = A22 V as an NT product (A22 is stored in full and symmetric: its rows are K-contiguous), 256 x 64 tiles,
    // K = n' split into S slices inside one launch (slab s = its own [n'][SB] partial, summed by the next kernel)
    float* A22 = A + r0 * lda + r0;
    const int64_t tiles_w = (np + 255) / 256;
    int Sw = sbr_pick_splits(tiles_w, S, np, (any_wsplit && np >= wsplit_min) ? ws_slots : 256);  // sbr_w_split: two workgroups per CU
    const int64_t kch = round_up((np + Sw - 1) / Sw, 32);
    Sw = (int)((np + kch - 1) / kch);
    if (any_wsplit && np >= wsplit_min) {
      SbrWArgs wa{A22, lda, Vimg, Wp, np, kch, (int64_t)SB * ldw, wsc};
      hipLaunchKernelGGL(sbr_w_split, dim3((unsigned)tiles_w, (unsigned)Sw), dim3(512), 2 * SBR_W_STAGE * sizeof(float), st, wa);
    } else {
      GemmArgs g{};
      g.P = A22; g.Q = Pt; g.C = Wp;
      g.M = np; g.N = SB; g.K = np;
      g.ldp = lda; g.ldq = lda; g.ldc = SB;
      g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
      g.splits = Sw; g.k_chunk = kch; g.c_split_off = (int64_t)SB * ldw;
      g.prefer_big = 1;
      SCL_TRY(gemm_f32(ctx, g));
    }
    int nsl_w = Sw;
    if (pending) {
      // G'[b][a] = sum_r V_{p+1}[r][b] WV_p[r][a] (64 x 128, the contraction over the n' rows split into slices), then the extra
      // slab of W: VW_p (n' x 128) G (128 x 64)
      const float* WVp = WV + r0 * LDU;  // slot 0, rows from this panel's r0 on
      const float* VWp = VW + r0 * LDU;
      int Sg = (int)std::min<int64_t>(GSL, std::max<int64_t>(1, np / 256));
      const int64_t gch = round_up((np + Sg - 1) / Sg, 16);
      Sg = (int)((np + gch - 1) / gch);
      float* Gs = Gp + (int64_t)GSL * SB * 2 * SB;
      {
        GemmArgs g{};
        g.P = Pt; g.Q = WVp; g.C = Gp;
        g.M = SB; g.N = 2 * SB; g.K = np;
        g.ldp = lda; g.ldq = LDU; g.ldc = 2 * SB;
        g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
        g.splits = Sg; g.k_chunk = gch; g.c_split_off = (int64_t)SB * 2 * SB;
        SCL_TRY(gemm_f32(ctx, g));
      }
      hipLaunchKernelGGL(sbr_sum_slabs, dim3(SB * 2 * SB / 256), dim3(256), 0, st, Gp, Sg, (int64_t)SB * 2 * SB, SB * 2 * SB, Gs);
      {
        GemmArgs g{};
        g.P = VWp; g.Q = Gs; g.C = Wp + (int64_t)Sw * SB * ldw;
        g.M = np; g.N = SB; g.K = 2 * SB;
        g.ldp = LDU; g.ldq = 2 * SB; g.ldc = SB;
        g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
        g.prefer_big = 1;
        SCL_TRY(gemm_f32(ctx, g));
      }
      nsl_w = Sw + 1;
    }
    // Y = W T; Sh = 1/2 T' (V'Y); Z = Y - V Sh and the row-major operands [V | Z], [Z | V] of the update (slot 1 when the
    // previous panel's operands are still pending in slot 0)
    const int slot = pending ? 1 : 0;
    float* VWs = VW + r0 * LDU + slot * 2 * SB;
    float* WVs = WV + r0 * LDU + slot * 2 * SB;
    hipLaunchKernelGGL((sbr_rmul_f32<2>), dim3(rtiles), dim3(256), 0, st, Wp, nsl_w, (int64_t)SB * ldw, Tp, (const double*)nullptr, np, Yr,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0);
    hipLaunchKernelGGL((sbr_gram64<false>), dim3(nparts), dim3(256), SBR_GRAM_LDS, st, Vr, (int64_t)SB, Yr, np, part);
    hipLaunchKernelGGL(sbr_sum_parts, dim3(SB * SB / 256), dim3(256), 0, st, part, nparts, psum);
    hipLaunchKernelGGL(sbr_small_s, dim3(1), dim3(256), 0, st, psum, Tp, Mat + SB * SB, zmax ? zmax + slot : (unsigned*)nullptr);
    hipLaunchKernelGGL((sbr_rmul_f32<3>), dim3(rtiles), dim3(256), 0, st, Vr, 1, (int64_t)0, (const float*)nullptr, Mat + SB * SB, np,
                       (float*)nullptr, Yr, VWs, WVs, LDU, zmax ? zmax + slot : (unsigned*)nullptr);
    // the update's operands: this panel's slot alone, or both slots (K = 4 SB) when the previous panel's update is pending
    const float* Up = VW + r0 * LDU;
    const float* Uq = WV + r0 * LDU;
    const int64_t Ku = pending ? 4 * SB : 2 * SB;
    auto update = [&](int64_t off, int64_t rows, int64_t cols, int lower) -> int {  // A22[off:off+rows, (lower ? off : 0) : +cols]
      if (lower && rows == cols && rows >= split_min) {  // the bulk of the update: split-fp16 products
        if (split_scales == 2) {  // one scale for the V columns, one for the Z columns (prepared in round 3, not yet the default)
          if (zmax) SCL_TRY(split_image_pair_zmax(ctx, Up + off * LDU, Uq + off * LDU, rows, Ku, LDU, (int)SB, imgP, imgQ, imgS, zmax, pending ? 2 : 1));
          else SCL_TRY(split_image_pair_scaled2(ctx, Up + off * LDU, Uq + off * LDU, rows, Ku, LDU, (int)SB, imgP, imgQ, imgS));
          return gemm_split_update(ctx, imgP, imgS, rows, imgQ, imgS + 2, rows, Ku, A22 + off * lda + off, lda, 1);
        }
        SCL_TRY(split_image_pair_scaled(ctx, Up + off * LDU, Uq + off * LDU, rows, Ku, LDU, imgP, imgQ, imgS));
        return gemm_split_update(ctx, imgP, imgS, rows, imgQ, imgS, rows, Ku, A22 + off * lda + off, lda, 1);
      }
      GemmArgs g{};
      g.P = Up + off * LDU; g.Q = Uq + (lower ? off : 0) * LDU; g.C = A22 + off * lda + (lower ? off : 0);
      g.M = rows; g.N = cols; g.K = Ku;
      g.ldp = LDU; g.ldq = LDU; g.ldc = lda;
      g.alpha = 1.f; g.beta = 1.f; g.q_kcontig = 1; g.lower = lower; g.colabsmax = nullptr;  // WV is stored negated
      g.acc_init = 1;
      g.prefer_big = 1;  // short K: bound by the traffic of C, whose mirrored half the large-tile kernel stores 16 bytes at a time
      return gemm_f32(ctx, g);
    };
    if (p + 1 < npan && lookahead && np > 2 * SB) {
      // (i) what panel p + 1 reads: the diagonal block (lower + mirror), the strip below it, and the strip's exact transpose
      if (fold_diag) {  // one launch for the diagonal block and the strip; the block's upper half then copied from its lower half
        SCL_TRY(update(0, np, SB, 0));
        hipLaunchKernelGGL(sbr_mirror_diag, dim3(1), dim3(256), 0, st, A22, lda);
      } else {
        SCL_TRY(update(0, SB, SB, 1));
        SCL_TRY(update(SB, np - SB, SB, 0));
      }
      SCL_TRY(transpose_f32(ctx, A22 + SB * lda, np - SB, SB, lda, A22 + SB, lda));
      SCL_HIP(ctx, hipEventRecord(ctx->aux_ev[0], st));
      SCL_HIP(ctx, hipStreamWaitEvent(st2, ctx->aux_ev[0], 0));
      SCL_TRY(factor_panel(p + 1, st2));
      SCL_HIP(ctx, hipEventRecord(ctx->aux_ev[1], st2));
      // (ii) the rest of the trailing matrix, concurrently with the factorisation of panel p + 1 -- or left to the next panel,
      //      while the trailing matrix is large enough for the saved pass over it to outweigh the exposed factorisation
      const bool delay = delay_ok && !pending && p + 2 < npan && np >= delay_min;
      if (!delay) SCL_TRY(update(SB, np - SB, np - SB, 1));
      SCL_HIP(ctx, hipStreamWaitEvent(st, ctx->aux_ev[1], 0));
      pending = delay;
    } else {
      SCL_TRY(update(0, np, np, 1));
      if (p + 1 < npan) SCL_TRY(factor_panel(p + 1, st));
      pending = false;
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  if (breakdown) {
    SCL_HIP(ctx, hipMemcpyAsync(breakdown, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
  }
  if (pprof) {
    unsigned long long h[11];
    SCL_HIP(ctx, hipStreamSynchronize(st2));
    SCL_HIP(ctx, hipMemcpy(h, pprof, sizeof(h), hipMemcpyDeviceToHost));
    static const char* nm[10] = {"loads", "Cholesky (64 steps)", "R^-1 (one wave)", "Q_top = P_top R^-1", "LU (64 steps)", "Rh, V1, U'",
                                 "(V1')^-1 (one wave)", "T = -U' (V1')^-1", "U'^-1 (one wave)", "M = R^-1 D U'^-1 + stores"};
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i) tot += h[i];
    fprintf(stderr, "[sbr_panel_small] n = %lld, %llu calls, %.0f clocks per call\n", (long long)n, h[10], (double)tot / (double)std::max<unsigned long long>(1, h[10]));
    for (int i = 0; i < 10; ++i)
      fprintf(stderr, "   %-28s %8.0f clocks (%4.1f %%)\n", nm[i], (double)h[i] / (double)std::max<unsigned long long>(1, h[10]), 100.0 * h[i] / (double)std::max<unsigned long long>(1, tot));
  }
  return SCLENS_OK;
}


// ---- first back-transformation: rows of Zt (eigenvectors of the band matrix) -> eigenvectors of A --------------------
// z_A = H_0 (H_1 (... H_{P-1} z_B)), H_p = I - V_p T_p V_p' on the coordinates >= r0_p. In row form, per panel from the
// last to the first:  Zt[:, r0:] -= ((Zt[:, r0:] V) T') V'  -- three GEMMs, the first one split over K in one launch and
// summed by the second through an S-fold replicated T (the scheme of ormtr_f32).
// Panels are applied in groups of Q1G = 4 or 8: H_a H_{a+1} .. H_{a+Q1G-1} = I - Vm Tm Vm' with Vm = [V_a .. ] (each later
// block starting 64 rows further down: a zero staircase) and the block upper triangular Tm whose diagonal blocks are the
// panels' own T factors and whose off-diagonal blocks follow the larft recurrence Tm[0:j, j] = -Tm[0:j, 0:j] (Vm[0:j]' V_j) T_j.
// A group therefore costs three products with a 64 Q1G-deep inner dimension instead of 3 Q1G with 64: the traffic of Z (read once
// by the first product, read + written by the last) per unit of work drops fourfold, which is what bounds the unmerged form.
constexpr int Q1G_MAX = 8;  // panels per group: Q1G = 4 or 8 at run time (context option q1_group), Q1W = 64 Q1G columns

// clean copies of the group's reflectors: Vm[c][i] (c = 64 q + j: reflector j of panel q, i = position relative to the FIRST
// panel's r0) and its transpose VmT[i][c]; entries above a panel's own start (i < 64 q) and rows of missing panels are zero
__global__ __launch_bounds__(256) void sbr_q1_build_vm(const float* __restrict__ A, int64_t lda, int64_t c0, int cnt, int64_t np,
                                                       float* __restrict__ Vm, int64_t ldv, float* __restrict__ VmT, int Q1W) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int cb = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int c = cb + r, q = c >> 6;
    const int64_t i = i0 + tx;
    float v = 0.f;
    if (q < cnt && i < np && i >= (int64_t)q * SB) v = A[(c0 + c) * lda + (c0 + SB) + i];
    tile[r][tx] = v;
    if (i < ldv) Vm[(int64_t)c * ldv + i] = (i < np) ? v : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = i0 + r;
    if (i < np) VmT[i * Q1W + cb + tx] = tile[tx][r];
  }
}

// G = sum of the split-K slabs of the group's Gram matrix, in a fixed order
__global__ __launch_bounds__(256) void sbr_q1_sum_g(const float* __restrict__ Gp, int nslab, float* __restrict__ G, int Q1W) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (int q = 0; q < nslab; ++q) s += Gp[(int64_t)q * Q1W * Q1W + idx];
  G[idx] = s;
}

// Tm, one block diagonal at a time (level 0 = the panels' own T factors, level d = the blocks (a, a + d)):
//   Tm[a][b] = -( sum_{q = a}^{b-1} Tm[a][q] G[q][b] ) T_b,
// one workgroup per 64 x 64 block, operands staged in LDS, fp64 accumulation. Launched once per level (the blocks of a level
// only need lower levels).
__global__ __launch_bounds__(256) void sbr_q1_merge_level(const float* __restrict__ G, const float* __restrict__ Tp, int cnt, int level,
                                                          float* __restrict__ Tm, int Q1W) {
  __shared__ float Ls[SB][SB + 1], Rs[SB][SB + 1];
  const int a = blockIdx.x, b = a + level, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  if (level == 0) {  // diagonal blocks, and zero everything else of block row a (upper part is filled by the later levels)
    for (int idx = tid; idx < SB * Q1W; idx += 256) {
      const int r = idx / Q1W, c = idx % Q1W;
      Tm[(a * SB + r) * Q1W + c] = ((c >> 6) == a && a < cnt) ? Tp[(int64_t)a * SB * SB + r * SB + (c & 63)] : 0.f;
    }
    return;
  }
  if (b >= cnt) return;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int q = a; q < b; ++q) {  // acc += Tm[a][q] G[q][b]
    for (int idx = tid; idx < SB * SB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      Ls[r][c] = Tm[(a * SB + r) * Q1W + q * SB + c];
      Rs[r][c] = G[(q * SB + r) * Q1W + b * SB + c];
    }
    __syncthreads();
    for (int k = 0; k < SB; ++k) {
      float l[4], r4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) l[u] = Ls[4 * ti + u][k];
#pragma unroll
      for (int v = 0; v < 4; ++v) r4[v] = Rs[k][4 * tj + v];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += (double)l[u] * (double)r4[v];
    }
    __syncthreads();
  }
  // X = acc (to LDS), then Tm[a][b] = -X T_b
  for (int idx = tid; idx < SB * SB; idx += 256) Rs[idx >> 6][idx & 63] = Tp[(int64_t)b * SB * SB + idx];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) Ls[4 * ti + u][4 * tj + v] = (float)acc[u][v];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k = 0; k < SB; ++k) {
    float l[4], r4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) l[u] = Ls[4 * ti + u][k];
#pragma unroll
    for (int v = 0; v < 4; ++v) r4[v] = Rs[k][4 * tj + v];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += (double)l[u] * (double)r4[v];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) Tm[(a * SB + 4 * ti + u) * Q1W + b * SB + 4 * tj + v] = (float)(-acc[u][v]);
}

// Ws[r][c] = sum_s W1[r][s][c] (the split-K slabs of the first product, fixed order), four columns per thread.
// (Round 2 summed the slabs inside the second product by contracting against an S-fold replicated Tm: 2 m Q1W^2 S flop per
// group on the 128 x 128 kernel, ~50 ms per back-transformation at n = 30 016; this pass reads m S Q1W floats once.)
__global__ __launch_bounds__(256) void sbr_q1_sum_w(const float* __restrict__ W1, int S, int64_t m, int Q1W, float* __restrict__ Ws) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // index of a float4 of Ws
  const int64_t q4 = Q1W / 4;
  if (i >= m * q4) return;
  const int64_t r = i / q4, c4 = i % q4;
  const f32x4* src = reinterpret_cast<const f32x4*>(W1 + r * (int64_t)S * Q1W) + c4;
  f32x4 acc = src[0];
  for (int q = 1; q < S; ++q) acc += src[(int64_t)q * q4];
  reinterpret_cast<f32x4*>(Ws)[i] = acc;
}

// ---- the group data of the first back-transformation, prepared ahead --------------------------------------------------------------
// Per group of Q1G panels the apply loop below needs the clean reflector block Vm, its transpose as a split-fp16 image, and the
// merged T factor: a Gram product of the group's reflectors, a slab sum and Q1G level kernels of ~40 us each that only depend on the
// band reduction's output -- 0.7 ms per group, 41 of the stage's 210 ms at order 30 016, all of it a chain of small launches in
// front of the group's three large products. sbr_q1_prepare builds the data of ALL groups on the auxiliary stream right after the
// band reduction, beside the bulge chase (a latency chain that leaves most of the chip idle); sbr_apply_q1 then only waits for one
// event. 3.7 GB of workspace at order 30 016 (context option q1_prep = 0: off, the groups are prepared inline as before).
struct Q1Layout {
  int Q1G = 0, Q1W = 0;
  int64_t ngrp = 0, vm_total = 0, img_total = 0, vimg_total = 0;
  // per group: rows below the group's first panel, stride of Vm, offsets of Vm (floats), of the split image of VmT and of Vm (bytes)
  std::vector<int64_t> np, ldv, vm_off, img_off, vimg_off;
};
static Q1Layout sbr_q1_layout(int64_t n, int Q1G) {
  Q1Layout L;
  L.Q1G = Q1G;
  L.Q1W = Q1G * SB;
  const int64_t npan = n / SB - 1;
  L.ngrp = (npan + Q1G - 1) / Q1G;
  for (int64_t g = 0; g < L.ngrp; ++g) {
    const int64_t np = n - (g * Q1G * SB + SB);
    L.np.push_back(np);
    L.ldv.push_back(round_up(np, 32));
    L.vm_off.push_back(L.vm_total);
    L.img_off.push_back(L.img_total);
    L.vimg_off.push_back(L.vimg_total);
    L.vm_total += (int64_t)L.Q1W * round_up(np, 32);
    L.img_total += (int64_t)round_up((int64_t)split_image_bytes(np, L.Q1W), 256);
    L.vimg_total += (int64_t)round_up((int64_t)split_image_bytes(L.Q1W, np), 256);
  }
  return L;
}
static int64_t sbr_q1_split_min(const Ctx* ctx) {  // the read-modify-write product of a group runs from split images from this many vectors / rows
  return ctx->opt.eff_q1_split_min();  // 0: off; N > 0: from N vectors and N rows (default 1024)
}

// Vm, VmT, Tm of the group that starts at panel p0 (cnt panels), on ctx->stream; Gp / Gs: scratch
static int sbr_q1_group_data(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* Tall, int64_t p0, int cnt, int Q1G, float* Vm,
                             int64_t ldv, float* VmT, float* Gp, float* Gs, float* Tm) {
  const int Q1W = Q1G * SB, SG = 64;  // K-slices of the Gram product of the group's reflectors (few output tiles: the slices are the parallelism)
  const int64_t c0 = p0 * SB, np = n - (c0 + SB);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(sbr_q1_build_vm, dim3((unsigned)((ldv + 31) / 32), Q1W / 32), dim3(256), 0, st, A, lda, c0, cnt, np, Vm, ldv, VmT, Q1W);
  {  // Gram matrix of the group's reflectors, split over K
    GemmArgs gm{};
    gm.P = Vm; gm.Q = Vm; gm.C = Gp;
    gm.M = Q1W; gm.N = Q1W; gm.K = np;
    gm.ldp = ldv; gm.ldq = ldv; gm.ldc = Q1W;
    gm.alpha = 1.f; gm.beta = 0.f; gm.q_kcontig = 1; gm.lower = 0; gm.colabsmax = nullptr;
    gm.splits = SG; gm.k_chunk = round_up((np + SG - 1) / SG, 32); gm.c_split_off = (int64_t)Q1W * Q1W;
    SCL_TRY(gemm_f32(ctx, gm));
  }
  hipLaunchKernelGGL(sbr_q1_sum_g, dim3(Q1W * Q1W / 256), dim3(256), 0, st, Gp, SG, Gs, Q1W);
  for (int level = 0; level < cnt; ++level)
    hipLaunchKernelGGL(sbr_q1_merge_level, dim3(level == 0 ? Q1G : Q1G - level), dim3(256), 0, st, Gs, Tall + p0 * SB * SB, cnt, level, Tm, Q1W);
  return SCLENS_OK;
}

static int sbr_q1_prepare_impl(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* Tall) {
  ctx->q1p_n = -1;
  const int64_t npan = n / SB - 1, q1_min = sbr_q1_split_min(ctx);
  const int q1g_env = (int)ctx->opt.q1_group;
  // for the block size of MANY vectors (8 panels); a later call with few vectors (4 panels per group) prepares its groups inline
  if (ctx->opt.q1_prep == 0 || !ctx->q2_prebuild || npan < 64 || q1_min <= 0 || (q1g_env != 0 && q1g_env != 8) || n % SB != 0) return SCLENS_OK;
  const int Q1G = 8;
  const Q1Layout L = sbr_q1_layout(n, Q1G);
  const int Q1W = L.Q1W, SG = 64;
  SCL_WS(ctx, VmAll, float, "sbr.q1pVm", L.vm_total);
  void* imgAll = ctx->workspace("sbr.q1pImg", (size_t)L.img_total);
  SCL_WS(ctx, TmAll, float, "sbr.q1pTm", L.ngrp * (int64_t)Q1W * Q1W);
  SCL_WS(ctx, SAll, float, "sbr.q1pS", L.ngrp * 8);
  void* vimgAll = ctx->workspace("sbr.q1pVimg", (size_t)L.vimg_total);  // split images of the Vm blocks (operand of W1 = Z Vm')
  if (!vimgAll) return SCLENS_ERR_OOM;
  SCL_WS(ctx, VmT, float, "sbr.q1pVmT", round_up(n, 32) * (int64_t)Q1W);
  SCL_WS(ctx, Gp, float, "sbr.q1pG", (int64_t)SG * Q1W * Q1W);
  SCL_WS(ctx, Gs, float, "sbr.q1pGs", (int64_t)Q1W * Q1W);
  if (!imgAll) return SCLENS_ERR_OOM;
  SCL_TRY(sbr_ensure_aux(ctx));
  if (!ctx->q1_ev) SCL_HIP(ctx, hipEventCreateWithFlags(&ctx->q1_ev, hipEventDisableTiming));
  SCL_HIP(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));
  SCL_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[0], 0));
  // every helper launches on ctx->stream: the auxiliary stream stands in for it while the groups are enqueued
  struct Swap {
    Ctx* c;
    hipStream_t main;
    explicit Swap(Ctx* c_) : c(c_), main(c_->stream) { c->stream = c->aux_stream; c->swapped_main = main; }  // ctx_quiesce still covers it
    ~Swap() { c->stream = main; c->swapped_main = nullptr; }
  } swap(ctx);
  for (int64_t g = L.ngrp - 1; g >= 0; --g) {  // the order the apply loop consumes them in
    if (L.np[g] < q1_min) continue;            // short groups keep the fp32 product and are prepared inline
    const int cnt = (int)std::min<int64_t>(Q1G, npan - g * Q1G);
    float* Vm = VmAll + L.vm_off[g];
    SCL_TRY(sbr_q1_group_data(ctx, A, n, lda, Tall, g * Q1G, cnt, Q1G, Vm, L.ldv[g], VmT, Gp, Gs, TmAll + g * (int64_t)Q1W * Q1W));
    SCL_TRY(split_image_scaled(ctx, VmT, L.np[g], Q1W, Q1W, static_cast<char*>(imgAll) + L.img_off[g], SAll + 8 * g));
    SCL_TRY(split_image_fixed(ctx, Vm, Q1W, L.np[g], L.ldv[g], static_cast<char*>(vimgAll) + L.vimg_off[g], SAll + 8 * g + 4, 8192.f));
  }
  SCL_HIP(ctx, hipEventRecord(ctx->q1_ev, ctx->stream));
  SCL_HIP(ctx, hipGetLastError());
  ctx->q1p_n = n;
  ctx->q1p_g = Q1G;
  return SCLENS_OK;
}
// The preparation is an overlap optimisation with 3.7 GB of workspace of its own at order 30 016: when it fails (out of memory on a
// smaller part, or with three worker contexts) the call goes on and sbr_apply_q1 prepares every group inline, as without it.
int sbr_q1_prepare(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* Tall) {
  const int rc = sbr_q1_prepare_impl(ctx, A, n, lda, Tall);
  if (rc == SCLENS_OK) return rc;
  ctx_quiesce(ctx);
  (void)hipGetLastError();
  ctx->err.clear();
  ctx->q1p_n = -1;
  for (const char* w : {"sbr.q1pVm", "sbr.q1pImg", "sbr.q1pTm", "sbr.q1pS", "sbr.q1pVimg", "sbr.q1pVmT", "sbr.q1pG", "sbr.q1pGs"}) ctx->release(w);
  return SCLENS_OK;
}

int sbr_apply_q1(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* Tall, float* Zt, int64_t m, int64_t ldz) {
  if (m <= 0) return SCLENS_OK;
  if (n % SB != 0) return ctx->fail(SCLENS_ERR_ARG, "sbr_apply_q1: the order must be a multiple of 64");
  if (ldz % 4 != 0 || (reinterpret_cast<uintptr_t>(Zt) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "sbr_apply_q1: Zt must be 16-byte aligned with ldz a multiple of 4");
  StageTimer tm(ctx, "sbr_q1");
  const int64_t npan = n / SB - 1;
  if (npan <= 0) return SCLENS_OK;
  // panels per block reflector: the three products of a group contract over 64 Q1G columns; the large-tile kernel reaches
  // 82 / 100 TF/s at K = 256 / 512 with a read-modify-write of C (scripts/perf_update.py), so wide groups pay for many vectors
  const int q1g_env = (int)ctx->opt.q1_group;
  const int Q1G = (q1g_env == 4 || q1g_env == 8) ? q1g_env : (m >= 2048 && npan >= 64 ? 8 : 4);
  const int Q1W = Q1G * SB;
  const int64_t ngrp = (npan + Q1G - 1) / Q1G;
  const int64_t ldv = round_up(n, 32);
  const int64_t tiles_w1 = ((m + 255) / 256) * (Q1W / 256);
  const int SMAX = 16;  // upper bound of the K-slices of W1 (the workspaces are sized for it); the count is chosen per group
  const int SG = 64;
  // The read-modify-write product of a group, Zt += W2 Vm, on the fp16 matrix cores from split operands when it is large enough
  // (gemm_split_update: 22-bit operands, fp32 accumulation, C added in the epilogue): at K = 512 the fp32 matrix-pipe time is four
  // fifths of the product. Context option q1_split_min = 0: fp32 products.
  const int64_t q1_min = sbr_q1_split_min(ctx);
  const bool q1_split = q1_min > 0 && m >= q1_min;
  // groups prepared ahead on the auxiliary stream (sbr_q1_prepare): valid for this order, this group size and the split product
  const bool prepared = q1_split && ctx->q1p_n == n && ctx->q1p_g == Q1G && ctx->q1_ev;
  Q1Layout L;
  float *VmAll = nullptr, *TmAll = nullptr, *SAll = nullptr;
  char *imgAll = nullptr, *vimgAll = nullptr;
  if (prepared) {
    L = sbr_q1_layout(n, Q1G);
    auto ws = [&](const char* name) -> void* { return ctx->ws.count(name) ? ctx->ws.at(name).first : nullptr; };
    VmAll = static_cast<float*>(ws("sbr.q1pVm"));
    imgAll = static_cast<char*>(ws("sbr.q1pImg"));
    TmAll = static_cast<float*>(ws("sbr.q1pTm"));
    SAll = static_cast<float*>(ws("sbr.q1pS"));
    vimgAll = static_cast<char*>(ws("sbr.q1pVimg"));
    if (!VmAll || !imgAll || !TmAll || !SAll || !vimgAll) return ctx->fail(SCLENS_ERR_STATE, "sbr_apply_q1: prepared group data missing");
    SCL_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->q1_ev, 0));
  }
  SCL_WS(ctx, Vm, float, "sbr.Vm", Q1W * ldv);
  SCL_WS(ctx, VmT, float, "sbr.VmT", ldv * Q1W);
  SCL_WS(ctx, Gp, float, "sbr.q1G", (int64_t)SG * Q1W * Q1W);
  SCL_WS(ctx, Tm, float, "sbr.q1T", (int64_t)Q1W * Q1W);
  SCL_WS(ctx, Gs, float, "sbr.q1Gs", (int64_t)Q1W * Q1W);
  SCL_WS(ctx, W1, float, "sbr.W1", m * (int64_t)SMAX * Q1W);
  SCL_WS(ctx, Ws, float, "sbr.Ws", m * (int64_t)Q1W);
  SCL_WS(ctx, W2, float, "sbr.W2", m * (int64_t)Q1W);
  void* imgW = q1_split ? ctx->workspace("sbr.q1imgW", split_image_bytes(m, Q1W)) : nullptr;
  void* imgV = q1_split ? ctx->workspace("sbr.q1imgV", split_image_bytes(n, Q1W)) : nullptr;
  float* imgS = q1_split ? static_cast<float*>(ctx->workspace("sbr.q1imgS", 16 * sizeof(float))) : nullptr;
  if (q1_split && (!imgW || !imgV || !imgS)) return SCLENS_ERR_OOM;
  // Round 4: the FIRST product of a group, W1 = Zt[:, r0:] Vm', from split images too (the largest product of the stage that was
  // still on the fp32 matrix cores: 0.11 of its 0.21 s). Rows of Zt are unit vectors and reflector entries are at most 1, so both
  // images take the fixed scale 2^13 -- no pass for the largest entry; the image of Zt's columns is formed per group (they change
  // with every group's update: one streaming pass, 8 bytes per entry), the image of Vm with the prepared group data or inline.
  // Context option q1_w1_split = 0: this product stays fp32.
  const int64_t ew1 = ctx->opt.q1_w1_split;
  const bool w1_split = q1_split && ew1 != 0;
  // (end of round 4) Z enters that product as it is and is split in registers by the kernel (gemm_split_nt_f32a): the image of Z was
  // a read and a write of the vector block per group, 62 ms of the 125 ms of this stage at n = 30 016. q1_w1_split = 2: the image.
  const bool w1_regs = w1_split && ew1 != 2;
  void* imgZ = (w1_split && !w1_regs) ? ctx->workspace("sbr.q1imgZ", split_image_bytes(m, n)) : nullptr;
  void* imgVm = w1_split ? ctx->workspace("sbr.q1imgVm", split_image_bytes(Q1W, n)) : nullptr;
  if (w1_split && ((!w1_regs && !imgZ) || !imgVm)) return SCLENS_ERR_OOM;
  hipStream_t st = ctx->stream;
  for (int64_t g = ngrp - 1; g >= 0; --g) {
    const int64_t p0 = g * Q1G;
    const int cnt = (int)std::min<int64_t>(Q1G, npan - p0);
    const int64_t c0 = p0 * SB, r0 = c0 + SB, np = n - r0;
    const int S = sbr_pick_splits(tiles_w1, SMAX, np);
    const bool split_g = q1_split && np >= q1_min;
    const bool ready = prepared && split_g;  // this group's Vm, Tm and the split image of VmT exist already
    const float* Vm_g = ready ? VmAll + L.vm_off[g] : Vm;
    const int64_t ldv_g = ready ? L.ldv[g] : ldv;
    const float* Tm_g = ready ? TmAll + g * (int64_t)Q1W * Q1W : Tm;
    if (!ready) SCL_TRY(sbr_q1_group_data(ctx, A, n, lda, Tall, p0, cnt, Q1G, Vm, ldv, VmT, Gp, Gs, Tm));
    const int64_t kch = round_up((np + S - 1) / S, 32);
    if (w1_split && split_g) {  // W1[m][s][Q1W] = split-K partials of Zt[:, r0:] Vm' on the fp16 matrix cores
      if (!w1_regs) SCL_TRY(split_image_fixed(ctx, Zt + r0, m, np, ldz, imgZ, imgS + 8, 8192.f));
      const void* ivm = imgVm;
      const float* svm = imgS + 12;
      if (ready) {
        ivm = vimgAll + L.vimg_off[g];
        svm = SAll + 8 * g + 4;
      } else {
        SCL_TRY(split_image_fixed(ctx, Vm_g, Q1W, np, ldv_g, imgVm, imgS + 12, 8192.f));
      }
      if (w1_regs) SCL_TRY(gemm_split_nt_f32a(ctx, Zt + r0, ldz, 8192.f, m, ivm, svm, Q1W, np, W1, (int64_t)S * Q1W, S, kch, Q1W));
      else SCL_TRY(gemm_split_nt(ctx, imgZ, imgS + 8, m, ivm, svm, Q1W, np, W1, (int64_t)S * Q1W, S, kch, Q1W));
    } else {  // W1[m][s][Q1W] = split-K partials of Zt[:, r0:] Vm'
      GemmArgs g1{};
      g1.P = Zt + r0; g1.Q = Vm_g; g1.C = W1;
      g1.M = m; g1.N = Q1W; g1.K = np;
      g1.ldp = ldz; g1.ldq = ldv_g; g1.ldc = (int64_t)S * Q1W;
      g1.alpha = 1.f; g1.beta = 0.f; g1.q_kcontig = 1; g1.lower = 0; g1.colabsmax = nullptr;
      g1.splits = S; g1.k_chunk = kch; g1.c_split_off = Q1W;
      g1.prefer_big = 1;
      SCL_TRY(gemm_f32(ctx, g1));
    }
    const float* Wsum = W1;
    if (S > 1) {
      hipLaunchKernelGGL(sbr_q1_sum_w, dim3((unsigned)((m * (Q1W / 4) + 255) / 256)), dim3(256), 0, st, W1, S, m, Q1W, Ws);
      Wsum = Ws;
    }
    {  // W2 = -(sum_s W1_s) Tm'
      GemmArgs g2{};
      g2.P = Wsum; g2.Q = Tm_g; g2.C = W2;
      g2.M = m; g2.N = Q1W; g2.K = Q1W;
      g2.ldp = Q1W; g2.ldq = Q1W; g2.ldc = Q1W;
      g2.alpha = -1.f; g2.beta = 0.f; g2.q_kcontig = 1; g2.lower = 0; g2.colabsmax = nullptr;  // W2 = -(...): g3 then adds
      SCL_TRY(gemm_f32(ctx, g2));
    }
    if (split_g) {  // Zt[:, r0:] += W2 Vm from split images
      SCL_TRY(split_image_scaled(ctx, W2, m, Q1W, Q1W, imgW, imgS));
      const void* iv = imgV;
      const float* sv = imgS + 4;
      if (ready) {
        iv = imgAll + L.img_off[g];
        sv = SAll + 8 * g;
      } else {
        SCL_TRY(split_image_scaled(ctx, VmT, np, Q1W, Q1W, imgV, imgS + 4));
      }
      SCL_TRY(gemm_split_update(ctx, imgW, imgS, m, iv, sv, np, Q1W, Zt + r0, ldz, 0));
    } else {  // Zt[:, r0:] += W2 Vm   (NT against the transposed copy; accumulators started from Zt)
      GemmArgs g3{};
      g3.P = W2; g3.Q = VmT; g3.C = Zt + r0;
      g3.M = m; g3.N = np; g3.K = Q1W;
      g3.ldp = Q1W; g3.ldq = Q1W; g3.ldc = ldz;
      g3.alpha = 1.f; g3.beta = 1.f; g3.q_kcontig = 1; g3.lower = 0; g3.colabsmax = nullptr;
      g3.acc_init = 1;
      g3.prefer_big = 1;
      SCL_TRY(gemm_f32(ctx, g3));
    }
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// ======================================================================================================================
// Stage 2: symmetric band (half-width SB) -> tridiagonal by bulge chasing. (Work in progress like stage 1: eigenvalues
// only so far -- the reflectors are stored for the second back-transformation, which does not exist yet.)
// Sweep s annihilates column s below the sub-diagonal with a reflector on rows s+1 .. s+SB and chases the bulge it creates
// down the band: task k of the sweep (rows r_k = s + 1 + k SB ...) right-applies the previous reflector to the
// off-diagonal block B_k, takes a new reflector from its first column, left-applies it, and applies it two-sidedly to the
// diagonal block D_k (Schwarz / Lang). Sweep s may run task k as soon as sweep s-1 has finished task k+1, so ~n/(2 SB)
// sweeps are in flight: one persistent kernel, sweep s on workgroup s mod G (all G resident), per-sweep progress counters
// with agent-scope release / acquire. Blocks live in LDS (2 x 64 x 64 floats); the band is packed as Bd[column][row - column]
// with room for the bulge (row - column <= 2 SB).
constexpr int LDB2 = 2 * SB + 4;  // floats per column of the packed band

__global__ void sbr_pack_band(const float* __restrict__ A, int64_t n, int64_t lda, float* __restrict__ Bd) {
  const int64_t j = blockIdx.x;
  for (int r = threadIdx.x; r < LDB2; r += blockDim.x)
    Bd[j * LDB2 + r] = (r <= SB && j + r < n) ? A[(j + r) * lda + j] : 0.f;
}


// One task = one workgroup step: 256 threads in two register mappings of a 64 x 64 block,
//   T1: lane = row i, wave = 16-column group jq   (global loads / stores are 256-byte runs per wave-instruction),
//   T2: lane = column j, wave = 16-row group       (the product v'B, reading the column-major LDS image with stride 65).
// Vectors (v, v_prev, w, z) live one entry per lane in EVERY wave (all four waves compute the reflector redundantly from the
// same numbers), so a product needs its vector as wave-uniform values: v_readlane, no LDS, no barrier. The block B is only
// read from LDS (never updated there): with z = v'B - tau_p (v'w) v_p' the two one-sided updates collapse into
// B <- B - tau_p w v_p' - tau v z', applied to the registers that hold the loaded block and stored straight to memory.
// Three workgroup barriers per task. Hand-off between sweeps (workgroups): every store of band data is write-through
// (`buffer_store ... sc1`), every storing wave drains `vmcnt(0)` before the barrier, one lane then stores the progress
// counter (sc1); the consumer polls that counter from one lane and reads band data only with `buffer_load ... sc1`
// (MI355X_MICROARCH.md, valid forms of an inter-workgroup hand-off, first table row). Out-of-range buffer offsets make
// masked lanes load 0 / store nothing without a branch. A bounded spin + a shared abort word keep a logic error from
// hanging the GPU.
struct SbrChaseArgs {
  float* Bd;
  int64_t n;
  float* V2;
  int64_t ldv2;
  float* TAU2;
  int64_t ldt;
  unsigned* done;   // [n] progress counters + [n] : abort word
};

__device__ __forceinline__ float sbr_rl(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
// Wave-wide sums on the DPP network (no LDS round trips: a ds_bpermute butterfly costs ~100 clocks per step, and a task has three
// sums on its critical path): xor-1 and xor-2 inside quads, half-row and row mirrors -> every lane holds its row's sum; the four row
// sums are combined as (r0 + r1) + (r2 + r3) from scalar registers. Every lane gets the same bits.
template <int CTRL>
__device__ __forceinline__ float sbr_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double sbr_dpp(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ float sbr_wave_sum(float x) {
  x += sbr_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += sbr_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += sbr_dpp<0x141>(x);  // row_half_mirror
  x += sbr_dpp<0x140>(x);  // row_mirror
  return (sbr_rl(x, 0) + sbr_rl(x, 16)) + (sbr_rl(x, 32) + sbr_rl(x, 48));
}
__device__ __forceinline__ double sbr_wave_sum(double x) {
  x += sbr_dpp<0xB1>(x);
  x += sbr_dpp<0x4E>(x);
  x += sbr_dpp<0x141>(x);
  x += sbr_dpp<0x140>(x);
  const long long b = __builtin_bit_cast(long long, x);
  double r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lo = __builtin_amdgcn_readlane((int)b, 16 * i), hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * i);
    r[i] = __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

__global__ __launch_bounds__(256) void sbr_chase(SbrChaseArgs a) {
#pragma clang fp contract(off)  // every fused multiply-add below is written out: sbr_chase and sbr_chase_mb give the same bits
  __shared__ float Bt[SB * 65], Dt[SB * 65];
  __shared__ float part[4 * SB], part2[4 * SB], partD[4 * SB];
  __shared__ int pd_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n = a.n;
  const unsigned nbytes = (unsigned)(n * LDB2 * sizeof(float));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.Bd, 0, nbytes, 0x00020000);
  const unsigned OOR = 0xfffffff0u;  // beyond num_records: loads return 0, stores are dropped
  unsigned* abort_w = a.done + n;
  for (int64_t s = blockIdx.x; s + 2 < n; s += gridDim.x) {
    const int K = sbr_tasks_of(s, n);
    const int Kprev = (s > 0) ? sbr_tasks_of(s - 1, n) : 0;
    int pd = (s > 0) ? 0 : 0x7fffffff;
    float vp = 0.f, tp = 0.f;
    for (int k = 0; k < K; ++k) {
      const int64_t rk = s + 1 + (int64_t)k * SB;
      const int L = (int)((n - rk < SB) ? n - rk : SB);
      const int need = (k + 2 < Kprev) ? k + 2 : Kprev;
      if (pd < need) {  // workgroup-uniform
        if (tid == 0) {
          int x = 0;
          for (unsigned spins = 0;; ++spins) {
            x = (int)__hip_atomic_load(a.done + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x >= need) break;
            if ((spins & 1023u) == 1023u) {
              if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24)) {
                __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x = -1;
                break;
              }
            }
            __builtin_amdgcn_s_sleep(1);
          }
          pd_s = x;
        }
        __syncthreads();
        pd = pd_s;
        if (pd < 0) return;  // aborted
      }
      // ---- loads (T1), write-through reads of another workgroup's stores
      float rb[16], rd[16];
      const unsigned colB0 = (unsigned)(rk - SB), colD0 = (unsigned)rk;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        const unsigned ob = (k > 0 && lane < L) ? ((colB0 + j) * LDB2 + SB + lane - j) * 4u : OOR;
        rb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ob, 0, 16));
        const unsigned od = (lane >= j && lane < L) ? ((colD0 + j) * LDB2 + lane - j) * 4u : OOR;
        rd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, od, 0, 16));
      }
      float ycol = 0.f;
      if (k == 0) {
        const unsigned oy = (lane < L) ? ((unsigned)s * LDB2 + 1 + lane) * 4u : OOR;
        ycol = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, oy, 0, 16));
      }
      // ---- LDS images (column-major: X[i][j] at j * 65 + i); D gets both triangles
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        Bt[j * 65 + lane] = rb[q];
        if (lane >= j) {
          Dt[j * 65 + lane] = rd[q];
          Dt[lane * 65 + j] = rd[q];
        }
      }
      // ---- w = B v_prev (partial sums over this wave's columns)
      float pw = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) pw = fmaf(rb[q], sbr_rl(vp, 16 * wq + q), pw);
      part[wq * SB + lane] = pw;
      __syncthreads();
      const float w = (part[lane] + part[SB + lane]) + (part[2 * SB + lane] + part[3 * SB + lane]);
      const float y = (k > 0) ? fmaf(-(tp * w), sbr_rl(vp, 0), Bt[lane]) : ycol;
      // ---- reflector from y (every wave, identical arithmetic)
      const float xi = (lane >= 1 && lane < L) ? y : 0.f;
      const double sg = sbr_wave_sum((double)xi * (double)xi);
      const float alpha = sbr_rl(y, 0);
      float tau = 0.f, beta = alpha, scale = 0.f;
      if (sg > 0.0) {
        const double nrm = sqrt(fma((double)alpha, (double)alpha, sg));
        beta = (float)((alpha >= 0.f) ? -nrm : nrm);
        tau = (beta - alpha) / beta;
        scale = 1.f / (alpha - beta);
      }
      const float v = (lane == 0) ? 1.f : xi * scale;
      // ---- z0 = v'B (T2: lane = column), D v (T1)
      float pz = 0.f;
      if (k > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) pz = fmaf(Bt[lane * 65 + 16 * wq + q], sbr_rl(v, 16 * wq + q), pz);
      }
      part2[wq * SB + lane] = pz;
      const float vw = sbr_wave_sum(v * w);
      float dd[16];
      float pdv = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dd[q] = Dt[(16 * wq + q) * 65 + lane];
        pdv = fmaf(dd[q], sbr_rl(v, 16 * wq + q), pdv);
      }
      partD[wq * SB + lane] = pdv;
      __syncthreads();
      const float z = fmaf(-(tp * vw), vp, (part2[lane] + part2[SB + lane]) + (part2[2 * SB + lane] + part2[3 * SB + lane]));
      float w2 = tau * ((partD[lane] + partD[SB + lane]) + (partD[2 * SB + lane] + partD[3 * SB + lane]));
      const float a2 = -0.5f * tau * sbr_wave_sum(v * w2);
      w2 = fmaf(a2, v, w2);
      // ---- B <- H (B H_prev), D <- H D H on the registers, stored write-through
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        if (k > 0) {
          float bn = fmaf(-(tau * v), sbr_rl(z, j), fmaf(-(tp * w), sbr_rl(vp, j), rb[q]));
          if (j == 0) bn = (lane == 0) ? beta : 0.f;
          const unsigned ob = (lane < L) ? ((colB0 + j) * LDB2 + SB + lane - j) * 4u : OOR;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bn), rs, ob, 0, 16);
        }
        const float dn = fmaf(-w2, sbr_rl(v, j), fmaf(-v, sbr_rl(w2, j), dd[q]));
        const unsigned od = (lane >= j && lane < L) ? ((colD0 + j) * LDB2 + lane - j) * 4u : OOR;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dn), rs, od, 0, 16);
      }
      if (wq == 0) {
        if (k == 0) {
          const unsigned oy = (lane < L) ? ((unsigned)s * LDB2 + 1 + lane) * 4u : OOR;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (lane == 0) ? beta : 0.f), rs, oy, 0, 16);
        }
        if (lane < L) a.V2[s * a.ldv2 + rk + lane] = v;
        if (lane == 0) a.TAU2[s * a.ldt + k] = tau;
      }
      vp = v;
      tp = tau;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have left
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.done + s, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- round 3: the same task arithmetic (bitwise the same results) with a shorter hand-off chain between sweeps.
// In sbr_chase a task is poll -> 32 loads -> compute -> 32 stores -> drain -> counter, and the successor sweep repeats that chain
// two tasks later: per sweep 2 x (two memory round trips + compute + drain) + one more round trip = 9.9 us at n = 30 016. Here
//  * the blocks of task k+1 are PREFETCHED at the end of task k: rows 0..62 of them were written by the predecessor's task k+1, which
//    the counter check that admitted OUR task k's prefetch one step earlier ... (see `need2`) already covers; only the last row
//    (global row r_k + 63 = first row of the predecessor's task k+2) is younger;
//  * that row travels in a MESSAGE: 65 {value, tag} pairs (8-byte single-copy-atomic stores / loads, tag = sweep + 1) that the
//    producer sends from registers the moment its update is computed, before its bulk stores; the consumer's waves poll the pairs
//    they need (one round trip, data included) and patch lane 63 of their registers. The producer does not store that row to the
//    band at all (the consumer's own store of its row 63 is the only writer: no write-write race);
//  * the counter (bulk visibility: stores drained) is still written after `vmcnt(0)` + barrier, but the same wait now also covers
//    the prefetch loads, and the consumer reads the counter with a load issued in the middle of its compute phase.
// Chain per task: message poll -> compute -> message send; in parallel: stores + prefetch -> drain -> counter.
constexpr int MBW = 72;  // 8-byte slots per message (65 used)
typedef unsigned u32x2 __attribute__((vector_size(8)));

struct SbrChaseMbArgs {
  float* Bd;
  int64_t n;
  float* V2;
  int64_t ldv2;
  float* TAU2;
  int64_t ldt;
  unsigned* done;            // [n] progress counters + [n] : abort word
  unsigned long long* MB;    // [R][kmax][MBW] messages of sweep s in ring slot s mod R
  int R, kmax;
  unsigned long long* prof;  // PROF only: [9]
};

// every lane of the wave polls the same counter; < 0: aborted
__device__ __forceinline__ int sbr_spin_flag(const unsigned* p, int need, unsigned* abort_w) {
  for (unsigned spins = 0;; ++spins) {
    const int x = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (x >= need) return x;
    if ((spins & 1023u) == 1023u) {
      if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24)) {
        __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// PROF: wave 0 samples the shader clock at eight points of every task and adds the differences into a.prof[0..7], the task count
// into a.prof[8] (context option chase_prof = 1 prints the averages).
template <bool PROF>
__global__ __launch_bounds__(256) void sbr_chase_mb(SbrChaseMbArgs a) {
#pragma clang fp contract(off)  // every fused multiply-add below is written out: sbr_chase and sbr_chase_mb give the same bits
  __shared__ float Bt[SB * 65], Dt[SB * 65];
  __shared__ float part[4 * SB], part2[4 * SB], partD[4 * SB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n = a.n;
  const unsigned nbytes = (unsigned)(n * LDB2 * sizeof(float));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.Bd, 0, nbytes, 0x00020000);
  const unsigned mbytes = (unsigned)((int64_t)a.R * a.kmax * MBW * 8);
  __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.MB, 0, mbytes, 0x00020000);
  const unsigned OOR = 0xfffffff0u;  // beyond num_records: loads return 0, stores are dropped
  // Block accesses: entry (row i = lane, column j = 16 wq + q) of a block whose first column is c sits at byte
  // ((c + 16 wq) LDB2 + r0 + lane - 16 wq) 4 + q QS: a per-lane base computed once per task + a per-instruction constant (the buffer
  // instruction's scalar offset), instead of five address / predicate instructions per access (measured: the 4 x 85 memory
  // instructions of a task took 45 % of its time). Masked lanes get OORB, far enough out that adding q QS cannot wrap.
  constexpr unsigned QS = (LDB2 - 1) * 4, OORB = 0x80000000u;
  const unsigned w16 = 16u * (unsigned)wq;
  const int dl = lane - 16 * wq;  // D is stored on and below the diagonal: entry (lane, 16 wq + q) exists for dl >= q
  unsigned* abort_w = a.done + n;
  // the pairs this lane polls: lanes 0..15 the D row (pair 1 + 16 wq + lane), lane 16 the predecessor's beta (pair 0)
  const bool act = lane <= 16;
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptasks = 0, pt = 0;
#define SBR_PROF_MARK(i)                                          \
  if (PROF) {                                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    pacc[i] += now_ - pt;                                         \
    pt = now_;                                                    \
  }
  const unsigned midx = (lane < 16) ? (1u + 16u * (unsigned)wq + (unsigned)lane) : 0u;
  for (int64_t s = blockIdx.x; s + 2 < n; s += gridDim.x) {
    const int K = sbr_tasks_of(s, n);
    const int Kprev = (s > 0) ? sbr_tasks_of(s - 1, n) : 0;
    const bool has_prev = s > 0, has_next = s + 3 < n;
    int pd = has_prev ? 0 : 0x7fffffff;
    const unsigned tag_in = (unsigned)s, tag_out = (unsigned)(s + 1);
    const unsigned mb_in = (unsigned)(((s + a.R - 1) % a.R) * a.kmax), mb_out = (unsigned)((s % a.R) * a.kmax);
    float vp = 0.f, tp = 0.f;
    float rb[16], rd[16], ycol = 0.f;
    u32x2 prn = {0u, 0u};  // first look at the next message, issued with the prefetch
    const unsigned* flagp = a.done + (has_prev ? s - 1 : 0);  // the predecessor's counter
    {  // blocks of task 0 (no off-diagonal block): the predecessor's task 0 must have drained
      if (has_prev) {
        pd = sbr_spin_flag(flagp, 1, abort_w);
        if (pd < 0) return;
      }
      const int64_t rk = s + 1;
      const int L = (int)((n - rk < SB) ? n - rk : SB);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        rb[q] = 0.f;
        const unsigned od = (lane >= j && lane < L) ? (((unsigned)rk + j) * LDB2 + lane - j) * 4u : OOR;
        rd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, od, 0, 16));
      }
      const unsigned oy = (lane < L) ? ((unsigned)s * LDB2 + 1 + lane) * 4u : OOR;
      ycol = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, oy, 0, 16));
    }
    for (int k = 0; k < K; ++k) {
      const int64_t rk = s + 1 + (int64_t)k * SB;
      const int L = (int)((n - rk < SB) ? n - rk : SB);
      const unsigned colB0 = (unsigned)(rk - SB), colD0 = (unsigned)rk;
      if (PROF) {
        pt = __builtin_amdgcn_s_memtime();
        ++ptasks;
      }
      // ---- the last row of the blocks: the message of the predecessor's task k+1 (exists exactly when L = SB there)
      if (has_prev && k + 1 < Kprev) {
        const unsigned mo = act ? ((mb_in + (unsigned)(k + 1)) * MBW + midx) * 8u : OOR;
        u32x2 pr = prn;
        for (unsigned spins = 0; !__all(!act || pr[1] == tag_in); ++spins) {
          if ((spins & 1023u) == 1023u) {
            if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24)) {
              __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              return;
            }
          }
          pr = __builtin_amdgcn_raw_buffer_load_b64(rm, mo, 0, 16);
        }
        const float mv = __builtin_bit_cast(float, pr[0]);
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // D[63][16 wq + q]: pair 1 + 16 wq + q (the last one is the predecessor's D[0][0])
          const float val = sbr_rl(mv, q);
          if (lane == 63) rd[q] = val;
        }
        const float b0 = sbr_rl(mv, 16);  // the predecessor's beta: B[63][63], or entry 63 of the sweep's first column
        if (k > 0) {
          if (wq == 3 && lane == 63) rb[15] = b0;
        } else if (lane == 63) {
          ycol = b0;
        }
      }
      SBR_PROF_MARK(0)  // message poll + patch
      // ---- LDS images (column-major: X[i][j] at j * 65 + i); D gets both triangles
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        Bt[j * 65 + lane] = rb[q];
        if (lane >= j) {
          Dt[j * 65 + lane] = rd[q];
          Dt[lane * 65 + j] = rd[q];
        }
      }
      // ---- w = B v_prev (partial sums over this wave's columns)
      float pw = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) pw = fmaf(rb[q], sbr_rl(vp, 16 * wq + q), pw);
      part[wq * SB + lane] = pw;
      __syncthreads();
      SBR_PROF_MARK(1)  // LDS images, w partials, barrier 1
      // the predecessor's counter for the prefetch at the end of this task: a first look now (back before it is needed), a second
      // one after the next barrier (younger, but the wave may have to wait for it)
      const bool more = k + 1 < K;
      const int need2 = (k + 2 < Kprev) ? k + 2 : Kprev;
      const unsigned fl_a = __hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float w = (part[lane] + part[SB + lane]) + (part[2 * SB + lane] + part[3 * SB + lane]);
      const float y = (k > 0) ? fmaf(-(tp * w), sbr_rl(vp, 0), Bt[lane]) : ycol;
      // ---- reflector from y (every wave, identical arithmetic)
      const float xi = (lane >= 1 && lane < L) ? y : 0.f;
      const double sg = sbr_wave_sum((double)xi * (double)xi);
      const float alpha = sbr_rl(y, 0);
      float tau = 0.f, beta = alpha, scale = 0.f;
      if (sg > 0.0) {
        const double nrm = sqrt(fma((double)alpha, (double)alpha, sg));
        beta = (float)((alpha >= 0.f) ? -nrm : nrm);
        tau = (beta - alpha) / beta;
        scale = 1.f / (alpha - beta);
      }
      const float v = (lane == 0) ? 1.f : xi * scale;
      const unsigned fl_b = __hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // ---- z0 = v'B (T2: lane = column), D v (T1)
      float pz = 0.f;
      if (k > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) pz = fmaf(Bt[lane * 65 + 16 * wq + q], sbr_rl(v, 16 * wq + q), pz);
      }
      part2[wq * SB + lane] = pz;
      const float vw = sbr_wave_sum(v * w);
      float dd[16];
      float pdv = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dd[q] = Dt[(16 * wq + q) * 65 + lane];
        pdv = fmaf(dd[q], sbr_rl(v, 16 * wq + q), pdv);
      }
      partD[wq * SB + lane] = pdv;
      __syncthreads();
      SBR_PROF_MARK(2)  // reflector, z / D v partials, barrier 2
      const float z = fmaf(-(tp * vw), vp, (part2[lane] + part2[SB + lane]) + (part2[2 * SB + lane] + part2[3 * SB + lane]));
      float w2 = tau * ((partD[lane] + partD[SB + lane]) + (partD[2 * SB + lane] + partD[3 * SB + lane]));
      const float a2 = -0.5f * tau * sbr_wave_sum(v * w2);
      w2 = fmaf(a2, v, w2);
      // ---- (1) row 0 to the successor. Wave 0 alone: its lane j forms B'[0][j] from the LDS image, w_0 and its own entries of
      // v_prev and z -- the same two fused multiply-adds on the same operands as lane 0 of the wave that owns column j (v_0 = 1) --
      // so one store carries the row; D'[0][0] is entry (lane 0, q 0) of the update below, formed here first.
      const bool send = k > 0 && has_next;
      const int lo = send ? 1 : 0;
      const unsigned mrow = (mb_out + (unsigned)k) * MBW;
      if (wq == 0) {
        float b0j = fmaf(-(tau * 1.f), z, fmaf(-(tp * sbr_rl(w, 0)), vp, Bt[lane * 65]));
        if (lane == 0) b0j = beta;
        const u32x2 pm = {__builtin_bit_cast(unsigned, b0j), tag_out};
        __builtin_amdgcn_raw_buffer_store_b64(pm, rm, send ? (mrow + (unsigned)lane) * 8u : OOR, 0, 16);
        const float d00 = fmaf(-w2, sbr_rl(v, 0), fmaf(-v, sbr_rl(w2, 0), dd[0]));
        const u32x2 pm2 = {__builtin_bit_cast(unsigned, d00), tag_out};
        __builtin_amdgcn_raw_buffer_store_b64(pm2, rm, (send && lane == 0) ? (mrow + 64u) * 8u : OOR, 0, 16);
      }
      SBR_PROF_MARK(3)  // z, w2 + message stores
      __builtin_amdgcn_sched_barrier(0);  // keep the wait for the counter loads behind the message
      {  // both looks are old enough to be back: no wait in the common case
        int xa, xb;  // volatile asm: the compiler's own readfirstlane floats up to the loads and waits for them there
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(xa) : "v"(fl_a));
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(xb) : "v"(fl_b));
        if (has_prev) pd = (xa > pd) ? xa : pd;
        if (has_prev) pd = (xb > pd) ? xb : pd;
      }
      if (more && pd < need2) {
        pd = sbr_spin_flag(flagp, need2, abort_w);
        if (pd < 0) return;
      }
      // ---- (2) B <- H (B H_prev), D <- H D H in registers, and between the columns the loads of the NEXT task's blocks (rows 0..62
      // valid once the predecessor's task k+1 has drained; row 63: message): a column's register is free once its update is formed,
      // and the ~13 clocks the CU's address unit takes per load pass under the update arithmetic instead of after it
      const int64_t rk1 = rk + SB;
      const int L1 = (int)((n - rk1 < SB) ? n - rk1 : SB);
      const unsigned cB1 = (unsigned)rk, cD1 = (unsigned)rk1;
      const unsigned lb = (more && lane < L1) ? ((cB1 + w16) * LDB2 + SB + lane - w16) * 4u : OORB;
      const unsigned ld = (more && lane < L1) ? ((cD1 + w16) * LDB2 + lane - w16) * 4u : OORB;  // above the diagonal: not used
      float bnv[16], dnv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 16 * wq + q;
        float bn = fmaf(-(tau * v), sbr_rl(z, j), fmaf(-(tp * w), sbr_rl(vp, j), rb[q]));
        if (j == 0) bn = (lane == 0) ? beta : 0.f;
        bnv[q] = bn;
        dnv[q] = fmaf(-w2, sbr_rl(v, j), fmaf(-v, sbr_rl(w2, j), dd[q]));
        rb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lb, q * QS, 16));
        rd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ld, q * QS, 16));
        __builtin_amdgcn_sched_barrier(0);  // keep the two loads of column q behind its arithmetic (the scheduler would issue all 32 first)
      }
      {  // a first look at the next task's message
        const bool msg_next = more && has_prev && k + 2 < Kprev;
        prn = __builtin_amdgcn_raw_buffer_load_b64(rm, (msg_next && act) ? ((mb_in + (unsigned)(k + 2)) * MBW + midx) * 8u : OOR, 0, 16);
      }
      SBR_PROF_MARK(4)  // counter check + update arithmetic with the prefetch loads between
      // ---- (3) the rest of the blocks, write-through
      const bool kb = k > 0;
      const unsigned sb_ = (kb && lane >= lo && lane < L) ? ((colB0 + w16) * LDB2 + SB + lane - w16) * 4u : OORB;
      const unsigned sd_ = (lane >= lo && lane < L) ? ((colD0 + w16) * LDB2 + lane - w16) * 4u : OORB;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bnv[q]), rs, sb_, q * QS, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dnv[q]), rs, (dl >= q) ? sd_ : OORB, q * QS, 16);
      }
      if (wq == 0) {
        if (k == 0) {
          const unsigned oy = (lane < L) ? ((unsigned)s * LDB2 + 1 + lane) * 4u : OOR;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (lane == 0) ? beta : 0.f), rs, oy, 0, 16);
        }
        if (lane < L) a.V2[s * a.ldv2 + rk + lane] = v;
        if (lane == 0) a.TAU2[s * a.ldt + k] = tau;
      }
      vp = v;
      tp = tau;
      SBR_PROF_MARK(5)  // bulk stores issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores have left (and the prefetch has landed)
      SBR_PROF_MARK(6)  // drain + prefetch landed
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.done + s, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SBR_PROF_MARK(7)  // barrier 3 + counter store
    }
  }
  if (PROF && tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(a.prof + i, pacc[i]);
    atomicAdd(a.prof + 8, ptasks);
  }
#undef SBR_PROF_MARK
}

__global__ void sbr_band_diag(const float* __restrict__ Bd, int64_t n, double* __restrict__ d, double* __restrict__ e) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    d[i] = (double)Bd[i * LDB2];
    e[i] = (i + 1 < n) ? (double)Bd[i * LDB2 + 1] : 0.0;
  }
}


// A: the output of sy2sb_f32 (lower band valid). d, e (fp64, device) receive the tridiagonal matrix.
int sb2st_f32(Ctx* ctx, const float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev) {
  if (n % SB != 0 || n < SB) return ctx->fail(SCLENS_ERR_ARG, "sb2st_f32: the order must be a positive multiple of 64");
  StageTimer tm(ctx, "sb2st");
  if (ctx->q2_ev && ctx->q2_tg_n >= 0) SCL_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->q2_ev, 0));  // T factors of the previous reflectors
  ctx->q2_tg_n = -1;
  const int64_t ldv2 = sbr_ldv2(n), ldt = n / SB + 2;
  SCL_WS(ctx, Bd, float, "sbr.Bd", n * LDB2);
  SCL_WS(ctx, V2, float, "sbr.V2", (n + 64) * ldv2);  // spare rows: the back-transformation reads whole 64-float runs
  SCL_WS(ctx, TAU2, float, "sbr.TAU2", n * ldt);
  SCL_WS(ctx, done, unsigned, "sbr.done", n + 4);  // progress counters + abort word
  hipStream_t st = ctx->stream;
  SCL_HIP(ctx, hipMemsetAsync(done, 0, sizeof(unsigned) * (n + 4), st));
  SCL_HIP(ctx, hipMemsetAsync(TAU2, 0, sizeof(float) * n * ldt, st));
  SCL_HIP(ctx, hipMemsetAsync(V2, 0, sizeof(float) * (n + 64) * ldv2, st));  // entries no reflector owns must read as zero
  hipLaunchKernelGGL(sbr_pack_band, dim3((unsigned)n), dim3(128), 0, st, A, n, lda, Bd);
  // every workgroup must be resident (a sweep spins on its predecessor): one per CU is always safe
  int dev = 0, cus = 0;
  SCL_HIP(ctx, hipGetDevice(&dev));
  SCL_HIP(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // (three workgroups fit a CU -- 145 VGPRs, 36 KB of LDS -- so the chases of up to three concurrent streams stay co-resident
  // even at one workgroup per sweep in flight; context option chase_wgs lowers it. 128 workgroups: 330 ms, 235: 297 ms at n = 30 016)
  int G = (int)std::min<int64_t>(cus > 0 ? cus : 64, n / (2 * SB) + 1);
  if (G < 1) G = 1;
  if (ctx->opt.chase_wgs > 0) G = std::max(1, std::min(G, (int)ctx->opt.chase_wgs));
  SbrChaseArgs ca{Bd, n, V2, ldv2, TAU2, ldt, done};
  const int use_mb = (int)ctx->opt.chase_mb;
  // (CU-masked streams for this and the other partial-chip stages were measured twice and removed. Round 3: every third CU, all ~235
  // workgroups: not all became resident, the chase ended through its bounded spin. Round 5: a contiguous range of 64 / 96 / 128 CUs of
  // the driver's numbering with the workgroups capped at 9/4 per CU: the chase takes 381 / 308 / 278 ms instead of 254, the second
  // back-transformation 661 / 550 / 352 instead of 267 -- packed 2-3 per CU these stages are NOT idle waiters, they are bound by their
  // CU's LDS / L2 delivery -- and a Gram product on another stream still gained nothing (eigensolve + Gram concurrently = their sum):
  // a whole call 41.8 s instead of 29.0. profiles/r05_pipe_masks_chefsi_split.log.)
  if (use_mb) {
    const int R = G + 1, kmax = (int)(n / SB) + 2;  // slot of sweep s is free again once sweep s+1 has ended: before sweep s+G+1 starts
    SCL_WS(ctx, MB, unsigned long long, "sbr.MB", (int64_t)R * kmax * MBW);
    SCL_HIP(ctx, hipMemsetAsync(MB, 0, sizeof(unsigned long long) * (size_t)R * kmax * MBW, st));  // tag 0 = no sweep
    SbrChaseMbArgs cm{Bd, n, V2, ldv2, TAU2, ldt, done, MB, R, kmax, nullptr};
    if (ctx->opt.chase_prof > 0) {
      SCL_WS(ctx, prof, unsigned long long, "sbr.prof", 16);
      SCL_HIP(ctx, hipMemsetAsync(prof, 0, sizeof(unsigned long long) * 16, st));
      cm.prof = prof;
      hipEvent_t e0, e1;
      SCL_HIP(ctx, hipEventCreate(&e0));
      SCL_HIP(ctx, hipEventCreate(&e1));
      SCL_HIP(ctx, hipEventRecord(e0, st));
      hipLaunchKernelGGL(sbr_chase_mb<true>, dim3(G), dim3(256), 0, st, cm);
      SCL_HIP(ctx, hipEventRecord(e1, st));
      unsigned long long h[9];
      SCL_HIP(ctx, hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, st));
      SCL_HIP(ctx, hipStreamSynchronize(st));
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      hipEventDestroy(e0);
      hipEventDestroy(e1);
      static const char* nm[8] = {"message poll + patch", "LDS images + w + barrier 1", "reflector + z, Dv + barrier 2", "z, w2 + message send",
                                  "counter check + update + prefetch", "bulk stores issue", "drain + prefetch wait", "barrier 3 + counter"};
      unsigned long long tot = 0;
      for (int i = 0; i < 8; ++i) tot += h[i];
      fprintf(stderr, "[sbr_chase_mb] n = %lld, G = %d, %.2f ms, %llu tasks, %.0f clocks per task inside a workgroup\n", (long long)n, G, ms,
              h[8], (double)tot / (double)h[8]);
      for (int i = 0; i < 8; ++i)
        fprintf(stderr, "   %-34s %8.1f clocks per task (%4.1f %%)\n", nm[i], (double)h[i] / (double)h[8], 100.0 * h[i] / tot);
    } else {
      hipLaunchKernelGGL(sbr_chase_mb<false>, dim3(G), dim3(256), 0, st, cm);
    }
  } else {
    hipLaunchKernelGGL(sbr_chase, dim3(G), dim3(256), 0, st, ca);
  }
  hipLaunchKernelGGL(sbr_band_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Bd, n, d_dev, e_dev);
  SCL_HIP(ctx, hipGetLastError());
  unsigned aborted = 0;
  SCL_HIP(ctx, hipMemcpyAsync(&aborted, done + n, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  if (aborted) return ctx->fail(SCLENS_ERR_HIP, "sb2st_f32: a sweep waited too long for its predecessor (bulge chasing aborted)");
  return SCLENS_OK;
}

// ---- the two-stage eigen-solver behind eig_values / eig_vectors (selected by Ctx::two_stage) ----------------------------
// Orders that are not multiples of SB are embedded in a padded copy: [A 0; 0 diag(sentinel)] with the sentinel above
// the Gershgorin bound of A, so that the true eigenvalues are the first n of the padded spectrum; the pad block is exactly
// decoupled (its reflector components stay zero), so it does not touch the accuracy of the rest. A itself is not modified.
__global__ __launch_bounds__(256) void sbr_row_abs_max(const float* __restrict__ A, int64_t n, int64_t lda,
                                                       unsigned* __restrict__ out) {
  __shared__ float sw[4];
  const float* a = A + (int64_t)blockIdx.x * lda;
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < n; c += 256) s += fabsf(a[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint((sw[0] + sw[1]) + (sw[2] + sw[3])));  // non-negative floats order as uints
}
__global__ void sbr_pad_copy(const float* __restrict__ A, int64_t n, int64_t lda, float* __restrict__ Ap, int64_t np,
                             int64_t ldp, const unsigned* __restrict__ bound, int64_t row0) {
  const int64_t r = row0 + blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ldp) return;
  float v = 0.f;
  if (r < n && c < n) v = A[r * lda + c];
  else if (r == c && r < np) v = 2.f * __uint_as_float(*bound) + 1.f + (float)(r - n);  // decoupled sentinels, distinct
  Ap[r * ldp + c] = v;
}
__global__ void sbr_copy_f64(const double* __restrict__ in, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// returns SCLENS_OK with *used = 1, or *used = 0 when the caller must take the one-stage path (order too small, or a
// panel broke down in the Cholesky QR)
int eig_values_two_stage(Ctx* ctx, const float* A, int64_t n, int64_t lda, double* w64_dev, int* used, int64_t n_low) {
  *used = 0;
  const int64_t np = round_up(n, SB);
  if (np < 2 * SB || np > 120000) return SCLENS_OK;  // 120000: 32-bit byte offsets of the packed band in sbr_chase
  const int64_t ldp = np;
  SCL_WS(ctx, Ap, float, "sbr.Ap", np * ldp);
  SCL_WS(ctx, Tall, float, "sbr.Tall", (np / SB) * SB * SB);
  SCL_WS(ctx, d, double, "sbr.d", np);
  SCL_WS(ctx, e, double, "sbr.e", np);
  SCL_WS(ctx, wp, double, "sbr.w", np);
  SCL_WS(ctx, bound, unsigned, "sbr.bound", 4);
  hipStream_t st = ctx->stream;
  // group data of the PREVIOUS decomposition may still be in preparation on the auxiliary stream when its vectors were never asked
  // for: it reads Ap, which the pad copy below overwrites
  if (ctx->q1p_n >= 0 && ctx->q1_ev) SCL_HIP(ctx, hipStreamWaitEvent(st, ctx->q1_ev, 0));
  SCL_HIP(ctx, hipMemsetAsync(bound, 0, sizeof(unsigned) * 4, st));
  hipLaunchKernelGGL(sbr_row_abs_max, dim3((unsigned)n), dim3(256), 0, st, A, n, lda, bound);
  for (int64_t r0 = 0; r0 < np; r0 += 65535) {
    const int64_t rows = (np - r0 < 65535) ? np - r0 : 65535;
    hipLaunchKernelGGL(sbr_pad_copy, dim3((unsigned)((ldp + 255) / 256), (unsigned)rows), dim3(256), 0, st, A, n, lda,
                       Ap, np, ldp, bound, r0);
  }
  int breakdown = 0;
  SCL_TRY(sy2sb_f32(ctx, Ap, np, ldp, Tall, &breakdown));
  if (breakdown) return SCLENS_OK;
  SCL_TRY(sbr_q1_prepare(ctx, Ap, np, ldp, Tall));  // on the auxiliary stream, beside the chase
  SCL_TRY(sb2st_f32(ctx, Ap, np, ldp, d, e));
  if (n_low < 0 || n_low >= n - 1) {
    SCL_TRY(stebz_f64(ctx, d, e, np, wp));
  } else {  // the padded spectrum ends with np - n sentinels: the largest true eigenvalue has index n - 1
    SCL_TRY(stebz_f64(ctx, d, e, np, wp, n_low, n - 1));
  }
  SCL_TRY(sbr_q2_prebuild(ctx, np));
  hipLaunchKernelGGL(sbr_copy_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wp, w64_dev, n);
  SCL_HIP(ctx, hipGetLastError());
  *used = 1;
  return SCLENS_OK;
}

int eig_values_two_stage_redo(Ctx* ctx, int64_t n, double* w64_dev) {
  const int64_t np = round_up(n, SB);
  auto ws = [&](const char* name) -> void* { return ctx->ws.count(name) ? ctx->ws.at(name).first : nullptr; };
  double* d = static_cast<double*>(ws("sbr.d"));
  double* e = static_cast<double*>(ws("sbr.e"));
  double* wp = static_cast<double*>(ws("sbr.w"));
  if (!d || !e || !wp) return ctx->fail(SCLENS_ERR_STATE, "eig_values_two_stage_redo: no preceding eig_values_two_stage");
  SCL_TRY(stebz_f64(ctx, d, e, np, wp));
  hipLaunchKernelGGL(sbr_copy_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, wp, w64_dev, n);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

__global__ void sbr_unpad_rows(const float* __restrict__ Zp, int64_t ldzp, int64_t n, float* __restrict__ Zt, int64_t ldz) {
  const int64_t r = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) Zt[r * ldz + c] = Zp[r * ldzp + c];
}

// eigenvectors vec_lo .. vec_hi-1 (ascending eigen-index) of the matrix of the preceding eig_values_two_stage call
int eig_vectors_two_stage(Ctx* ctx, int64_t n, int64_t vec_lo, int64_t vec_hi, float* Zt, int64_t ldz) {
  const int64_t m = vec_hi - vec_lo;
  if (m <= 0) return SCLENS_OK;
  const int64_t np = round_up(n, SB), ldp = np;
  auto ws = [&](const char* name) -> void* { return ctx->ws.count(name) ? ctx->ws.at(name).first : nullptr; };
  float* Ap = static_cast<float*>(ws("sbr.Ap"));
  float* Tall = static_cast<float*>(ws("sbr.Tall"));
  double* d = static_cast<double*>(ws("sbr.d"));
  double* e = static_cast<double*>(ws("sbr.e"));
  double* wp = static_cast<double*>(ws("sbr.w"));
  if (!Ap || !Tall || !d || !e || !wp) return ctx->fail(SCLENS_ERR_STATE, "eig_vectors_two_stage: no preceding eig_values_two_stage");
  SCL_WS(ctx, Zp, float, "sbr.Zp", m * ldp);
  SCL_TRY(stein_f64(ctx, d, e, np, wp, vec_lo, vec_hi, Zp, ldp));
  SCL_TRY(sbr_apply_q2(ctx, np, Zp, m, ldp));
  SCL_TRY(sbr_apply_q1(ctx, Ap, np, ldp, Tall, Zp, m, ldp));
  for (int64_t r0 = 0; r0 < m; r0 += 65535) {
    const int64_t rows = (m - r0 < 65535) ? m - r0 : 65535;
    hipLaunchKernelGGL(sbr_unpad_rows, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream,
                       Zp + r0 * ldp, ldp, n, Zt + r0 * ldz, ldz);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
