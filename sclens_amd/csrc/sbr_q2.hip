// Second back-transformation of the two-stage symmetric eigensolver (gfx950): eigenvectors of the tridiagonal matrix -> eigenvectors of
// the band matrix, z_B = Q2 z_T with Q2 the product of the bulge chase's reflectors (sb2st_f32 in sbr.hip leaves them in the context's
// workspaces "sbr.V2" / "sbr.TAU2"). Part of what replaces `_get_eigen` (scLENS.jl:375-387 -> cuSOLVER ssyevd / LAPACK dsyevr).
// Kernels: the unblocked reference (tests), the fp32 kernel of round 3 (`precision = 0`), and the image-fed split-fp16 kernel (default).
#include <algorithm>

#include "sbr_common.h"

namespace scl {

static int sbr_q2_launch_build_t(Ctx* ctx, int64_t n, hipStream_t st);

// The T factors of the second back-transformation (33 ms at n = 30 016) depend only on the reflectors of the chase: enqueued on the
// auxiliary stream BEHIND what the main stream holds now (the bisection: it is bound by vector-ALU issue and lost 40 ms with this
// kernel beside it), so that they are built beside the inverse iteration -- a few hundred waves waiting for memory -- instead of in
// front of sbr_apply_q2. profiles/r03_eig_30016_final.log.
int sbr_q2_prebuild(Ctx* ctx, int64_t n) {
  if (ctx->opt.q2_tg_early == 0 || n - 2 <= 0 || !ctx->q2_prebuild) return SCLENS_OK;
  SCL_TRY(sbr_ensure_aux(ctx));
  SCL_HIP(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));
  SCL_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[0], 0));
  SCL_TRY(sbr_q2_launch_build_t(ctx, n, ctx->aux_stream));
  SCL_HIP(ctx, hipEventRecord(ctx->q2_ev, ctx->aux_stream));
  ctx->q2_tg_n = n;
  return SCLENS_OK;
}


// ---- second back-transformation, reference version: rows of Zt (eigenvectors of the tridiagonal matrix) -> eigenvectors
// of the band matrix. z_B = Q2 z_T with Q2 = prod_{s ascending} prod_k H_{s,k}: the reflectors are applied in the reverse
// order of their creation, sweep by sweep (the tasks of one sweep act on disjoint coordinates). One workgroup keeps
// `VT` whole vectors in LDS and streams all reflectors: every workgroup reads all of V2, so this version is only meant
// for tests and small orders; the blocked version (groups of consecutive sweeps as WY blocks) replaces it.
__global__ __launch_bounds__(256) void sbr_q2_simple(const float* __restrict__ V2, int64_t ldv2, const float* __restrict__ TAU2,
                                                     int64_t ldt, int64_t n, float* __restrict__ Zt, int64_t m, int64_t ldz,
                                                     int VT) {
  extern __shared__ float zs[];  // [VT][n]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t v0 = (int64_t)blockIdx.x * VT;
  const int nv = (int)((m - v0 < VT) ? m - v0 : VT);
  for (int q = 0; q < nv; ++q)
    for (int64_t c = tid; c < n; c += 256) zs[(int64_t)q * n + c] = Zt[(v0 + q) * ldz + c];
  __syncthreads();
  for (int64_t s = n - 3; s >= 0; --s) {
    const int K = sbr_tasks_of(s, n);
    for (int k = wv; k < K; k += 4) {
      const int64_t rk = s + 1 + (int64_t)k * SB;
      const int L = (int)((n - rk < SB) ? n - rk : SB);
      const float tau = TAU2[s * ldt + k];
      if (tau == 0.f) continue;  // wave-uniform
      const float vi = (lane < L) ? V2[s * ldv2 + rk + lane] : 0.f;
      for (int q = 0; q < nv; ++q) {
        float* z = zs + (int64_t)q * n + rk;
        const float zi = (lane < L) ? z[lane] : 0.f;
        float dot = vi * zi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        if (lane < L) z[lane] = zi - tau * dot * vi;
      }
    }
    __syncthreads();
  }
  for (int q = 0; q < nv; ++q)
    for (int64_t c = tid; c < n; c += 256) Zt[(v0 + q) * ldz + c] = zs[(int64_t)q * n + c];
}

// ---- second back-transformation, blocked: WY groups on the matrix cores -----------------------------------------------------
// Group (blk, k) = the reflectors (s, k) of the QW consecutive sweeps s = QW blk + c, c < QW: in the window of rows
// R0 = QW blk + 1 + SB k .. R0 + SB + QW - 2 they form a parallelogram Vg (column c occupies the rows c .. c + L_c - 1);
// H_S H_{S+1} ... = I - Vg Tg Vg' with the forward columnwise T factor. Row form: zw <- zw - ((zw Vg) Tg') Vg'.
// Order (derived from which reflectors overlap): sweep blocks from the last to the first, inside a block k ascending.
constexpr int QW = 32;            // sweeps per group
constexpr int QH = SB + QW;       // window height, padded (SB + QW - 1 rows are used)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sbr_vg(const float* __restrict__ V2, int64_t ldv2, int64_t n, int64_t S, int k, int r, int c) {
  // Vg[r][c] of group (S, k): entry of reflector (S + c, k) at window row r
  const int64_t s = S + c;
  if (s + 2 >= n) return 0.f;                       // no such sweep
  const int64_t rk = s + 1 + (int64_t)k * SB;       // first row of the reflector = R0 + c
  if (rk >= n) return 0.f;                          // the sweep has no task k
  const int64_t L = (n - rk < SB) ? n - rk : SB;
  const int rr = r - c;
  return (rr >= 0 && rr < L) ? V2[s * ldv2 + rk + rr] : 0.f;
}

// T factors of all groups: grid (nk, nblk), one wave
__global__ __launch_bounds__(64) void sbr_q2_build_t(const float* __restrict__ V2, int64_t ldv2, const float* __restrict__ TAU2,
                                                     int64_t ldt, int64_t n, int nk, float* __restrict__ Tg) {
  __shared__ float Vg[QH][QW + 1];
  __shared__ float T[QW][QW + 1];
  __shared__ float g[QW];
  const int k = blockIdx.x, blk = blockIdx.y, l = threadIdx.x;
  const int64_t S = (int64_t)blk * QW;
  for (int idx = l; idx < QH * QW; idx += 64) {
    const int r = idx / QW, c = idx % QW;
    Vg[r][c] = sbr_vg(V2, ldv2, n, S, k, r, c);
  }
  if (l < QW)
    for (int c = 0; c < QW; ++c) T[l][c] = 0.f;
  __syncthreads();
  for (int c = 0; c < QW; ++c) {
    const int64_t s = S + c;
    float tau = 0.f;
    if (s + 2 < n && s + 1 + (int64_t)k * SB < n) tau = TAU2[s * ldt + k];
    if (l < c) {  // g_l = Vg[:, l]' Vg[:, c]
      float acc = 0.f;
      for (int r = c; r < QH; ++r) acc += Vg[r][l] * Vg[r][c];
      g[l] = acc;
    }
    __syncthreads();
    if (l < c) {
      float acc = 0.f;
      for (int j = l; j < c; ++j) acc += T[l][j] * g[j];
      T[l][c] = -tau * acc;
    }
    if (l == c) T[c][c] = tau;
    __syncthreads();
  }
  float* out = Tg + ((int64_t)blk * nk + k) * QW * QW;
  for (int idx = l; idx < QW * QW; idx += 64) out[idx] = T[idx / QW][idx % QW];
}

// apply: one wave per tile of 16 vectors, 4 waves per workgroup, `v_mfma_f32_16x16x4_f32`. The window of the vector tile lives in
// REGISTERS in the MFMA result layout (lane = vector + 16 * row quad, register = row inside the quad), which is also the
// B-operand layout of the next product, so a group costs three chained MFMA products and no LDS traffic for Z at all:
//   W' = Vg' Zw'   (32 x 16)      A = Vg' from LDS ([reflector][row], 16-byte reads), B = the window registers
//   U' = Tg W'     (32 x 16)      A = Tg from LDS, B = W' registers
//   Zw' -= Vg U'   (96 x 16)      A = Vg from the same LDS image (4-byte reads), B = -U' registers, C = the window
// Only the 16 x 16 tile pairs that meet the parallelogram are multiplied (40 + 12 + 40 MFMAs per group instead of 48 + 16 + 48).
// QJ consecutive sweep blocks are applied in one pass over Z (wavefront order: k ascending, inside a k the blocks descending;
// groups of different blocks at the same k overlap by 32 rows, groups at different k of that order are disjoint), so Z is
// streamed n / (32 QJ) times instead of n / 32 times: the union window of a step is 96 + 32 (QJ - 1) rows = QNT register
// tiles, 64 rows leave and 64 enter per step. The group data (reflectors + T) is staged through a double-buffered LDS image
// shared by the four waves, fetched one group ahead.
// Z is addressed as Zq[v * ldq + 3 + row]: every window starts at a row = 1 (mod 4), so the 4-row register quads are
// 16-byte aligned in this shifted layout.
constexpr int Q_RS = 100;  // floats per reflector in the LDS image: b128 reads 2-way, b32 reads conflict-free
constexpr int Q_RT = 40;   // floats per row of Tg: conflict-free b128 reads
constexpr int Q_BUF = QW * Q_RS + QW * Q_RT;

struct SbrQ2Args {
  const float* V2;
  int64_t ldv2;
  const float* Tg;
  int nk, nblk;
  int64_t n;
  float* Zq;
  int64_t m, ldq;
  unsigned long long* prof;  // context option q2_prof = 1 (image-fed kernels): per-phase shader clocks of wave 0 of one workgroup, else null
};

struct SbrQ2Fetch {
  float v[8], t[4];
};

// ---- variant 3 (round 3): a second, row-major LDS image of the group's reflectors feeds the third product with 16-byte reads
// (four consecutive reflectors of one row = the four k-steps of one MFMA group: 10 reads instead of 40), its reads are issued
// when the first product's MFMAs have been issued (they land during the T product), and the fetch of the NEXT group's data
// (8 + 4 global loads per thread, unconditional: see sbr_ldv2) sits between the MFMAs of the first product instead of in front
// of the group, where the matrix pipe idles.
constexpr int Q_NS = 36;                                  // floats per window row of the row-major image
constexpr int Q_BUF3 = QW * Q_RS + QW * Q_RT + 96 * Q_NS;  // 7 936 floats per buffer

struct SbrQ2Ptr {           // per-thread fetch state of variant 3
  const float* v;           // V2 + wv (ldv2 + 1) + 1 + lane: reflector c = wv + 4 q of group (b, t) sits at
                            // v + 32 b (ldv2 + 1) + 64 t + 4 q (ldv2 + 1)
  const float* tg;          // Tg + tid
  int64_t vstride;          // ldv2 + 1
};

__device__ __forceinline__ void sbr_q2_fetch16v3(SbrQ2Fetch& f, const SbrQ2Args& a, const SbrQ2Ptr& p, int b, int t) {
  const int bb = b < 0 ? 0 : b;                // b < 0: past the last group; the data is never used
  const int tt = t < a.nk ? t : a.nk - 1;
  const float* src = p.v + ((int64_t)bb * QW) * p.vstride + (int64_t)tt * SB;
#pragma unroll
  for (int q = 0; q < 8; ++q) f.v[q] = src[(int64_t)(4 * q) * p.vstride];
  const float* tg = p.tg + ((int64_t)bb * a.nk + tt) * QW * QW;
#pragma unroll
  for (int q = 0; q < 4; ++q) f.t[q] = tg[256 * q];
}

__device__ __forceinline__ void sbr_q2_stash16v3(const SbrQ2Fetch& f, float* buf, int tid) {
  float* N = buf + QW * Q_RS + QW * Q_RT;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q, c = idx >> 6, rr = idx & 63;
    buf[c * Q_RS + c + rr] = f.v[q];
    N[(c + rr) * Q_NS + c] = f.v[q];
  }
  float* T = buf + QW * Q_RS;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    T[(idx >> 5) * Q_RT + (idx & 31)] = f.t[q];
  }
}

template <bool DO>
__device__ __forceinline__ void sbr_q2_group16v3(f32x4* z, const float* buf, int vi, int g, SbrQ2Fetch& pf, const SbrQ2Args& a,
                                                  const SbrQ2Ptr& p, int nb, int nt) {
  const float* VgT = buf;
  const float* T = buf + QW * Q_RS;
  const float* N = T + QW * Q_RT;
  if (!DO) {  // group outside the matrix: only the fetch of the next one
    sbr_q2_fetch16v3(pf, a, p, nb, nt);
    return;
  }
  f32x4 a0[5], a1[5];
#pragma unroll
  for (int rt = 0; rt < 5; ++rt) {
    a0[rt] = *reinterpret_cast<const f32x4*>(VgT + vi * Q_RS + 16 * rt + 4 * g);
    a1[rt] = *reinterpret_cast<const f32x4*>(VgT + (16 + vi) * Q_RS + 16 * (rt + 1) + 4 * g);
  }
  const f32x4 t00 = *reinterpret_cast<const f32x4*>(T + vi * Q_RT + 4 * g);
  const f32x4 t01 = *reinterpret_cast<const f32x4*>(T + vi * Q_RT + 16 + 4 * g);
  const f32x4 t11 = *reinterpret_cast<const f32x4*>(T + (16 + vi) * Q_RT + 16 + 4 * g);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][e], z[0][e], w0, 0, 0, 0);
    w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0][e], z[1][e], w1, 0, 0, 0);
  }
  // the next group's global loads + their address arithmetic, spread over the MFMAs of this product by the scheduler
  sbr_q2_fetch16v3(pf, a, p, nb, nt);
#pragma unroll
  for (int rt = 1; rt < 5; ++rt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rt][e], z[rt][e], w0, 0, 0, 0);
      w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt][e], z[rt + 1][e], w1, 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // operands of the third product (row-major image): issued now, they land while the T product runs
  f32x4 n0[5], n1[5];
#pragma unroll
  for (int rt = 0; rt < 5; ++rt) {
    n0[rt] = *reinterpret_cast<const f32x4*>(N + (16 * rt + vi) * Q_NS + 4 * g);
    n1[rt] = *reinterpret_cast<const f32x4*>(N + (16 * (rt + 1) + vi) * Q_NS + 16 + 4 * g);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f}, u2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t00[e], w0[e], u0, 0, 0, 0);
    u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t11[e], w1[e], u1, 0, 0, 0);
    u2 = __builtin_amdgcn_mfma_f32_16x16x4f32(t01[e], w1[e], u2, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    u0[e] = -(u0[e] + u2[e]);
    u1[e] = -u1[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int rt = 0; rt < 5; ++rt) z[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(n0[rt][e], u0[e], z[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 5; ++rt) z[rt + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(n1[rt][e], u1[e], z[rt + 1], 0, 0, 0);
  }
}

__device__ __forceinline__ f32x4 sbr_q2_ldz(const float* zrow, int64_t row, int64_t n, bool live) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (live && row >= -3 && row < n) {
    if (row + 3 < n) {
      v = *reinterpret_cast<const f32x4*>(zrow + row);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row + e < n) v[e] = zrow[row + e];
    }
  }
  return v;
}
__device__ __forceinline__ void sbr_q2_stz(float* zrow, int64_t row, int64_t n, bool live, f32x4 v) {
  if (live && row >= -3 && row < n) {
    if (row + 3 < n) {
      *reinterpret_cast<f32x4*>(zrow + row) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row + e < n) zrow[row + e] = v[e];
    }
  }
}

template <int QJ, int QNT>
__global__ __launch_bounds__(256, 1) void sbr_q2_apply16v3(SbrQ2Args a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Q_BUF3];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, vi = lane & 15, g = lane >> 4;
  const int64_t v = (int64_t)blockIdx.x * 64 + wv * 16 + vi;
  const bool live = v < a.m;
  float* zrow = a.Zq + (live ? v : 0) * a.ldq + 3;
  for (int i = tid; i < 2 * Q_BUF3; i += 256) lds[i] = 0.f;  // outside the parallelogram the images stay zero
  __syncthreads();
  const int nsb = (a.nblk + QJ - 1) / QJ;
  SbrQ2Ptr p;
  p.vstride = a.ldv2 + 1;
  p.v = a.V2 + (int64_t)wv * p.vstride + 1 + lane;
  p.tg = a.Tg + tid;
  SbrQ2Fetch pf;
  sbr_q2_fetch16v3(pf, a, p, a.nblk - 1, 0);
  sbr_q2_stash16v3(pf, lds, tid);
  __syncthreads();
  int cur = 0;
  f32x4 z[QNT];
  for (int sb = 0; sb < nsb; ++sb) {
    const int bh = a.nblk - 1 - sb * QJ, blow = bh - QJ + 1;
    const int Kmax = sbr_tasks_of((int64_t)(blow > 0 ? blow : 0) * QW, a.n);
    const int64_t base0 = (int64_t)blow * QW + 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores of the previous pass have left before rows are re-read
#pragma unroll
    for (int i = 0; i < QNT; ++i) z[i] = sbr_q2_ldz(zrow, base0 + 16 * i + 4 * g, a.n, live);
    for (int t = 0; t < Kmax; ++t) {
      const int64_t base = base0 + (int64_t)t * SB;
      f32x4 pz[4];
      const bool more = t + 1 < Kmax;
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pz[i] = sbr_q2_ldz(zrow, base + 16 * (QNT + i) + 4 * g, a.n, live);
      }
#pragma unroll
      for (int j = 0; j < QJ; ++j) {
        int nb, nt;  // the group after this one in the sequence
        if (j + 1 < QJ) {
          nb = bh - (j + 1);
          nt = t;
        } else if (more) {
          nb = bh;
          nt = t + 1;
        } else {
          nb = bh - QJ;
          nt = 0;
        }
        const int b = bh - j;
        if (b >= 0 && t < sbr_tasks_of((int64_t)b * QW, a.n))
          sbr_q2_group16v3<true>(z + 2 * (QJ - 1 - j), lds + cur * Q_BUF3, vi, g, pf, a, p, nb, nt);
        else
          sbr_q2_group16v3<false>(z + 2 * (QJ - 1 - j), lds + cur * Q_BUF3, vi, g, pf, a, p, nb, nt);
        sbr_q2_stash16v3(pf, lds + (cur ^ 1) * Q_BUF3, tid);
        __syncthreads();
        cur ^= 1;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) sbr_q2_stz(zrow, base + 16 * i + 4 * g, a.n, live, z[i]);
      if (more) {
#pragma unroll
        for (int i = 0; i + 4 < QNT; ++i) z[i] = z[i + 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[QNT - 4 + i] = pz[i];
      } else {
#pragma unroll
        for (int i = 4; i < QNT; ++i) sbr_q2_stz(zrow, base + 16 * i + 4 * g, a.n, live, z[i]);
      }
    }
  }
}

// index of group (b, t) among the groups that exist (t < tasks of sweep 32 b), for n a multiple of 64: blocks 2c and 2c + 1 have
// n / 64 - c tasks each
__host__ __device__ __forceinline__ int64_t sbr_q2_img_index(int b, int t, int64_t n) {
  const int64_t q = n / SB, c = b >> 1;
  return 2 * c * q - c * (c - 1) + ((b & 1) ? (q - c) : 0) + t;
}
static inline int64_t sbr_q2_img_count(int64_t n) { const int64_t q = n / SB; return q * q + q; }

// ---- image-fed apply kernel (variants 14 / 15 / 16, rounds 4-5): the group data as a PRE-BUILT LDS image, moved global -> LDS by DMA.
// What bound the kernels that staged the reflectors themselves (round 3) was not the matrix pipe but ~430 vector instructions around
// the 69 matrix instructions of a group: every workgroup fetched the group's 32 reflectors + T as floats, split them into fp16 pieces
// and wrote LDS with two-byte stores -- the same work in all m / 64 workgroups. Here one kernel (sbr_q2_build_img, beside the inverse
// iteration on the auxiliary stream) writes the finished image of every group (block b, task t) once, 16 384 bytes (3.6 GB at
// n = 30 016): ONE copy of the reflectors as fp16 pieces + the T factor; gfx950's transposing LDS read (`ds_read_b64_tr_b16`) hands
// the SAME Vg' image to the third product as its row operand: Zw' <- Zw' + Vg (-Tg (Vg' Zw')), three products on
// `v_mfma_f32_16x16x32_f16`, the middle one 32 x 32 x 16 per wave. (Round 4 also measured 28 KB images with T folded into a second
// operand copy, K = 16 matrix instructions and loader waves: profiles/r04_q2_variants.log; removed in round 5.)
//   part A: four planes (s, hl) of 3 072 bytes, s = which 16-row half of a 32-row K step, hl = hi / lo piece; inside a plane
//           [reflector tile ct 2][K step p 3][512 bytes]; the 8-byte cell of (m = reflector in the tile, p' = 0..3) holds the four
//           window rows 32 p + 16 s + 4 p' + e at cell index ((m ^ 8 (p' & 1)) & 15) + 16 ((p' >> 1) ^ (m >> 3)) + 32 (m >> 3):
//           conflict-free for the row reads of the first product, whether they are issued as ds_read_b64 (64 banks, lanes m = 0..15
//           x p' in {0, 1} or {2, 3} per 32-lane half) or paired by the compiler into ds_read2st64_b64 (32 banks, 16 lanes m = 0..15 of
//           one p'), and for the transposed reads of the third (64 banks, lanes m = 0..7 or 8..15 x p' = 0..3 per half)
//   T part: -Tg as the row operand of the middle product, [hl 2][tile ct' 2][g 4][m 16] units of 16 bytes = the eight k slots
//           {c = 4 g + e} and {c = 16 + 4 g + e} of row c' = 16 ct' + m
constexpr int Q_IMG2 = 4096, Q2_PLANE = 768, Q2_TOFF = 4 * Q2_PLANE;
__host__ __device__ __forceinline__ int sbr_q2_cell(int m, int pq) {  // float offset of the cell inside its 512-byte block
  return 2 * (((m ^ ((pq & 1) << 3)) & 15) + 16 * (((pq >> 1) ^ (m >> 3)) & 1) + 32 * (m >> 3));
}

__global__ __launch_bounds__(256) void sbr_q2_build_img(const float* __restrict__ V2, int64_t ldv2, const float* __restrict__ TAU2,
                                                        int64_t ldt, int64_t n, float* __restrict__ img) {
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (t >= sbr_tasks_of((int64_t)b * QW, n)) return;
  __shared__ float Vg[QH][QW + 1];
  __shared__ float G[QW][QW + 1];
  __shared__ float T[QW][QW + 1];
  __shared__ float tau[QW];
  const int64_t S = (int64_t)b * QW;
  for (int idx = tid; idx < QH * QW; idx += 256) {
    const int c = idx / QH, r = idx % QH;  // consecutive threads read consecutive entries of one reflector
    Vg[r][c] = sbr_vg(V2, ldv2, n, S, t, r, c);
  }
  for (int idx = tid; idx < QW * QW; idx += 256) T[idx >> 5][idx & 31] = 0.f;
  if (tid < QW) {
    const int64_t s = S + tid;
    tau[tid] = (s + 2 < n && s + 1 + (int64_t)t * SB < n) ? TAU2[s * ldt + t] : 0.f;
  }
  __syncthreads();
  {  // G = Vg' Vg
    const int i = tid >> 3, j0 = 4 * (tid & 7);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < QH; ++r) {
      const double x = (double)Vg[r][i];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += x * (double)Vg[r][j0 + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) G[i][j0 + e] = (float)acc[e];
  }
  __syncthreads();
  // forward columnwise T factor: T[0:c, c] = -tau_c T[0:c, 0:c] (Vg[:, 0:c]' v_c), T[c][c] = tau_c (the recurrence of sbr_q2_build_t)
  for (int c = 0; c < QW; ++c) {
    if (tid < c) {
      double acc = 0.0;
      for (int j = tid; j < c; ++j) acc += (double)T[tid][j] * (double)G[j][c];
      T[tid][c] = (float)(-(double)tau[c] * acc);
    }
    if (tid == c) T[c][c] = tau[c];
    __syncthreads();
  }
  {
    float* out2 = img + sbr_q2_img_index(b, t, n) * Q_IMG2;
    for (int it = tid; it < 2 * 2 * 3 * 16 * 4; it += 256) {
      const int pq = it & 3, m = (it >> 2) & 15, blk = it >> 6, p = blk % 3, ct = (blk / 3) & 1, sh = blk / 6;
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = Vg[32 * p + 16 * sh + 4 * pq + e][16 * ct + m];
      const SbrHL o = sbr_split_pk(x);
      const int off = (ct * 3 + p) * 128 + sbr_q2_cell(m, pq);
      f32x2 rh, rl;
      __builtin_memcpy(&rh, &o.h, 8);
      __builtin_memcpy(&rl, &o.l, 8);
      *reinterpret_cast<f32x2*>(out2 + (2 * sh) * Q2_PLANE + off) = rh;
      *reinterpret_cast<f32x2*>(out2 + (2 * sh + 1) * Q2_PLANE + off) = rl;
    }
    if (tid < 128) {
      const int m = tid & 15, gq = (tid >> 4) & 3, ct = tid >> 6;
      f32x4 x0, x1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = -T[16 * ct + m][4 * gq + e];
        x1[e] = -T[16 * ct + m][16 + 4 * gq + e];
      }
      const SbrHL8 o = sbr_cat(sbr_split_pk(x0), sbr_split_pk(x1));
      f32x4 rh, rl;
      __builtin_memcpy(&rh, &o.h, 16);
      __builtin_memcpy(&rl, &o.l, 16);
      *reinterpret_cast<f32x4*>(out2 + Q2_TOFF + ((0 * 2 + ct) * 4 + gq) * 64 + 4 * m) = rh;
      *reinterpret_cast<f32x4*>(out2 + Q2_TOFF + ((1 * 2 + ct) * 4 + gq) * 64 + 4 * m) = rl;
    }
  }
}

// the NP DMA instructions of one image: lane l of wave w moves bytes [(q 256 + 64 w + l) 16, +16) of the image for q < NP
template <int NP>
__device__ __forceinline__ void sbr_q2_dma(const float* __restrict__ img, int64_t index, float* buf, int tid) {
  const float* src = img + index * (NP * 1024) + 4 * tid;
  float* dst = buf + 256 * (tid >> 6);
#pragma unroll
  for (int q = 0; q < NP; ++q)
    __builtin_amdgcn_global_load_lds((glb_void*)(src + 1024 * q), (lds_void*)(dst + 1024 * q), 16, 0, 0);
}

__device__ __forceinline__ f32x4 sbr_mfma3_k32(const SbrHL8& a, const SbrHL8& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.l, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b.h, c, 0, 0, 0);
  return c;
}
constexpr float Q_ZSCALE = 256.f;  // the vector window lives scaled by 2^8 (exact) inside the image-fed kernel

// the group of variants 14 / 15 / 16: three products from the 16 KB image (layout at Q_IMG2)
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t lds_fp16x4;
__device__ __forceinline__ f16x4 sbr_ld_tr(const float* p) {  // transposing read: EXEC must be all ones (it is: whole-wave code)
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)p));
}
__device__ __forceinline__ f16x4 sbr_ld_h4(const float* p) {
  const f32x2 r = *reinterpret_cast<const f32x2*>(p);
  return __builtin_bit_cast(f16x4, r);
}
__device__ __forceinline__ f16x8 sbr_cat4(f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void sbr_q2_group16t(f32x4* z, const float* buf, int vi, int g) {
  // With one wave per SIMD (m = n / 2: 938 wave tiles for 1 024 SIMDs) nothing hides a wave's own latencies, and left to itself the
  // compiler puts every LDS read next to its use (~20 exposed round trips per group, profiles/r04_q2_phase_clocks.log); a scheduling
  // barrier does not stop it, a memory-clobbering statement makes it wait for the reads at once, and it guards the transposing read
  // (a builtin) with `s_waitcnt vmcnt(0)`, i.e. with the DMA of the images still in flight. So the LDS reads of this function are
  // volatile statements (kept in program order), each stage's reads are all in flight before the arithmetic that hides them, and the
  // waits are written out: one `lgkmcnt(0)` per stage, tied to the registers it releases.
  const unsigned lb = (unsigned)(__UINTPTR_TYPE__)(lds_void*)buf;
  // (1) operands of W' = Vg' Zw' (rows = reflectors, two tiles; K = the 96 window rows in three steps) and of U = -Tg W'
  const unsigned aa = lb + 4u * (unsigned)sbr_q2_cell(vi, g);
  f32x2 ah[2][3][2], al[2][3][2];  // [reflector tile][K step][16-row half]
  asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(ah[0][0][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(al[0][0][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(ah[0][0][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(al[0][0][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(ah[0][1][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:3584" : "=v"(al[0][1][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:6656" : "=v"(ah[0][1][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:9728" : "=v"(al[0][1][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(ah[0][2][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(al[0][2][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:7168" : "=v"(ah[0][2][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:10240" : "=v"(al[0][2][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(ah[1][0][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:4608" : "=v"(al[1][0][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:7680" : "=v"(ah[1][0][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:10752" : "=v"(al[1][0][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(ah[1][1][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:5120" : "=v"(al[1][1][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:8192" : "=v"(ah[1][1][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:11264" : "=v"(al[1][1][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:2560" : "=v"(ah[1][2][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:5632" : "=v"(al[1][2][0]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:8704" : "=v"(ah[1][2][1]) : "v"(aa));
  asm volatile("ds_read_b64 %0, %1 offset:11776" : "=v"(al[1][2][1]) : "v"(aa));
  const unsigned ta = lb + 4u * (unsigned)(Q2_TOFF + g * 64 + 4 * vi);
  f32x4 th[2], tl[2];
  asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(th[0]) : "v"(ta));
  asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(tl[0]) : "v"(ta));
  asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(th[1]) : "v"(ta));
  asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(tl[1]) : "v"(ta));
  // (2) the window in fp16 pieces: vector instructions under the reads' latency (the window passes through a volatile statement
  //     behind the reads, so that its splits cannot be scheduled in front of them)
  asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]));
  SbrHL8 zz[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) zz[p] = sbr_cat(sbr_split_pk(z[2 * p]), sbr_split_pk(z[2 * p + 1]));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(ah[0][0][0]), "+v"(ah[0][0][1]), "+v"(ah[0][1][0]), "+v"(ah[0][1][1]), "+v"(ah[0][2][0]), "+v"(ah[0][2][1]),
                 "+v"(ah[1][0][0]), "+v"(ah[1][0][1]), "+v"(ah[1][1][0]), "+v"(ah[1][1][1]), "+v"(ah[1][2][0]), "+v"(ah[1][2][1]),
                 "+v"(al[0][0][0]), "+v"(al[0][0][1]), "+v"(al[0][1][0]), "+v"(al[0][1][1]), "+v"(al[0][2][0]), "+v"(al[0][2][1]),
                 "+v"(al[1][0][0]), "+v"(al[1][0][1]), "+v"(al[1][1][0]), "+v"(al[1][1][1]), "+v"(al[1][2][0]), "+v"(al[1][2][1]),
                 "+v"(th[0]), "+v"(th[1]), "+v"(tl[0]), "+v"(tl[1]));
  // (3) W'
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    SbrHL8 a0, a1;
    a0.h = sbr_cat4(__builtin_bit_cast(f16x4, ah[0][p][0]), __builtin_bit_cast(f16x4, ah[0][p][1]));
    a0.l = sbr_cat4(__builtin_bit_cast(f16x4, al[0][p][0]), __builtin_bit_cast(f16x4, al[0][p][1]));
    a1.h = sbr_cat4(__builtin_bit_cast(f16x4, ah[1][p][0]), __builtin_bit_cast(f16x4, ah[1][p][1]));
    a1.l = sbr_cat4(__builtin_bit_cast(f16x4, al[1][p][0]), __builtin_bit_cast(f16x4, al[1][p][1]));
    w0 = sbr_mfma3_k32(a0, zz[p], w0);
    w1 = sbr_mfma3_k32(a1, zz[p], w1);
  }
  // (4) Vg for the third product out of the same image by transposing reads, all 24 in flight behind the matrix instructions of (3)
  //     (the address passes through a statement that reads W'): lane 4 q + p'' of a 16-lane group addresses the cell of reflector
  //     4 g + q (then 16 + 4 g + q), rows 4 p'' .. 4 p'' + 3 of the tile
  unsigned ya = lb + 4u * (unsigned)sbr_q2_cell(4 * g + (vi >> 2), vi & 3);
  asm volatile("" : "+v"(ya), "+v"(w0), "+v"(w1));
  f32x2 yh[6][2], yl[6][2];  // [window-row tile][reflector tile]
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(yh[0][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(yl[0][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1536" : "=v"(yh[0][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4608" : "=v"(yl[0][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(yh[1][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(yl[1][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:7680" : "=v"(yh[1][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:10752" : "=v"(yl[1][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(yh[2][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3584" : "=v"(yl[2][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(yh[2][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:5120" : "=v"(yl[2][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6656" : "=v"(yh[3][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9728" : "=v"(yl[3][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(yh[3][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:11264" : "=v"(yl[3][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(yh[4][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(yl[4][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2560" : "=v"(yh[4][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:5632" : "=v"(yl[4][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:7168" : "=v"(yh[5][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:10240" : "=v"(yl[5][0]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8704" : "=v"(yh[5][1]) : "v"(ya));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:11776" : "=v"(yl[5][1]) : "v"(ya));
  // (5) U = -Tg W'
  const SbrHL8 ws = sbr_cat(sbr_split_pk(w0), sbr_split_pk(w1));
  f16x8 t0h, t1h, t0l, t1l;
  __builtin_memcpy(&t0h, &th[0], 16);
  __builtin_memcpy(&t1h, &th[1], 16);
  __builtin_memcpy(&t0l, &tl[0], 16);
  __builtin_memcpy(&t1l, &tl[1], 16);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 u0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t0h, ws.h, zero, 0, 0, 0);
  f32x4 u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1h, ws.h, zero, 0, 0, 0);
  u0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t0h, ws.l, u0, 0, 0, 0);
  u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1h, ws.l, u1, 0, 0, 0);
  u0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t0l, ws.h, u0, 0, 0, 0);
  u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1l, ws.h, u1, 0, 0, 0);
  const SbrHL8 us = sbr_cat(sbr_split_pk(u0), sbr_split_pk(u1));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(yh[0][0]), "+v"(yh[0][1]), "+v"(yh[1][0]), "+v"(yh[1][1]), "+v"(yh[2][0]), "+v"(yh[2][1]), "+v"(yh[3][0]), "+v"(yh[3][1]),
                 "+v"(yh[4][0]), "+v"(yh[4][1]), "+v"(yh[5][0]), "+v"(yh[5][1]), "+v"(yl[0][0]), "+v"(yl[0][1]), "+v"(yl[1][0]), "+v"(yl[1][1]),
                 "+v"(yl[2][0]), "+v"(yl[2][1]), "+v"(yl[3][0]), "+v"(yl[3][1]), "+v"(yl[4][0]), "+v"(yl[4][1]), "+v"(yl[5][0]), "+v"(yl[5][1]));
  // (6) Zw' += Vg U: rows = window rows (six tiles), K = the 32 reflectors
#pragma unroll
  for (int rt = 0; rt < 6; ++rt) {
    SbrHL8 y;
    y.h = sbr_cat4(__builtin_bit_cast(f16x4, yh[rt][0]), __builtin_bit_cast(f16x4, yh[rt][1]));
    y.l = sbr_cat4(__builtin_bit_cast(f16x4, yl[rt][0]), __builtin_bit_cast(f16x4, yl[rt][1]));
    z[rt] = sbr_mfma3_k32(y, us, z[rt]);
  }
}

// Window loads / stores of the image-fed kernel: buffer instructions on a resource that covers the workgroup's 64 vectors, one
// instruction per lane and call WHATEVER the row (quads outside [-3, n - 3] and vectors past m get an offset beyond the resource:
// the load returns 0, the store is dropped) -- the number of memory instructions between two waits is then a constant, which the
// counted `s_waitcnt vmcnt(N)` below rely on. A quad that starts at row n - 3 ends in the row's padding (ldq >= n + 4 floats, zeroed
// by sbr_q2_shift).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct SbrZWin {
  __amdgpu_buffer_rsrc_t rs;
  int64_t lane_base;  // float index of row 0 of this lane's vector inside the resource
  int64_t n;
};
__device__ __forceinline__ unsigned sbr_zoff(const SbrZWin& w, int64_t row) {
  return (row >= -3 && row <= w.n - 3) ? (unsigned)((w.lane_base + row) * 4) : 0xfffffff0u;
}
__device__ __forceinline__ f32x4 sbr_zld(const SbrZWin& w, int64_t row) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, sbr_zoff(w, row), 0, 0));
}
__device__ __forceinline__ void sbr_zst(const SbrZWin& w, int64_t row, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), w.rs, sbr_zoff(w, row), 0, 0);
}
template <int N>
__device__ __forceinline__ void sbr_vmcnt() {  // at most N vector-memory instructions of this wave may still be in flight
  static_assert(N == 0 || N == 4 || N == 8 || N == 12, "counts of the image-fed kernel");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// The image-fed apply kernel: one wave per 16 vectors, four waves per workgroup, the window of QNT row tiles in registers (scaled by
// Q_ZSCALE so that the low fp16 pieces of entries of size 1 / sqrt(n) stay normal), NBUF image buffers in LDS (NBUF - 1 groups ahead).
template <int QJ, int QNT, int NBUF>
__global__ __launch_bounds__(256, 2) void sbr_q2_apply16e(SbrQ2Args a, const float* __restrict__ img) {
  static_assert(NBUF == 2 || NBUF == 3, "one or two groups ahead");
  constexpr int NP = 4, QI = NP * 1024;                  // DMA pieces (4 KB) and floats per image
  constexpr int AH = NBUF - 1, DM = (AH == 2) ? NP : 0;  // DMA instructions that may stay in flight past the end of a group
  extern __shared__ __attribute__((aligned(16))) float q2lds[];  // NBUF images
  float* lds = q2lds;
  const int tid = threadIdx.x, lane = tid & 63, vi = lane & 15, g = lane >> 4, wv = tid >> 6;
  const int64_t v0 = (int64_t)blockIdx.x * 64;
  const int64_t nvec = (a.m - v0 < 64) ? a.m - v0 : 64;
  SbrZWin zw;
  zw.rs = __builtin_amdgcn_make_buffer_rsrc(a.Zq + v0 * a.ldq, 0, (unsigned)(nvec * a.ldq * 4), 0x00020000);
  zw.lane_base = (int64_t)(wv * 16 + vi) * a.ldq + 3;
  zw.n = a.n;
  const int nsb = (a.nblk + QJ - 1) / QJ;
  auto index_of = [&](int b, int t) -> int64_t {  // groups that do not exist read image 0 (never used)
    return (b >= 0 && t < sbr_tasks_of((int64_t)b * QW, a.n)) ? sbr_q2_img_index(b, t, a.n) : 0;
  };
  // the first AH groups of the sequence: blocks nblk - 1, nblk - 2 at task 0 (QJ > AH)
#pragma unroll
  for (int i = 0; i < AH; ++i) sbr_q2_dma<NP>(img, index_of(a.nblk - 1 - i, 0), lds + i * QI, tid);
  sbr_vmcnt<DM>();
  __syncthreads();
  int cur = 0;
  f32x4 z[QNT];
  // phase clocks (a.prof): 0 DMA issue, 1 window loads / stores issue, 2 products, 3 counted wait, 4 barrier, 5 between groups
  const bool prof = a.prof != nullptr && blockIdx.x == gridDim.x / 2 && tid < 64;
  unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0}, pn = 0, pt = prof ? __builtin_amdgcn_s_memtime() : 0ull;
#define SBR_Q2_STAMP(i)                                               \
  if (prof) {                                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
    pacc[i] += now_ - pt;                                             \
    pt = now_;                                                        \
  }
  for (int sb = 0; sb < nsb; ++sb) {
    const int bh = a.nblk - 1 - sb * QJ, blow = bh - QJ + 1;
    const int Kmax = sbr_tasks_of((int64_t)(blow > 0 ? blow : 0) * QW, a.n);
    const int64_t base0 = (int64_t)blow * QW + 1;
    sbr_vmcnt<0>();  // the stores of the previous pass have left before rows are re-read
#pragma unroll
    for (int i = 0; i < QNT; ++i) z[i] = sbr_zld(zw, base0 + 16 * i + 4 * g) * Q_ZSCALE;
    f32x4 zout[4];       // the 64 rows that left the window at the end of the previous task: stored inside the next task's first
    bool pend = false;   // group, BEHIND its DMA instructions (memory instructions complete in issue order: a wait for the image
                         // would otherwise also wait one HBM round trip for stores and loads nobody needs yet -- 0.5 us per group,
                         // 111 of the kernel's 295 ms, profiles/r04_q2_variants.log)
    for (int t = 0; t < Kmax; ++t) {
      const int64_t base = base0 + (int64_t)t * SB;
      f32x4 pz[4];
      const bool more = t + 1 < Kmax;
#pragma unroll
      for (int j = 0; j < QJ; ++j) {
        int nb, nt;  // the group AH steps after this one in the sequence (j ascending inside a task, then the next task, then the next pass)
        const int jj = j + AH;
        if (jj < QJ) {
          nb = bh - jj;
          nt = t;
        } else if (more) {
          nb = bh - (jj - QJ);
          nt = t + 1;
        } else {
          nb = bh - QJ - (jj - QJ);
          nt = 0;
        }
        // its image goes to the buffer the PREVIOUS group was read from (all waves have passed the barrier behind it)
        asm volatile("" ::: "memory");
        SBR_Q2_STAMP(5)
        int nxt = cur + AH;
        if (nxt >= NBUF) nxt -= NBUF;
        sbr_q2_dma<NP>(img, index_of(nb, nt), lds + nxt * QI, tid);
        asm volatile("" ::: "memory");
        SBR_Q2_STAMP(0)
        // the window traffic of this task, behind the DMA: 4 stores (rows that left), 4 loads (rows that will enter). Right behind the
        // DMA pieces their issue costs ~550 clocks per group (they queue up behind the workgroup's 16 pieces); issued behind the
        // products instead they issue at once but land later: 265 against 250 ms (profiles/r04_q2_final_variants.log) -- the early
        // position stays.
        if (j == 0) {
          if (pend) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sbr_zst(zw, base - SB + 16 * i + 4 * g, zout[i]);
          }
          if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pz[i] = sbr_zld(zw, base + 16 * (QNT + i) + 4 * g);
          }
          asm volatile("" ::: "memory");
        }
        SBR_Q2_STAMP(1)
        const int b = bh - j;
        if (b >= 0 && t < sbr_tasks_of((int64_t)b * QW, a.n)) sbr_q2_group16t(z + 2 * (QJ - 1 - j), lds + cur * QI, vi, g);
        if (prof) {  // the products' results are in registers when the stamp is taken
#pragma unroll
          for (int i = 0; i < QNT; ++i) asm volatile("" ::"v"(z[i]));
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        }
        SBR_Q2_STAMP(2)
        // the image of the next group must have landed (AH == 1), or the one after it may still be in flight (AH == 2); the window
        // traffic issued in this group (j == 0) may stay in flight as well: it is waited for one group later (j == 1: everything).
        // (Two groups ahead: the image needed next was requested BEFORE this task's window traffic, so that may stay in flight for
        // one more group.)
        if (j == 0 || (j == 1 && AH == 2)) {
          const int nz = (pend ? 4 : 0) + (more ? 4 : 0);
          if (nz == 8) sbr_vmcnt<DM + 8>();
          else if (nz == 4) sbr_vmcnt<DM + 4>();
          else sbr_vmcnt<DM>();
        } else {
          sbr_vmcnt<DM>();
        }
        SBR_Q2_STAMP(3)
        // a bare barrier: __syncthreads() carries a workgroup-scope release fence, which the compiler implements as `s_waitcnt vmcnt(0)`
        // -- every group then waited for ALL of the wave's memory instructions (the image two groups ahead, the window loads and
        // stores), and the counted waits above were void (found in the ISA at the end of round 4: why two groups ahead never paid).
        // What the barrier has to guarantee here is covered by the counted wait of every wave for its own pieces of the image.
        __builtin_amdgcn_s_barrier();
        SBR_Q2_STAMP(4)
        ++pn;
        cur = (cur + 1 == NBUF) ? 0 : cur + 1;
      }
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) zout[i] = z[i] * (1.f / Q_ZSCALE);
        pend = true;
#pragma unroll
        for (int i = 0; i + 4 < QNT; ++i) z[i] = z[i + 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[QNT - 4 + i] = pz[i] * Q_ZSCALE;
      } else {
#pragma unroll
        for (int i = 0; i < QNT; ++i) sbr_zst(zw, base + 16 * i + 4 * g, z[i] * (1.f / Q_ZSCALE));
        pend = false;
      }
    }
  }
#undef SBR_Q2_STAMP
  if (prof && tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.prof[i] = pacc[i];
    a.prof[6] = pn;
  }
}

// Zq[v][3 + r] <-> Zt[v][r]
__global__ void sbr_q2_shift(const float* __restrict__ in, int64_t ldi, int64_t offi, float* __restrict__ out, int64_t ldo,
                             int64_t offo, int64_t n) {
  const int64_t r = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[r * ldo + offo + c] = in[r * ldi + offi + c];
  // into the shifted layout (offo = 3): the three floats in front of row 0 and the padding behind row n - 1 are read as window rows
  if (offo > 0 && c < offo) out[r * ldo + c] = 0.f;
  if (offo > 0 && c < ldo - offo - n) out[r * ldo + offo + n + c] = 0.f;
}

// 16 (default): 16 KB group images, two groups ahead, passes of eight blocks; 15: passes of four; 14: one group ahead; 3: fp32 products from reflectors staged by every workgroup
// (every product of the solver on the fp32 matrix cores: precision = 0). Measured at n = 30 016, m = 15 008 (profiles/r04_q2_*):
// 500 ms (3), 250 (15). What the phase clocks of a group showed (context option q2_prof, profiles/r04_q2_phase_clocks.log; one wave per
// SIMD at m = n / 2, so nothing hides a wave's own latencies): (i) __syncthreads() carries a release fence = `s_waitcnt vmcnt(0)`, so
// every group waited for ALL memory instructions; (ii) an image takes ~3 000 clocks to arrive whatever is in flight; (iii) the compiler
// put every LDS read next to its use: ~20 exposed round trips per group. Hence the bare barrier, the 16 KB image and the reads of a
// stage issued together.
static int sbr_q2_variant(const Ctx* ctx, int64_t n) {
  int v = ctx->opt.eff_q2_variant();
  if (v != 14 && v != 15 && v != 16) v = 3;
  if (v != 3 && n % SB != 0) v = 3;  // the image index assumes an order that is a multiple of 64 (the two-stage solver pads)
  return v;
}

// what the apply kernel of the selected variant needs besides the reflectors: the groups' T factors, or their finished LDS images
static int sbr_q2_launch_build_t(Ctx* ctx, int64_t n, hipStream_t st) {
  const int64_t ldv2 = sbr_ldv2(n), ldt = n / SB + 2, nsweep = n - 2;
  const float* V2 = static_cast<const float*>(ctx->ws.count("sbr.V2") ? ctx->ws.at("sbr.V2").first : nullptr);
  const float* TAU2 = static_cast<const float*>(ctx->ws.count("sbr.TAU2") ? ctx->ws.at("sbr.TAU2").first : nullptr);
  if (!V2 || !TAU2 || nsweep <= 0) return ctx->fail(SCLENS_ERR_STATE, "sbr_q2_build_t: no reflectors of a preceding sb2st_f32 on this context");
  const int nblk = (int)((nsweep + QW - 1) / QW), nk = (int)((n - 1 + SB - 1) / SB);
  const int variant = sbr_q2_variant(ctx, n);
  if (variant != 3) {
    SCL_WS(ctx, img, float, "sbr.Q2img", (sbr_q2_img_count(n) + 1) * Q_IMG2);
    hipLaunchKernelGGL(sbr_q2_build_img, dim3((unsigned)nk, (unsigned)nblk), dim3(256), 0, st, V2, ldv2, TAU2, ldt, n, img);
  } else {
    SCL_WS(ctx, Tg, float, "sbr.Tg", (int64_t)nblk * nk * QW * QW);
    hipLaunchKernelGGL(sbr_q2_build_t, dim3((unsigned)nk, (unsigned)nblk), dim3(64), 0, st, V2, ldv2, TAU2, ldt, n, nk, Tg);
  }
  ctx->q2_built_variant = variant;
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int sbr_apply_q2(Ctx* ctx, int64_t n, float* Zt, int64_t m, int64_t ldz) {
  if (m <= 0) return SCLENS_OK;
  StageTimer tm(ctx, "sbr_q2");
  const int64_t ldv2 = sbr_ldv2(n), ldt = n / SB + 2;
  const float* V2 = static_cast<const float*>(ctx->ws.count("sbr.V2") ? ctx->ws.at("sbr.V2").first : nullptr);
  const float* TAU2 = static_cast<const float*>(ctx->ws.count("sbr.TAU2") ? ctx->ws.at("sbr.TAU2").first : nullptr);
  if (!V2 || !TAU2) return ctx->fail(SCLENS_ERR_STATE, "sbr_apply_q2: no reflectors of a preceding sb2st_f32 on this context");
  if (ctx->opt.q2_reference) {  // the unblocked reference version (tests)
    int VT = (int)((150 * 1024) / (4 * n));
    if (VT > 8) VT = 8;
    if (VT < 1) return ctx->fail(SCLENS_ERR_ARG, "sbr_apply_q2 (reference version): order too large for one vector in LDS");
    const size_t lds = sizeof(float) * (size_t)VT * (size_t)n;
    SCL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sbr_q2_simple), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(sbr_q2_simple, dim3((unsigned)((m + VT - 1) / VT)), dim3(256), lds, ctx->stream, V2, ldv2, TAU2, ldt, n,
                       Zt, m, ldz, VT);
    SCL_HIP(ctx, hipGetLastError());
    return SCLENS_OK;
  }
  const int64_t nsweep = n - 2;
  if (nsweep <= 0) return SCLENS_OK;
  const int nblk = (int)((nsweep + QW - 1) / QW), nk = (int)((n - 1 + SB - 1) / SB);
  const int q2_variant = sbr_q2_variant(ctx, n);
  if (ctx->q2_tg_n == n && ctx->q2_ev && ctx->q2_built_variant == q2_variant) {  // built on the auxiliary stream after the chase
    SCL_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->q2_ev, 0));                 // (or by an earlier call on these reflectors)
  } else {
    SCL_TRY(sbr_q2_launch_build_t(ctx, n, ctx->stream));
    ctx->q2_tg_n = n;  // valid until the next chase on this context (sb2st_f32 resets it)
  }
  const float* Tg = static_cast<const float*>(ctx->ws.count("sbr.Tg") ? ctx->ws.at("sbr.Tg").first : nullptr);
  const float* q2img = static_cast<const float*>(ctx->ws.count("sbr.Q2img") ? ctx->ws.at("sbr.Q2img").first : nullptr);
  // the apply kernel works on the shifted layout Zq[v][3 + row] (16-byte aligned register quads, see above)
  const int64_t ldq = round_up(n + 3, 4);
  SCL_WS(ctx, Zq, float, "sbr.Zq", m * ldq);
  for (int64_t r0 = 0; r0 < m; r0 += 65535) {
    const int64_t rows = (m - r0 < 65535) ? m - r0 : 65535;
    hipLaunchKernelGGL(sbr_q2_shift, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream, Zt + r0 * ldz, ldz,
                       (int64_t)0, Zq + r0 * ldq, ldq, (int64_t)3, n);
  }
  SbrQ2Args qa{V2, ldv2, Tg, nk, nblk, n, Zq, m, ldq, nullptr};
  if (ctx->opt.q2_prof > 0 && q2_variant != 3) {
    qa.prof = static_cast<unsigned long long*>(ctx->workspace("sbr.q2prof", 8 * sizeof(unsigned long long)));
    if (!qa.prof) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemsetAsync(qa.prof, 0, 8 * sizeof(unsigned long long), ctx->stream));
  }
  const dim3 q2grid((unsigned)((m + 63) / 64));
  if (q2_variant == 15) {  // two groups ahead: three image buffers
    const int lds_bytes = 3 * Q_IMG2 * (int)sizeof(float);
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(sbr_q2_apply16e<4, 12, 3>), lds_bytes));
    hipLaunchKernelGGL((sbr_q2_apply16e<4, 12, 3>), q2grid, dim3(256), lds_bytes, ctx->stream, qa, q2img);
  } else if (q2_variant == 16) {
    // 15 with passes of EIGHT blocks of sweeps instead of four: the window (now 320 rows, 224 VGPRs) is loaded and stored half as often
    // (the window traffic of a pass, 8 memory instructions per task and the drain at its end, was a fifth of the kernel): 264 -> 229 ms
    // at n = 30 016 with 15 008 vectors, 425 -> 340 ms with all vectors, same bits (profiles/r05_q2_pass_length.log). Passes of 6: 234 ms;
    // of 12 (460 VGPRs, one workgroup per CU): 252 ms; of 16: spills.
    const int lds_bytes = 3 * Q_IMG2 * (int)sizeof(float);
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(sbr_q2_apply16e<8, 20, 3>), lds_bytes));
    hipLaunchKernelGGL((sbr_q2_apply16e<8, 20, 3>), q2grid, dim3(256), lds_bytes, ctx->stream, qa, q2img);
  } else if (q2_variant == 14) {
    const int lds_bytes = 2 * Q_IMG2 * (int)sizeof(float);
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(sbr_q2_apply16e<4, 12, 2>), lds_bytes));
    hipLaunchKernelGGL((sbr_q2_apply16e<4, 12, 2>), q2grid, dim3(256), lds_bytes, ctx->stream, qa, q2img);
  } else if (ctx->opt.q2_fp32_blocks == 16) {
    hipLaunchKernelGGL((sbr_q2_apply16v3<16, 36>), q2grid, dim3(256), 0, ctx->stream, qa);
  } else if (ctx->opt.q2_fp32_blocks == 12) {
    hipLaunchKernelGGL((sbr_q2_apply16v3<12, 28>), q2grid, dim3(256), 0, ctx->stream, qa);
  } else if (ctx->opt.q2_fp32_blocks == 8) {
    // the fp32 kernel with longer passes (8 / 12 / 16 blocks of 32 sweeps: a window of up to 576 rows in registers): fewer window loads,
    // stores and drains -- 486 -> 451 / 442 / 439 ms ALONE at order 30 016 with 15 008 vectors, same bits -- but 221 .. 256 VGPRs, and
    // inside a call, beside the twin kernel of the other stream, the call gets SLOWER (40.9 -> 41.7 / 42.1 s at cfg4,
    // profiles/r06_q2_fp32_pass_length.log): the default stays at four blocks
    hipLaunchKernelGGL((sbr_q2_apply16v3<8, 20>), q2grid, dim3(256), 0, ctx->stream, qa);
  } else {
    hipLaunchKernelGGL((sbr_q2_apply16v3<4, 12>), q2grid, dim3(256), 0, ctx->stream, qa);
  }
  for (int64_t r0 = 0; r0 < m; r0 += 65535) {
    const int64_t rows = (m - r0 < 65535) ? m - r0 : 65535;
    hipLaunchKernelGGL(sbr_q2_shift, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream, Zq + r0 * ldq, ldq,
                       (int64_t)3, Zt + r0 * ldz, ldz, (int64_t)0, n);
  }
  SCL_HIP(ctx, hipGetLastError());
  if (qa.prof) {
    unsigned long long h[8];
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SCL_HIP(ctx, hipMemcpy(h, qa.prof, sizeof(h), hipMemcpyDeviceToHost));
    static const char* nm[6] = {"DMA issue", "window loads / stores issue", "products", "counted wait", "barrier", "between groups"};
    unsigned long long tot = 0;
    for (int i = 0; i < 6; ++i) tot += h[i];
    fprintf(stderr, "[sbr_q2 variant %d] n = %lld, m = %lld: %llu groups, %.0f shader clocks per group (wave 0 of one workgroup)\n", q2_variant, (long long)n,
            (long long)m, h[6], (double)tot / (double)std::max<unsigned long long>(1, h[6]));
    for (int i = 0; i < 6; ++i)
      fprintf(stderr, "   %-30s %8.0f clocks (%4.1f %%)\n", nm[i], (double)h[i] / (double)std::max<unsigned long long>(1, h[6]), 100.0 * h[i] / (double)std::max<unsigned long long>(1, tot));
  }
  return SCLENS_OK;
}


}  // namespace scl
