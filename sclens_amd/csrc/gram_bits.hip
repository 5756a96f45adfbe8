// Gram matrix of the scaled matrix of a BINARY count pattern without forming the scaled matrix.
//
// The sparsity search (scLENS.jl:725-751) decomposes ~20 matrices `logn_scale(pre_scale(P))` whose counts are all 0 or 1
// (scLENS.jl:664 binarises, :735 adds ones). For such a matrix every stored entry of cell i has the same value
// l_i = log1p(1 / TGC_i) (proj_l + log1p, scLENS.jl:607, :650), so with the statistics of scale.hip
//     B_ij = s_i (P_ij l_i / std_j - mu_j) - cent_j          (scale.hip header; mean centring, cent_j = column mean)
// and the Gram matrix over genes is a weighted co-occurrence count plus rank-one terms:
//     (B'B)_jk = d_j d_k C_jk - u_j mu_k - mu_j u_k + S2 mu_j mu_k - N cent_j cent_k ,
//     C = P' diag(w) P ,  w_i = (s_i l_i)^2 ,  d_j = 1/std_j ,  u_j = d_j sum_i P_ij s_i^2 l_i ,  S2 = sum(s_i^2) .
// C is the only O(M^2 N) term and both of its factors are exact in 16 bits: P as fp16 0/1, and w split into NW fp16 pieces
// w = w^(1) + w^(2) (+ w^(3)) (22 or 33 significant bits, scaled by a power of two into the fp16 range), so
//     C = sum_t P' (diag(w^(t)) P)
// runs on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) with every product exact: the rounding is that of the fp32
// accumulation, as in the fp32 product it replaces, at 1/16 of the MFMA cycles per piece. The rank-one terms and the final
// combination are evaluated in fp64 in the epilogue.
//
// Kernel = the 256 x 256 NT kernel of gemm.hip re-typed: the same direct global -> LDS staging (rows of 128 B = 64 cells of
// fp16), the same swizzle and XCD-aware tile list; per 16 cells a wave reads its 4 + 2 operand fragments and the NW weight
// fragments (LDS broadcast), forms the weighted B fragments with packed fp16 multiplies (1.0 * w and 0 * w are exact) and
// issues 8 NW MFMAs.
#include <algorithm>

#include "common.h"
#include "pattern.h"

namespace scl {

int big_tile_list(Ctx* ctx, int64_t tm, int64_t tn, int lower, const int2** out, int64_t* nblocks);  // gemm.hip

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float gb_f32x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 gb_f16x2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// w_i = (s_i l_i)^2 and sa_i = s_i^2 l_i, l_i as the closure path rounds it (scale.hip k_col_stats); block maxima -> wmax_part
__global__ __launch_bounds__(256) void k_cell_weights(int64_t N, const double* __restrict__ tgc, const double* __restrict__ srow,
                                                      int f32path, double* __restrict__ w, double* __restrict__ sa,
                                                      double* __restrict__ wmax_part) {
  __shared__ double sm[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double x = 0.0;
  if (i < N) {
    const double t = tgc[i];
    double l = 0.0;
    if (t > 0.0) l = f32path ? (double)log1pf((1.0f / (float)t) * 1.0f) : log1p(1.0 / t);
    const double a = srow[i] * l;
    x = a * a;
    w[i] = x;
    sa[i] = srow[i] * a;
  }
  double m = x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) wmax_part[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// sc[0] = 1 / scale with scale = the power of two that puts max w into [2^14, 2^15); sc[1] = scale; sc[2] = sum s_i^2
__global__ __launch_bounds__(1024) void k_weight_scale(const double* __restrict__ wmax_part, int64_t nparts,
                                                       const double* __restrict__ srow, int64_t N, double* __restrict__ sc) {
  __shared__ double sm[16], ss[16];
  double m = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < nparts; i += 1024) m = fmax(m, wmax_part[i]);
  for (int64_t i = threadIdx.x; i < N; i += 1024) s2 += srow[i] * srow[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  s2 = wsum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6] = m;
    ss[threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mm = 0.0, t = 0.0;
    for (int w = 0; w < 16; ++w) {
      mm = fmax(mm, sm[w]);
      t += ss[w];
    }
    const double scale = (mm > 0.0 && isfinite(mm)) ? ldexp(1.0, 14 - ilogb(mm)) : 1.0;
    sc[0] = 1.0 / scale;
    sc[1] = scale;
    sc[2] = t;
  }
}

// wq[(i / 8) * NW + t][i % 8] = t-th fp16 piece of scale * w_i (0 for the padding cells i >= N)
__global__ __launch_bounds__(256) void k_split_weights(const double* __restrict__ w, int64_t N, int64_t ldm, int nw,
                                                       const double* __restrict__ sc, _Float16* __restrict__ wq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ldm) return;
  double x = i < N ? w[i] * sc[1] : 0.0;
  for (int t = 0; t < nw; ++t) {
    const _Float16 hv = (_Float16)(float)x;  // any rounding will do: the residual below is exact in fp64
    wq[((i >> 3) * nw + t) * 8 + (i & 7)] = hv;
    x -= (double)hv;
  }
}

// gv[j] = (d_j, u_j, mu_j, cent_j), u_j = d_j sum over the cells of gene j of sa_i (CSC view, one wave per gene, fixed order)
__global__ __launch_bounds__(256) void k_gene_vecs(PatternDev p, const float* __restrict__ val, const double* __restrict__ sa,
                                                   const double* __restrict__ stdv, const double* __restrict__ mu,
                                                   const double* __restrict__ cent, double4* __restrict__ gv) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  double s = 0.0;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64)
    if (val[q] != 0.f) s += sa[p.row[q]];
  s = wsum_d(s);
  const double d = 1.0 / stdv[col];
  if (lane == 0) gv[col] = make_double4(d, d * s, mu[col], cent[col]);
}

// fp16 image of P, genes-major: Pm[j * ldm + i] = 1.0 where val != 0 (the buffer was zero-filled)
__global__ __launch_bounds__(256) void k_mask_scatter(PatternDev p, const float* __restrict__ val, unsigned short* __restrict__ Pm,
                                                      int64_t ldm) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  unsigned short* dst = Pm + col * ldm;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64)
    if (val[q] != 0.f) dst[p.row[q]] = 0x3C00u;
}

// The same image written ONCE (round 5): one workgroup per gene walks the cells in chunks of MF_CH, zeroes the chunk in LDS, marks the
// gene's stored entries that fall into it and stores the chunk with 16-byte stores -- no 6 GB memset in front, no 2-byte scattered
// stores that re-write a sector each (memset 1.5 ms + scatter 5.1 ms at 100 000 x 30 000, profiles/r05_cfg4_kernel_stats_one_stream.csv).
// The walk is k_dense_fused's (scale.hip): a column holds its counts with ascending cells, then the zero candidates in draw order; a
// running position covers the ordered prefix, what is left after the last chunk is scattered behind a barrier.
constexpr int MF_CH = 8192;
__global__ __launch_bounds__(256) void k_mask_fused(PatternDev p, const float* __restrict__ val, unsigned short* __restrict__ Pm,
                                                    int64_t ldm) {
  __shared__ __attribute__((aligned(16))) unsigned short chunk[MF_CH];
  __shared__ int wlead[4];
  const int64_t col = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int64_t q = p.colptr[col];
  const int64_t qend = p.colptr[col + 1];
  unsigned short* dst = Pm + col * ldm;
  for (int64_t c0 = 0; c0 < ldm; c0 += MF_CH) {
    const int64_t cend = (c0 + MF_CH < ldm) ? c0 + MF_CH : ldm;
    for (int c = tid; c < MF_CH / 8; c += 256) reinterpret_cast<uint4*>(chunk)[c] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    while (true) {  // block-uniform trip count
      const int64_t idx = q + tid;
      const int64_t r = (idx < qend) ? (int64_t)p.row[idx] : (int64_t)-1;
      const int in = (r >= c0 && r < cend) ? 1 : 0;
      const unsigned long long out = ~__ballot(in);
      if (lane == 0) wlead[wv] = out ? (__ffsll((long long)out) - 1) : 64;  // this wave's leading entries inside the chunk
      __syncthreads();
      int cnt = wlead[0];
      if (cnt == 64) cnt += wlead[1];
      if (cnt == 128) cnt += wlead[2];
      if (cnt == 192) cnt += wlead[3];
      if (tid < cnt && val[idx] != 0.f) chunk[r - c0] = 0x3C00u;
      __syncthreads();
      q += cnt;
      if (cnt < 256) break;
    }
    for (int c = 8 * tid; c0 + c < cend; c += 2048)  // ldm is a multiple of 64: whole 16-byte pieces
      *reinterpret_cast<uint4*>(dst + c0 + c) = *reinterpret_cast<const uint4*>(chunk + c);
    __syncthreads();
  }
  for (int64_t idx = q + tid; idx < qend; idx += 256)  // the entries that did not come in cell order
    if (val[idx] != 0.f) dst[p.row[idx]] = 0x3C00u;
}

struct GramBitsArgs {
  const unsigned short* Pm;  // [n][ldm] fp16 0/1, zero beyond the N cells
  const _Float16* wq;        // [ldm / 8][NW][8]
  const double4* gv;         // [n]
  const double* sc;          // [0] = 1/scale of the weights, [2] = sum s^2
  float* C;
  int64_t n, ldm, ldc;
  double Ncells, inv_div;
  int accumulate;  // C += (chunked sessions: the Gram matrix is a sum over chunks of cells)
};

// PIPE (round 5, context option split_pipe): the stage loop as a software pipeline -- see split_mainloop_pipe; here a stage is four
// units of 8 NW matrix instructions (one per 16 cells), the barrier sits between units 2 and 3, the DMA pieces go out one per two
// matrix instructions behind it (four during unit 3, the other four and the weights during unit 0 of the next stage).
template <int NW, bool PIPE>
__global__ __launch_bounds__(512, 2) void gram_bits_kernel(GramBitsArgs a, const int2* __restrict__ tiles) {
  constexpr int TM = 4, TN = 2;
  constexpr int OPB = 256 * 128;             // bytes of one operand image: 256 rows of 64 fp16
  constexpr int STAGE = 2 * OPB + 128 * NW;  // + the weight pieces of the stage's 64 cells
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int2 tl = tiles[blockIdx.x];
  if (tl.x < 0) return;
  const int64_t m0 = (int64_t)tl.x * 256, n0 = (int64_t)tl.y * 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, l31 = lane & 31, h = lane >> 5;

  // staging: wave w moves the 8-row groups 4 w .. 4 w + 3 of both operands; chunk q of row r lands in slot q ^ ((r >> 1) & 7)
  const int srow = lane >> 3, sq = lane & 7;
  const unsigned short* srcA[4];
  const unsigned short* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    const int chunk = sq ^ ((r >> 1) & 7);
    int64_t ra = m0 + r, rb = n0 + r;
    if (ra > a.n - 1) ra = a.n - 1;
    if (rb > a.n - 1) rb = a.n - 1;
    srcA[i] = a.Pm + ra * a.ldm + 8 * chunk;
    srcB[i] = a.Pm + rb * a.ldm + 8 * chunk;
  }
  const int64_t nkt = a.ldm / 64;
  auto stage = [&](int buf, int64_t kt) {
    unsigned char* As = lds + buf * STAGE;
    unsigned char* Bs = As + OPB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[i] + kt * 64), (lds_void_t*)(As + (wid * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[i] + kt * 64), (lds_void_t*)(Bs + (wid * 4 + i) * 1024), 16, 0, 0);
    if (wid == 7 && lane < 8 * NW)  // 8 chunks x NW pieces x 16 B, lane-linear
      __builtin_amdgcn_global_load_lds((glb_void_t*)(a.wq + (kt * 8 * NW + lane) * 8), (lds_void_t*)(Bs + OPB), 16, 0, 0);
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * 128 + i * 32 + l31;
    offA[i] = r * 128;
    swA[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * 64 + j * 32 + l31;
    offB[j] = OPB + r * 128;
    swB[j] = (r >> 1) & 7;
  }

  if (PIPE) {
    // one DMA piece of the stage kt (clamped to the last one: a reload nobody reads) into buffer buf: 0..3 A, 4..7 B, 8 the weights
    auto piece = [&](int buf, int64_t kt, int pc) {
      if (kt > nkt - 1) kt = nkt - 1;
      unsigned char* As = lds + buf * STAGE;
      if (pc < 4) __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[pc] + kt * 64), (lds_void_t*)(As + (wid * 4 + pc) * 1024), 16, 0, 0);
      else if (pc < 8) __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[pc - 4] + kt * 64), (lds_void_t*)(As + OPB + (wid * 4 + pc - 4) * 1024), 16, 0, 0);
      else if (wid == 7 && lane < 8 * NW)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(a.wq + (kt * 8 * NW + lane) * 8), (lds_void_t*)(As + 2 * OPB), 16, 0, 0);
    };
    h16x8 av[2][TM], bv[2][TN], wv[2][NW];
    auto reads = [&](int set, const unsigned char* S, int kk) {
      const int c = 2 * kk + h;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[set][i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((c ^ swA[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[set][j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((c ^ swB[j]) << 4));
#pragma unroll
      for (int t = 0; t < NW; ++t) wv[set][t] = *reinterpret_cast<const h16x8*>(S + 2 * OPB + ((c * NW + t) << 4));
    };
    // a unit = the cells of one kk: NW x TM groups of TN matrix instructions; `pre(g)` in front of group g, `next()` behind group 0
    auto unit = [&](int set, auto pre, auto next) {
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        h16x8 bw[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bw[j] = bv[set][j] * wv[set][t];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          pre(t * TM + i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[set][i], bw[j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (t == 0 && i == 0) {
            next();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    auto none = [](int) {};
#pragma unroll
    for (int pc = 0; pc < 9; ++pc) piece(0, 0, pc);
    __syncthreads();
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) piece(1, 1, pc);
    reads(0, lds, 0);
    for (int64_t kt = 0; kt < nkt; ++kt) {
      const int cur = (int)(kt & 1);
      const unsigned char* S = lds + cur * STAGE;
      const unsigned char* Sn = lds + (cur ^ 1) * STAGE;
      unit(0, [&](int g) { if (g < 5) piece(cur ^ 1, kt + 1, 4 + g); }, [&]() { reads(1, S, 1); });
      unit(1, none, [&]() { reads(0, S, 2); });
      unit(0, none, [&]() { reads(1, S, 3); });
      __syncthreads();  // every wave has issued its reads of `cur`; stage kt + 1 has landed in the other buffer
      unit(1, [&](int g) { if (g < 4) piece(cur, kt + 2, g); }, [&]() { reads(0, Sn, 0); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA piece may outlive the workgroup's LDS
  } else {
  stage(0, 0);
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const unsigned char* S = lds + buf * STAGE;
    const unsigned char* W = S + 2 * OPB;
    // MFMA step kk takes the cells 8 (2 kk + h) .. + 7 of the stage: lane half h reads chunk c = 2 kk + h of A, B and w
    h16x8 av[2][TM], bv[2][TN], wv[2][NW];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((h ^ swA[i]) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((h ^ swB[j]) << 4));
#pragma unroll
    for (int t = 0; t < NW; ++t) wv[0][t] = *reinterpret_cast<const h16x8*>(W + ((h * NW + t) << 4));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < 4) {
        const int c = 2 * (kk + 1) + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nxt][i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((c ^ swA[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nxt][j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((c ^ swB[j]) << 4));
#pragma unroll
        for (int t = 0; t < NW; ++t) wv[nxt][t] = *reinterpret_cast<const h16x8*>(W + ((c * NW + t) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads of the next step ahead of this step's MFMAs
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        h16x8 bw[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bw[j] = bv[cur][j] * wv[cur][t];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cur][i], bw[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  }

  // ---- epilogue (D layout: col = lane & 31, rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5)): fp64 combination with the rank-one
  //      terms, lower triangle + mirrored 16-byte stores
  const double unscale = a.sc[0], S2 = a.sc[2];
  const bool vec_mirror = (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + l31;
      const int64_t row0 = m0 + wm * 128 + i * 32 + 4 * h;
      if (col >= a.n) continue;
      const double4 gc = a.gv[col];
      const double cd = gc.x * unscale;
      v16f outv;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
        double v = 0.0;
        if (row < a.n) {
          const double4 gr = a.gv[row];
          v = (double)acc[i][j][e] * (cd * gr.x) - gr.y * gc.z - gr.z * gc.y + S2 * (gr.z * gc.z) - a.Ncells * (gr.w * gc.w);
        }
        outv[e] = (float)(v * a.inv_div);
        if (a.accumulate && row < a.n && col <= row) outv[e] += a.C[row * a.ldc + col];  // (the mirrored entry is written from here too)
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
        if (row < a.n && col <= row) a.C[row * a.ldc + col] = outv[e];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t rowq = row0 + 8 * q;
        if (vec_mirror && rowq + 3 < a.n && col < rowq) {
          v4f v = {outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
          *reinterpret_cast<v4f*>(&a.C[col * a.ldc + rowq]) = v;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (rowq + u < a.n && col < rowq + u) a.C[col * a.ldc + rowq + u] = outv[4 * q + u];
        }
      }
    }
}

template <int NW>
int launch_gram_bits(Ctx* ctx, const GramBitsArgs& a, const int2* tiles, int64_t nb) {
  constexpr int LDS_BYTES = 2 * (2 * 256 * 128 + 128 * NW);
  if (ctx->opt.split_pipe != 0) {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gram_bits_kernel<NW, true>), LDS_BYTES));
    hipLaunchKernelGGL((gram_bits_kernel<NW, true>), dim3((unsigned)nb), dim3(512), LDS_BYTES, ctx->stream, a, tiles);
  } else {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gram_bits_kernel<NW, false>), LDS_BYTES));
    hipLaunchKernelGGL((gram_bits_kernel<NW, false>), dim3((unsigned)nb), dim3(512), LDS_BYTES, ctx->stream, a, tiles);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}


// ---- fp32-accurate NT product on the fp16 MFMA from operands split into two fp16 pieces ------------------------------------
// The search statistic max_i |Vr2_i' nV2_j| (scLENS.jl:742) is a 3 10^4 x 1.5 10^4 x 3 10^4 product of unit vectors per step.
// Each operand is scaled by 2^12 and split x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant bits, the residual
// is exact in fp32); a b = ah bh + ah bl + al bh up to the dropped al bl <= 2^-22 |a b| -- three fp16 MFMAs with fp32
// accumulation in place of sixteen fp32 MFMA cycles. "Split image" of a row-major [rows][K] matrix: per row and per 32 k,
// 32 hi then 32 lo halves (128 B), zero beyond K: the staging, swizzle and fragment reads of gram_bits_kernel apply unchanged
// (chunks 0-3 of a stage row are the hi pieces, 4-7 the lo pieces).
__global__ __launch_bounds__(256) void k_split_image(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld, int64_t Kp,
                                                     _Float16* __restrict__ dst) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float x = k < K ? src[r * ld + k] * 4096.f : 0.f;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    _Float16* d = dst + r * 2 * Kp + (k >> 5) * 64 + (k & 31);
    d[0] = hi;
    d[32] = lo;
  }
}

struct SplitCorrArgs {
  const _Float16* A;  // split image, M rows
  const _Float16* B;  // split image, N rows
  int64_t M, N, Kp;
  unsigned* colabsmax;  // [N] float bit patterns, zero-filled by the caller
  float alpha;          // 2^-24: undoes the two operand scalings
};

// (Measured and removed in round 5: the operand stages through REGISTERS -- global_load_dwordx4 -> ds_write_b128 -- instead of
// LDS-DMA: the same time, profiles/r04_corr_register_staging.log.)
__global__ __launch_bounds__(512, 2) void corr_split_kernel(SplitCorrArgs a, const int2* __restrict__ tiles) {
  constexpr int TM = 4, TN = 2;
  constexpr int OPB = 256 * 128, STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int2 tl = tiles[blockIdx.x];
  if (tl.x < 0) return;
  const int64_t m0 = (int64_t)tl.x * 256, n0 = (int64_t)tl.y * 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, l31 = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, sq = lane & 7;
  const _Float16* srcA[4];
  const _Float16* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    const int chunk = sq ^ ((r >> 1) & 7);
    int64_t ra = m0 + r, rb = n0 + r;
    if (ra > a.M - 1) ra = a.M - 1;
    if (rb > a.N - 1) rb = a.N - 1;
    srcA[i] = a.A + ra * 2 * a.Kp + 8 * chunk;
    srcB[i] = a.B + rb * 2 * a.Kp + 8 * chunk;
  }
  const int64_t nkt = a.Kp / 32;
  auto stage = [&](int buf, int64_t kt) {
    unsigned char* As = lds + buf * STAGE;
    unsigned char* Bs = As + OPB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[i] + kt * 64), (lds_void_t*)(As + (wid * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[i] + kt * 64), (lds_void_t*)(Bs + (wid * 4 + i) * 1024), 16, 0, 0);
  };
  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * 128 + i * 32 + l31;
    offA[i] = r * 128;
    swA[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * 64 + j * 32 + l31;
    offB[j] = OPB + r * 128;
    swB[j] = (r >> 1) & 7;
  }
  stage(0, 0);
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const unsigned char* S = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {  // 16 k per step: lane half h takes the 8 k of chunk 2 kk + h (hi) and 4 + 2 kk + h (lo)
      const int ch = 2 * kk + h, cl = 4 + 2 * kk + h;
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((ch ^ swA[i]) << 4));
        al[i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((cl ^ swA[i]) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((ch ^ swB[j]) << 4));
        bl[j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((cl ^ swB[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // column maxima of |alpha acc| (rows beyond M hold copies of row M - 1: harmless for a maximum)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(a.alpha * acc[i][j][e]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int64_t col = n0 + wn * 64 + j * 32 + l31;
    if (h == 0 && col < a.N) atomicMax(&a.colabsmax[col], __float_as_uint(mx));
  }
}

// The same product with the stage loop written as a software pipeline (round 5; context option split_pipe). What the two-buffer loop
// above loses per 32 of K (ISA + stage clocks: 6 500 clocks against 3 100 of matrix instructions at two waves per SIMD): its 8 DMA
// instructions are issued back to back at the top of a stage by BOTH waves of a SIMD at once (60-185 clocks of issue each, nothing on
// the matrix pipe meanwhile), the compiler places every group of LDS reads right in front of its matrix instructions with a full
// `lgkmcnt(0)` wait (~10 exposed LDS round trips per stage, again in both waves at once), and the stage ends in a barrier behind which
// the first reads of the next stage are exposed once more. Here a stage is four units of 12 matrix instructions (u = 2 kk + half of
// the row tiles); the fragments of unit u + 1 are read before the matrix instructions of unit u (two register sets), the barrier sits
// between units 2 and 3 -- the point at which every wave has issued its last read of the current buffer and the next stage has
// landed -- and the DMA pieces of the stage after next go out one per three matrix instructions behind that barrier (four during
// unit 3, four during unit 0 of the next stage), so that the matrix pipe stays fed while they issue. Same products in the same
// order per accumulator: the same bits as the loop above.
// acc += A B' over `nst` stages of 32 of K starting at byte offset koff0 of the image rows (A: 256 rows from baseA, B: 256 rows from
// baseB, rows beyond mlast / nlast re-read the last one); all 512 threads of the workgroup, 128 KB of dynamic LDS. scale_acc: the
// accumulators hold values whose loads may still be in flight (C of a read-modify-write product): they are multiplied by acc_scale
// behind the wait for the first stage, which covers those loads (memory instructions complete in issue order)
__device__ __forceinline__ void split_mainloop_pipe(v16f (&acc)[4][2], unsigned char* lds, const unsigned char* baseA, const unsigned char* baseB,
                                                    unsigned rowb, int mlast, int nlast, unsigned koff0, int64_t nst, bool scale_acc = false,
                                                    float acc_scale = 1.f) {
  constexpr int TN = 2;
  constexpr int OPB = 256 * 128, STAGE = 2 * OPB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, l31 = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, sq = lane & 7;
  // DMA sources: a uniform base per operand + one 32-bit byte offset per piece (a stage row is 128 bytes, an image row 4 Kp bytes: a
  // tile's 256 rows span < 2^31 bytes); pieces 0..3 are the wave's four 8-row groups of A, 4..7 those of B
  unsigned offp[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    const unsigned chunk = (unsigned)(sq ^ ((r >> 1) & 7));
    offp[i] = (unsigned)(r < mlast ? r : mlast) * rowb + 16u * chunk;
    offp[4 + i] = (unsigned)(r < nlast ? r : nlast) * rowb + 16u * chunk;
  }
  const unsigned ldsw = (unsigned)(wid * 4096);  // the wave's four 1 KB pieces of an operand image
  auto dma = [&](int buf, unsigned koff, int piece) {  // piece 0..7 of the stage at byte offset koff of the image rows into buffer buf
    unsigned char* dst = lds + buf * STAGE + (piece >> 2) * OPB + ldsw + (piece & 3) * 1024;
    const unsigned char* g = (piece < 4 ? baseA : baseB) + (offp[piece] + koff);
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)dst, 16, 0, 0);
  };
  // fragment addresses: row r = wm 128 + i 32 + l31 of A (wn 64 + j 32 + l31 of B) at r * 128 bytes, chunk c in slot c ^ ((r >> 1) & 7)
  // = c ^ ((l31 >> 1) & 7) for every tile of the lane: one base per operand, the tile as an immediate, four slot offsets per lane
  const int sw = (l31 >> 1) & 7;
  const int fa = (wm * 128 + l31) * 128, fb = OPB + (wn * 64 + l31) * 128;
  int slot[2][2];  // [kk][hi / lo]: byte offset of the lane's chunk inside its row
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    slot[kk][0] = ((2 * kk + h) ^ sw) << 4;
    slot[kk][1] = ((4 + 2 * kk + h) ^ sw) << 4;
  }
  h16x8 ah[2][2], al[2][2], bh[2][TN], bl[2][TN];  // [register set][row tile of the unit / column tile]
  auto readA = [&](int set, const unsigned char* S, int kk, int half) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * half + q;
      ah[set][q] = *reinterpret_cast<const h16x8*>(S + fa + i * 4096 + slot[kk][0]);
      al[set][q] = *reinterpret_cast<const h16x8*>(S + fa + i * 4096 + slot[kk][1]);
    }
  };
  auto readB = [&](int set, const unsigned char* S, int kk) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[set][j] = *reinterpret_cast<const h16x8*>(S + fb + j * 4096 + slot[kk][0]);
      bl[set][j] = *reinterpret_cast<const h16x8*>(S + fb + j * 4096 + slot[kk][1]);
    }
  };
  // the 12 matrix instructions of a unit in four groups of three; mode 1 / 2: DMA piece g / 4 + g of the stage at `koff` into `buf` goes
  // out in front of group g. No branches in the loop body: beyond the last stage the offsets are clamped to it (a reload nobody reads).
  // `next()` issues the LDS reads of the following unit BEHIND the first group: the wait the compiler puts in front of a unit's first
  // matrix instruction is `lgkmcnt(0)`, which must not cover reads issued a moment ago
  const unsigned klast = koff0 + (unsigned)(nst - 1) * 128u;
  auto unit = [&](int aset, int bset, int half, int mode, int buf, unsigned koff, auto next) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (mode) dma(buf, koff < klast ? koff : klast, (mode == 2 ? 4 : 0) + q * TN + j);
        __builtin_amdgcn_sched_barrier(0);
        const int i = 2 * half + q;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[aset][q], bh[bset][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[aset][q], bl[bset][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[aset][q], bh[bset][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q == 0 && j == 0) {
          next();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  // prologue: stage 0 landed, the first half of stage 1 in flight, the fragments of unit 0 in register set 0
#pragma unroll
  for (int pc = 0; pc < 8; ++pc) dma(0, koff0, pc);
  __syncthreads();
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) dma(1, koff0 + 128u < klast ? koff0 + 128u : klast, pc);
  if (scale_acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] *= acc_scale;
  }
  readA(0, lds, 0, 0);
  readB(0, lds, 0);
  for (int64_t t = 0; t < nst; ++t) {
    const int cur = (int)(t & 1);
    const unsigned char* S = lds + cur * STAGE;
    const unsigned char* Sn = lds + (cur ^ 1) * STAGE;
    const unsigned koff = koff0 + (unsigned)t * 128u;
    // unit 0 (kk 0, row tiles 0-1): sets A0, B0; reads of unit 1 -> A1; the second half of the DMA of stage t + 1 goes out here
    unit(0, 0, 0, 2, cur ^ 1, koff + 128u, [&]() { readA(1, S, 0, 1); });
    // unit 1 (kk 0, row tiles 2-3): sets A1, B0; reads of unit 2: A (kk 1, tiles 0-1) -> A0, B (kk 1) -> B1. (Set A0 is free: unit 0
    // is done; B1 was last used by unit 3 of the previous stage.)
    unit(1, 0, 1, 0, 0, 0u, [&]() { readA(0, S, 1, 0); readB(1, S, 1); });
    // unit 2 (kk 1, row tiles 0-1): sets A0, B1; reads of unit 3 -> A1: the last reads of this buffer
    unit(0, 1, 0, 0, 0, 0u, [&]() { readA(1, S, 1, 1); });
    // every wave has issued its reads of `cur`, stage t + 1 has landed in the other buffer (vmcnt(0) + barrier)
    __syncthreads();
    // unit 3 (kk 1, row tiles 2-3): sets A1, B1; the fragments of unit 0 of stage t + 1 -> A0, B0; first half of the DMA of stage t + 2
    unit(1, 1, 1, 1, cur, koff + 256u, [&]() { readA(0, Sn, 0, 0); readB(0, Sn, 0); });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA piece may outlive the workgroup's LDS
}

__global__ __launch_bounds__(512, 2) void corr_split_kernel_p(SplitCorrArgs a, const int2* __restrict__ tiles) {
  constexpr int TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int2 tl = tiles[blockIdx.x];
  if (tl.x < 0) return;
  tl.x = __builtin_amdgcn_readfirstlane(tl.x);
  tl.y = __builtin_amdgcn_readfirstlane(tl.y);
  const int64_t m0 = (int64_t)tl.x * 256, n0 = (int64_t)tl.y * 256;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wn = wid & 3, l31 = lane & 31, h = lane >> 5;
  const unsigned rowb = (unsigned)(4 * a.Kp);
  const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.A) + m0 * rowb;
  const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.B) + n0 * rowb;
  const int mlast = (int)((a.M - 1 - m0 < 255) ? a.M - 1 - m0 : 255), nlast = (int)((a.N - 1 - n0 < 255) ? a.N - 1 - n0 : 255);
  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  split_mainloop_pipe(acc, lds, baseA, baseB, rowb, mlast, nlast, 0u, a.Kp / 32);
  // column maxima of |alpha acc| (rows beyond M hold copies of row M - 1: harmless for a maximum)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(a.alpha * acc[i][j][e]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int64_t col = n0 + wn * 64 + j * 32 + l31;
    if (h == 0 && col < a.N) atomicMax(&a.colabsmax[col], __float_as_uint(mx));
  }
}

// ---- C += P Q' from split images, accumulators started from C (round 3) ------------------------------------------------------
__global__ __launch_bounds__(256) void k_absmax_bits(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld,
                                                     unsigned* __restrict__ out) {
  float mx = 0.f;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
    for (int64_t k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(src[r * ld + k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __shared__ float wmx[4];
  if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
  __syncthreads();
  // one atomic per workgroup (8 192 atomics on one word took 0.7 ms per call); non-negative floats order like their bit patterns
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]))));
}
__global__ void k_pick_scale(const unsigned* __restrict__ mxbits, float* __restrict__ scale) {
  const float mx = __uint_as_float(mxbits[0]);
  int ex = 0;
  float sc = 1.f;
  if (mx > 0.f && mx < 1.0e30f) {
    (void)frexpf(mx, &ex);       // mx = f 2^ex, 1/2 <= f < 1
    sc = ldexpf(1.f, 14 - ex);   // mx sc in [2^13, 2^14)
  }
  scale[0] = sc;
}
__global__ void k_set_scale(float* __restrict__ scale, float v) { scale[0] = v; }
__global__ __launch_bounds__(256) void k_split_image_scaled(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld, int64_t Kp,
                                                            const float* __restrict__ scale, _Float16* __restrict__ dst) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  const float sc = scale[0];
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float x = k < K ? src[r * ld + k] * sc : 0.f;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    _Float16* d = dst + r * 2 * Kp + (k >> 5) * 64 + (k & 31);
    d[0] = hi;
    d[32] = lo;
  }
}

// The split image of a short, fat block (the b rows of a subspace-iteration block: magnitudes differ by up to 1e5 from row to row) with
// ONE POWER-OF-TWO SCALE PER ROW: row r is scaled so that its largest |entry| lands in [2^13, 2^14) and inv_scale[r] receives the
// reciprocal (exact), which the consumer multiplies back into row r of the product. Two launches over (chunks of the row, rows): the
// largest |entry| of every row by atomicMax on the bit patterns of non-negative floats (order-independent), then the image. (One
// workgroup per row took 0.55 ms per call at 64 x 100 000: 22 ms of every ensemble member, profiles/r05_cfg4_kernel_stats_*.csv.)
__global__ __launch_bounds__(256) void k_row_absmax_bits(const float* __restrict__ src, int64_t K, int64_t ld, unsigned* __restrict__ out) {
  const int64_t r = blockIdx.y, k0 = (int64_t)blockIdx.x * 4096;
  const float* a = src + r * ld;
  float mx = 0.f;
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    const int64_t k = k0 + threadIdx.x + 256 * q;
    if (k < K) mx = fmaxf(mx, fabsf(a[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(out + r, __float_as_uint(mx));
}
__global__ __launch_bounds__(256) void k_split_image_rows(const float* __restrict__ src, int64_t K, int64_t ld, int64_t Kp,
                                                          const unsigned* __restrict__ mxbits, _Float16* __restrict__ dst,
                                                          float* __restrict__ inv_scale) {
  const int64_t r = blockIdx.y, k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float mx = __uint_as_float(mxbits[r]);
  float sc = 1.f;
  if (mx > 0.f && mx < 3.0e38f) {
    int ex = 0;
    (void)frexpf(mx, &ex);
    int e = 14 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);  // sc and 1 / sc stay normal numbers
    sc = ldexpf(1.f, e);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[r] = 1.f / sc;
  if (k >= Kp) return;
  const float x = k < K ? src[r * ld + k] * sc : 0.f;
  const _Float16 hi = (_Float16)x;
  _Float16* d = dst + r * 2 * Kp + (k >> 5) * 64 + (k & 31);
  d[0] = hi;
  d[32] = (_Float16)(x - (float)hi);
}

// The split image of the TRANSPOSE of src [rows][K] (image rows = the K columns of src, contraction index = its rows) under the scale
// scale[0] that the image of src itself carries: tiles of 32 rows x 64 columns through LDS, 16-byte stores of eight hi / eight lo pieces.
__global__ __launch_bounds__(256) void k_split_image_transposed(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld, int64_t Rp,
                                                                const float* __restrict__ scale, _Float16* __restrict__ dst) {
  __shared__ float tile[32][65];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
  const float sc = scale[0];
  {
    const int rr = t >> 3, cc = (t & 7) * 8;
    const int64_t r = r0 + rr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t c = c0 + cc + e;
      tile[rr][cc + e] = (r < rows && c < K) ? src[r * ld + c] * sc : 0.f;
    }
  }
  __syncthreads();
  const int cell = t >> 2, part = t & 3;
  const int64_t c = c0 + cell;
  if (c >= K) return;
  h16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = tile[part * 8 + e][cell];
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  _Float16* d = dst + c * 2 * Rp + (r0 >> 5) * 64 + part * 8;
  *reinterpret_cast<h16x8*>(d) = hi;
  *reinterpret_cast<h16x8*>(d + 32) = lo;
}

// ---- C_s[M <= 64][N] = A B' from split images, for a short, fat A (the block of the partial eigensolver against the scaled matrix) ----
// The 256 x 256 kernel spends a stage (64 KB of operands in flight per CU) waiting for the slowest of its DMA pieces, ~6 000 clocks,
// whatever the matrix instruction: a product with 64 rows on it pays for 256 -- 4 ms per pass over a 12 GB operand, as long as the fp32
// product it would replace (profiles/r05_pipe_masks_chefsi_split.log). Here the tile is 64 x 256: a stage is 8 KB of A + 32 KB of B, two
// stage buffers are 80 KB, so TWO workgroups share a CU and 64 KB of the streamed operand are in flight per CU: the kernel moves B at
// the rate HBM delivers it. 8 waves; wave w owns the 32 columns 32 w .. of the tile for both 32-row tiles of A (2 x 16 accumulators).
// rowscale (nullable): row m of the result is multiplied by rowscale[m] (the per-row power-of-two scales of split_image_rows).
struct SplitSkinnyArgs {
  const _Float16* A;  // split image, M <= 64 rows
  const _Float16* B;  // split image, N rows
  const float* sA;    // scales (device); sA may be a constant 1 when A carries per-row scales
  const float* sB;
  const float* rowscale;
  int64_t M, N, Kp;
  float* C;
  int64_t ldc;
  int64_t kt_chunk;     // 32-deep steps per K-slice (gridDim.y slices)
  int64_t c_split_off;  // slab pitch
  float post;
};
__global__ __launch_bounds__(512, 4) void gemm_split_skinny_kernel(SplitSkinnyArgs a) {
  constexpr int OPA = 64 * 128, OPB = 256 * 128, STAGE = OPA + OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int64_t n0 = (int64_t)blockIdx.x * 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5, srow = lane >> 3, sq = lane & 7;
  int64_t kt_lo = (int64_t)blockIdx.y * a.kt_chunk, nkt = a.Kp / 32;
  {
    const int64_t kt_hi = kt_lo + a.kt_chunk;
    nkt = kt_hi < nkt ? kt_hi : nkt;
    if (kt_lo > nkt) kt_lo = nkt;
  }
  float* C = a.C + (int64_t)blockIdx.y * a.c_split_off;
  // DMA pieces of a stage: 8 of A (wave w moves rows 8 w .. 8 w + 7) and 32 of B (wave w moves rows 32 w .. 32 w + 31 in four pieces);
  // chunk q of row r lands in slot q ^ ((r >> 1) & 7) by permuting the source
  const unsigned rowb = (unsigned)(4 * a.Kp);
  const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.B) + n0 * rowb;
  const int mlast = (int)(a.M - 1), nlast = (int)((a.N - 1 - n0 < 255) ? a.N - 1 - n0 : 255);
  unsigned offA, offB[4];
  {
    const int r = wid * 8 + srow;
    offA = (unsigned)(r < mlast ? r : mlast) * rowb + 16u * (unsigned)(sq ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    offB[i] = (unsigned)(r < nlast ? r : nlast) * rowb + 16u * (unsigned)(sq ^ ((r >> 1) & 7));
  }
  auto stage = [&](int buf, int64_t kt) {
    unsigned char* As = lds + buf * STAGE;
    unsigned char* Bs = As + OPA;
    const unsigned koff = (unsigned)kt * 128u;
    __builtin_amdgcn_global_load_lds((glb_void_t*)(baseA + (offA + koff)), (lds_void_t*)(As + wid * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(baseB + (offB[i] + koff)), (lds_void_t*)(Bs + (wid * 4 + i) * 1024), 16, 0, 0);
  };
  v16f acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int sw = (l31 >> 1) & 7;
  const int fa = l31 * 128, fb = OPA + (wid * 32 + l31) * 128;
  if (kt_lo < nkt) stage(0, kt_lo);
  __syncthreads();
  for (int64_t kt = kt_lo; kt < nkt; ++kt) {
    const int buf = (int)((kt - kt_lo) & 1);
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const unsigned char* S = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {  // 16 k per step: lane half h takes chunk 2 kk + h (hi) and 4 + 2 kk + h (lo)
      const int sh = ((2 * kk + h) ^ sw) << 4, sl = ((4 + 2 * kk + h) ^ sw) << 4;
      const h16x8 bh = *reinterpret_cast<const h16x8*>(S + fb + sh), bl = *reinterpret_cast<const h16x8*>(S + fb + sl);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const h16x8 ah = *reinterpret_cast<const h16x8*>(S + fa + i * 4096 + sh), al = *reinterpret_cast<const h16x8*>(S + fa + i * 4096 + sl);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D layout of a 32 x 32 tile: column = lane & 31 (row of B = output column), rows (e & 3) + 8 (e >> 2) + 4 h
  const float alpha = a.post / (a.sA[0] * a.sB[0]);
  const int64_t col = n0 + wid * 32 + l31;
  if (col < a.N) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.M) C[row * a.ldc + col] = acc[i][e] * (a.rowscale ? alpha * a.rowscale[row] : alpha);
      }
  }
}

__global__ __launch_bounds__(256) void k_split_image_pair_scaled(const float* __restrict__ src1, const float* __restrict__ src2, int64_t rows,
                                                                 int64_t K, int64_t ld, int64_t Kp, const float* __restrict__ scale,
                                                                 _Float16* __restrict__ dst1, _Float16* __restrict__ dst2) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  const float sc = scale[0];
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float x1 = k < K ? src1[r * ld + k] * sc : 0.f, x2 = k < K ? src2[r * ld + k] * sc : 0.f;
    const _Float16 h1 = (_Float16)x1, h2 = (_Float16)x2;
    const int64_t o = r * 2 * Kp + (k >> 5) * 64 + (k & 31);
    dst1[o] = h1;
    dst1[o + 32] = (_Float16)(x1 - (float)h1);
    dst2[o] = h2;
    dst2[o + 32] = (_Float16)(x2 - (float)h2);
  }
}

// Two scales for an operand pair whose columns alternate between two kinds in blocks of `half` (the trailing update of the band
// reduction: src1 = [V | Z | ...], src2 = [-Z | -V | ...], half = 64): V has entries up to 1 whatever the matrix, Z scales with
// its norm, and under ONE scale the smaller kind loses bits to the fp16 subnormal spacing once the two differ by more than
// ~2^10. Kind 0 of src1 and kind 1 of src2 (the V columns) take scale[0], the others scale[2]; every term of P Q' then carries
// the same factor scale[0] * scale[2], which is what gemm_split_update divides by when given the two as sP and sQ.
// scale_dev: [0] scale of kind 0, [1] its maximum (bits), [2] scale of kind 1, [3] its maximum (bits).
__global__ __launch_bounds__(256) void k_absmax_bits2(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld, int half,
                                                      unsigned* __restrict__ sd) {
  float mx0 = 0.f, mx1 = 0.f;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const float v = fabsf(src[r * ld + k]);
      if ((k / half) & 1) mx1 = fmaxf(mx1, v);
      else mx0 = fmaxf(mx0, v);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx0 = fmaxf(mx0, __shfl_xor(mx0, o));
    mx1 = fmaxf(mx1, __shfl_xor(mx1, o));
  }
  __shared__ float wmx[8];
  if ((threadIdx.x & 63) == 0) {
    wmx[threadIdx.x >> 6] = mx0;
    wmx[4 + (threadIdx.x >> 6)] = mx1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(sd + 1, __float_as_uint(fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]))));
    atomicMax(sd + 3, __float_as_uint(fmaxf(fmaxf(wmx[4], wmx[5]), fmaxf(wmx[6], wmx[7]))));
  }
}
__global__ void k_pick_scale2(float* __restrict__ sd) {
  for (int c = 0; c < 2; ++c) {
    const float mx = sd[2 * c + 1];  // the bits of a non-negative float, written through the unsigned view
    int ex = 0;
    float sc = 1.f;
    if (mx > 0.f && mx < 1.0e30f) {
      (void)frexpf(mx, &ex);
      sc = ldexpf(1.f, 14 - ex);
    }
    sd[2 * c] = sc;
  }
}
__global__ __launch_bounds__(256) void k_split_image_pair_scaled2(const float* __restrict__ src1, const float* __restrict__ src2, int64_t rows,
                                                                  int64_t K, int64_t ld, int64_t Kp, int half, const float* __restrict__ scale,
                                                                  _Float16* __restrict__ dst1, _Float16* __restrict__ dst2) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  const int kind = (int)((k / half) & 1);
  const float s1 = kind ? scale[2] : scale[0], s2 = kind ? scale[0] : scale[2];
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float x1 = k < K ? src1[r * ld + k] * s1 : 0.f, x2 = k < K ? src2[r * ld + k] * s2 : 0.f;
    const _Float16 h1 = (_Float16)x1, h2 = (_Float16)x2;
    const int64_t o = r * 2 * Kp + (k >> 5) * 64 + (k & 31);
    dst1[o] = h1;
    dst1[o + 32] = (_Float16)(x1 - (float)h1);
    dst2[o] = h2;
    dst2[o + 32] = (_Float16)(x2 - (float)h2);
  }
}

// The same pair of images when the largest entries are known beforehand: kind 0 (the reflector columns of the band reduction, entries
// at most 1) under the FIXED scale 2^13, kind 1 under the scale that brings max(zmax[0], zmax[1]) -- the bits of non-negative floats,
// written by the kernels that produced the columns -- into [2^13, 2^14). No pass over the operands for their maxima, no memset, no
// scale kernel: every thread derives the two scales itself, thread 0 of block 0 leaves them in scale[0], scale[2] for the product.
__global__ __launch_bounds__(256) void k_split_image_pair_zmax(const float* __restrict__ src1, const float* __restrict__ src2, int64_t rows,
                                                               int64_t K, int64_t ld, int64_t Kp, int half, const unsigned* __restrict__ zmax,
                                                               int nz, float* __restrict__ scale, _Float16* __restrict__ dst1,
                                                               _Float16* __restrict__ dst2) {
  float mx = __uint_as_float(zmax[0]);
  if (nz > 1) mx = fmaxf(mx, __uint_as_float(zmax[1]));
  int ex = 0;
  float sz = 1.f;
  if (mx > 0.f && mx < 1.0e30f) {
    (void)frexpf(mx, &ex);
    // the exponent clamped as in sbr_w_scale: sz, 1 / (sv sz) and C / alpha of the update that follows stay normal numbers even when
    // the Z columns of a nearly deflated trailing matrix are tiny (entries that small then lose their low piece, an absolute error far
    // below the rounding of C)
    int e = 14 - ex;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    sz = ldexpf(1.f, e);
  }
  const float sv = 8192.f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    scale[0] = sv;
    scale[2] = sz;
  }
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  const int kind = (int)((k / half) & 1);
  const float s1 = kind ? sz : sv, s2 = kind ? sv : sz;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float x1 = k < K ? src1[r * ld + k] * s1 : 0.f, x2 = k < K ? src2[r * ld + k] * s2 : 0.f;
    const _Float16 h1 = (_Float16)x1, h2 = (_Float16)x2;
    const int64_t o = r * 2 * Kp + (k >> 5) * 64 + (k & 31);
    dst1[o] = h1;
    dst1[o + 32] = (_Float16)(x1 - (float)h1);
    dst2[o] = h2;
    dst2[o + 32] = (_Float16)(x2 - (float)h2);
  }
}

struct SplitUpdArgs {
  const _Float16* A;  // split image, M rows
  const _Float16* B;  // split image, N rows
  const float* sA;    // their scales (device)
  const float* sB;
  int64_t M, N, Kp;
  float* C;
  int64_t ldc;
  int lower;
  int tiles_n;
  float post;          // C += post * P Q'
  const int2* tiles;   // tile list (compact squares per XCD, gemm.hip) for long contractions, or nullptr: decoded from the block index
  // split over K inside one launch (gridDim.y slices of kt_chunk 32-deep steps; slice s writes C + s * c_split_off), and
  // overwrite != 0: C = post * P Q' (C is not read) -- the split-K partials of a product with few output tiles
  int64_t kt_chunk;
  int64_t c_split_off;
  int overwrite;
  // acc_init != 0 (|post| == 1, no overwrite): the accumulators start from C / alpha (alpha a power of two: exact), every element of
  // the C tile requested before the first operand stage -- one memory round trip per tile with 256 KB in flight per CU instead of
  // eight of 32 KB in the epilogue, which is what bounded the rank-256 updates of the band reduction (65 us per tile, 39 at the fair
  // share of HBM)
  int acc_init;
  // AF32 instantiation: the first operand is an fp32 matrix (row pitch ldaf, rows K-contiguous) instead of a split image; its tile
  // is staged as it is and every lane splits the eight k of its fragment in registers, scaled by af_scale (a power of two)
  const float* Af;
  int64_t ldaf;
  float af_scale;
};

// x = hi + lo (fp16 pieces) of eight scaled fp32 values: two packed conversions and one mixed-precision fma per element
__device__ __forceinline__ void split_pk8(gb_f32x4 x0, gb_f32x4 x1, h16x8& hi, h16x8& lo) {
  gb_f16x2 h[4], l[4];
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = __builtin_convertvector(gb_f32x2{x[2 * q], x[2 * q + 1]}, gb_f16x2);
    const unsigned u = __builtin_bit_cast(unsigned, h[q]);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(u), "v"(x[2 * q]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(u), "v"(x[2 * q + 1]));
    l[q] = __builtin_convertvector(gb_f32x2{r0, r1}, gb_f16x2);
  }
  hi = h16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
  lo = h16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
}

// The main loop of corr_split_kernel with the tile decode, the accumulator start from C and the lower + mirror epilogue of
// gemm_nt_big<2, 4, 4, 2> (same 256 x 256 tile, same 32 x 32 accumulator layout).
// (Measured and removed in round 5: operands in HALF stages of 16 of K, four LDS buffers, three half stages in flight behind counted
// waits -- dense Gram 166 against 155 ms, first back-transformation 108 against 100 ms, profiles/r04_split_deep.log.)
// PIPE (image x image products only): the stage loop as the software pipeline of split_mainloop_pipe
template <bool AF32, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void gemm_split_kernel(SplitUpdArgs a) {
  static_assert(!(AF32 && PIPE), "the pipelined loop reads split images on both sides");
  constexpr int TM = 4, TN = 2;
  constexpr int OPB = 256 * 128, STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int2 tl;
  if (a.tiles) {
    tl = a.tiles[blockIdx.x];
    if (tl.x < 0) return;
  } else {
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    if (a.lower) {
      int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((unsigned)(ti + 1) * (unsigned)(ti + 2) / 2 <= t) ++ti;
      while ((unsigned)ti * (unsigned)(ti + 1) / 2 > t) --ti;
      tl = make_int2(ti, (int)(t - (unsigned)ti * (unsigned)(ti + 1) / 2));
    } else {
      tl = make_int2((int)(t / (unsigned)a.tiles_n), (int)(t % (unsigned)a.tiles_n));
    }
  }
  tl.x = __builtin_amdgcn_readfirstlane(tl.x);  // uniform by construction: tile corner, C pointers and the buffer resource in scalars
  tl.y = __builtin_amdgcn_readfirstlane(tl.y);
  const int64_t m0 = (int64_t)tl.x * 256, n0 = (int64_t)tl.y * 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3, l31 = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, sq = lane & 7;
  const _Float16* srcA[4];
  const float* srcAf[4];
  const _Float16* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    const int chunk = sq ^ ((r >> 1) & 7);
    int64_t ra = m0 + r, rb = n0 + r;
    if (ra > a.M - 1) ra = a.M - 1;
    if (rb > a.N - 1) rb = a.N - 1;
    srcA[i] = a.A + ra * 2 * a.Kp + 8 * chunk;
    srcAf[i] = AF32 ? a.Af + ra * a.ldaf + 4 * chunk : nullptr;  // 128 bytes per row and 32 of K either way
    srcB[i] = a.B + rb * 2 * a.Kp + 8 * chunk;
  }
  int64_t kt_lo = 0, nkt = a.Kp / 32;
  if (gridDim.y > 1) {  // this slice of the contraction
    kt_lo = (int64_t)blockIdx.y * a.kt_chunk;
    const int64_t kt_hi = kt_lo + a.kt_chunk;
    nkt = kt_hi < nkt ? kt_hi : nkt;
    if (kt_lo > nkt) kt_lo = nkt;
    a.C += (int64_t)blockIdx.y * a.c_split_off;
  }
  auto stage = [&](int buf, int64_t kt) {
    unsigned char* As = lds + buf * STAGE;
    unsigned char* Bs = As + OPB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AF32) __builtin_amdgcn_global_load_lds((glb_void_t*)(srcAf[i] + kt * 32), (lds_void_t*)(As + (wid * 4 + i) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[i] + kt * 64), (lds_void_t*)(As + (wid * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[i] + kt * 64), (lds_void_t*)(Bs + (wid * 4 + i) * 1024), 16, 0, 0);
  };
  if (!PIPE && kt_lo < nkt) stage((int)(kt_lo & 1), kt_lo);
  const float alpha = a.post / ((AF32 ? a.af_scale : a.sA[0]) * a.sB[0]);  // the scales are powers of two
  // 32-bit indices relative to the tile's corner (and to the corner of its mirror image): the 64-bit row * ldc + col of every
  // element cost this kernel 191 spilled registers
  float* Ct = a.C + m0 * a.ldc + n0;
  float* Cm = a.C + n0 * a.ldc + m0;
  const int ldc = (int)a.ldc;
  const int mrem = (int)((a.M - m0 < 256) ? a.M - m0 : 256), nrem = (int)((a.N - n0 < 256) ? a.N - n0 : 256);
  const int diag = a.lower ? (int)(m0 - n0) : 0x40000000;  // lower: element (r, c) of the tile is kept for c <= r + diag
  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const bool cinit = a.acc_init != 0 && !a.overwrite;
  if (cinit) {
    // buffer loads: the lane's part of the address in one register, the row of the element in a scalar, no predicate -- rows beyond
    // the matrix are beyond the resource (zero), columns beyond it and the upper part of a diagonal tile are loaded and never stored
    const unsigned nrec = (unsigned)(((int64_t)(mrem - 1) * a.ldc + nrem) * 4);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, nrec, 0x00020000);
    const unsigned vo = (unsigned)((4 * h * ldc + wn * 64 + l31) * 4);
    const int wms = __builtin_amdgcn_readfirstlane(wm);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned so = (unsigned)((wms * 128 + i * 32 + (e & 3) + 8 * (e >> 2)) * ldc) * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, vo + (unsigned)(j * 128), so, 0));
      }
  }
  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * 128 + i * 32 + l31;
    offA[i] = r * 128;
    swA[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * 64 + j * 32 + l31;
    offB[j] = OPB + r * 128;
    swB[j] = (r >> 1) & 7;
  }
  if (cinit && !(PIPE && kt_lo < nkt)) {
    const float inv_alpha = 1.f / alpha;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] *= inv_alpha;
  }
  if (PIPE) {
    if (kt_lo < nkt) {
      const unsigned rowb = (unsigned)(4 * a.Kp);
      split_mainloop_pipe(acc, lds, reinterpret_cast<const unsigned char*>(a.A) + m0 * rowb, reinterpret_cast<const unsigned char*>(a.B) + n0 * rowb,
                          rowb, (int)((a.M - 1 - m0 < 255) ? a.M - 1 - m0 : 255), (int)((a.N - 1 - n0 < 255) ? a.N - 1 - n0 : 255),
                          (unsigned)kt_lo * 128u, nkt - kt_lo, cinit, 1.f / alpha);
    }
  } else {
  __syncthreads();
  for (int64_t kt = kt_lo; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const unsigned char* S = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = 2 * kk + h, cl = 4 + 2 * kk + h;
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (AF32) {  // the fp32 tile: k = 16 kk + 8 h .. + 7 of this row are the 16-byte chunks 4 kk + 2 h and the next one
          const int c0 = 4 * kk + 2 * h;
          const gb_f32x4 x0 = *reinterpret_cast<const gb_f32x4*>(S + offA[i] + ((c0 ^ swA[i]) << 4));
          const gb_f32x4 x1 = *reinterpret_cast<const gb_f32x4*>(S + offA[i] + (((c0 + 1) ^ swA[i]) << 4));
          split_pk8(x0 * a.af_scale, x1 * a.af_scale, ah[i], al[i]);
        } else {
          ah[i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((ch ^ swA[i]) << 4));
          al[i] = *reinterpret_cast<const h16x8*>(S + offA[i] + ((cl ^ swA[i]) << 4));
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((ch ^ swB[j]) << 4));
        bl[j] = *reinterpret_cast<const h16x8*>(S + offB[j] + ((cl ^ swB[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  }
  const bool vec_mirror = a.lower && (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = wn * (32 * TN) + j * 32 + l31;
      const int r0 = wm * (32 * TM) + i * 32 + 4 * h;
      // C is read per 32 x 32 tile here (all 16 loads before the first store): started from C, the accumulators + their 128
      // address registers did not fit 256 registers at two waves per SIMD (191 spilled), and with 5 us of matrix work per tile
      // there is nothing to hide the loads behind anyway
      float cin[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = r0 + (e & 3) + 8 * (e >> 2);
        cin[e] = (!cinit && !a.overwrite && r < mrem && c < nrem && c <= r + diag) ? Ct[r * ldc + c] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(acc[i][j][e], alpha, cin[e]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = r0 + (e & 3) + 8 * (e >> 2);
        if (r < mrem && c < nrem && c <= r + diag) Ct[r * ldc + c] = acc[i][j][e];
      }
      if (a.lower && c < nrem) {  // mirror image: element (c, r) of the matrix for the strictly lower part
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rq = r0 + 8 * q;
          if (vec_mirror && rq + 3 < mrem && c < rq + diag) {
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            f32x4_ v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4_*>(&Cm[c * ldc + rq]) = v;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (rq + u < mrem && c < rq + u + diag) Cm[c * ldc + rq + u] = acc[i][j][4 * q + u];
          }
        }
      }
    }
}

}  // namespace

size_t gram_binary_scratch_bytes(int64_t N, int64_t M) { return sizeof(unsigned short) * (size_t)M * (size_t)round_up(N, 64); }

// sh != nullptr (row-sharded session): p holds this rank's cells; the per-gene statistics are global (all-reduced inside
// scale_stats_sharded), the cell weights, the co-occurrence product, u_j = d_j sum_i P_ij s_i^2 l_i and S2 = sum s_i^2 run over
// the local cells and `N` in the -N cent_j cent_k term is the local count, so that A receives this rank's additive part of
// B'B / divisor (the caller sums the parts over the ranks).
int gram_binary(Ctx* ctx, const PatternDev& p, const float* val, int f32path, void* scratch, float divisor, float* A, int64_t lda,
                const ShardReduce* sh) {
  return gram_binary_stats(ctx, p, val, f32path, nullptr, scratch, divisor, A, lda, sh, false);
}
// given != nullptr: the statistics are the caller's (a chunk of cells of a chunked session: per-gene vectors of ALL cells, per-cell
// vectors of the chunk, srow = 1 / l and cent = 0 -- the chunk's U'U, see scale.hip "chunked variant"); accumulate: A += the product
int gram_binary_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const ScaleStats* given, void* scratch, float divisor,
                      float* A, int64_t lda, const ShardReduce* sh, bool accumulate) {
  const int64_t N = p.N, M = p.M, ldm = round_up(N, 64);
  const int nw = ctx->opt.eff_gram_bits_terms();  // fp16 pieces of the cell weights: 22 bits (default) or 33 (always with precision = 0)
  hipStream_t st = ctx->stream;
  unsigned short* Pm = static_cast<unsigned short*>(scratch);
  const int64_t nparts = (N + 255) / 256;
  SCL_WS(ctx, w, double, "gb.w", N);
  SCL_WS(ctx, sa, double, "gb.sa", N);
  SCL_WS(ctx, wpart, double, "gb.wpart", nparts);
  SCL_WS(ctx, sc, double, "gb.sc", 4);
  SCL_WS(ctx, wq, _Float16, "gb.wq", ldm * 3);
  SCL_WS(ctx, gv, double4, "gb.gv", M);
  {
    StageTimer tm(ctx, "scale");
    ScaleStats ss;
    if (given) ss = *given;
    else if (sh && sh->on()) SCL_TRY(scale_stats_sharded(ctx, p, val, f32path, *sh, &ss));
    else SCL_TRY(scale_stats(ctx, p, val, f32path, 0, &ss));
    hipLaunchKernelGGL(k_cell_weights, dim3((unsigned)nparts), dim3(256), 0, st, N, ss.tgc, ss.srow, f32path, w, sa, wpart);
    hipLaunchKernelGGL(k_weight_scale, dim3(1), dim3(1024), 0, st, wpart, nparts, ss.srow, N, sc);
    hipLaunchKernelGGL(k_split_weights, dim3((unsigned)((ldm + 255) / 256)), dim3(256), 0, st, w, N, ldm, nw, sc, wq);
    hipLaunchKernelGGL(k_gene_vecs, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, val, sa, ss.stdv, ss.mu, ss.cent, gv);
    if (ctx->opt.dense_fused != 0 && (reinterpret_cast<uintptr_t>(Pm) & 15u) == 0) {
      hipLaunchKernelGGL(k_mask_fused, dim3((unsigned)M), dim3(256), 0, st, p, val, Pm, ldm);
    } else {  // (same image)
      SCL_HIP(ctx, hipMemsetAsync(Pm, 0, gram_binary_scratch_bytes(N, M), st));
      hipLaunchKernelGGL(k_mask_scatter, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, val, Pm, ldm);
    }
    SCL_HIP(ctx, hipGetLastError());
  }
  StageTimer tm(ctx, "gram");
  if (!accumulate) SCL_HIP(ctx, hipMemsetAsync(A, 0, sizeof(float) * (size_t)M * lda, st));
  const int64_t bm = (M + 255) / 256;
  const int2* tiles = nullptr;
  int64_t nb = 0;
  SCL_TRY(big_tile_list(ctx, bm, bm, 1, &tiles, &nb));
  GramBitsArgs a{Pm, wq, gv, sc, A, M, ldm, lda, (double)N, 1.0 / (double)divisor, accumulate ? 1 : 0};
  return nw == 3 ? launch_gram_bits<3>(ctx, a, tiles, nb) : launch_gram_bits<2>(ctx, a, tiles, nb);
}

size_t split_image_bytes(int64_t rows, int64_t K) { return sizeof(_Float16) * (size_t)rows * 2 * (size_t)round_up(K, 32); }

int split_image_f16(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst) {
  if (rows <= 0) return SCLENS_OK;
  const int64_t Kp = round_up(K, 32);
  hipLaunchKernelGGL(k_split_image, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0,
                     ctx->stream, src, rows, K, ld, Kp, static_cast<_Float16*>(dst));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_scaled(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* scale_dev) {
  if (rows <= 0) return SCLENS_OK;
  const int64_t Kp = round_up(K, 32);
  hipStream_t st = ctx->stream;
  unsigned* mx = reinterpret_cast<unsigned*>(scale_dev + 1);
  SCL_HIP(ctx, hipMemsetAsync(mx, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(k_absmax_bits, dim3((unsigned)std::min<int64_t>(rows, 256)), dim3(256), 0, st, src, rows, K, ld, mx);
  hipLaunchKernelGGL(k_pick_scale, dim3(1), dim3(1), 0, st, mx, scale_dev);
  hipLaunchKernelGGL(k_split_image_scaled, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0, st,
                     src, rows, K, ld, Kp, scale_dev, static_cast<_Float16*>(dst));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_pair_scaled(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, void* dst1, void* dst2,
                            float* scale_dev) {
  if (rows <= 0) return SCLENS_OK;
  const int64_t Kp = round_up(K, 32);
  hipStream_t st = ctx->stream;
  unsigned* mx = reinterpret_cast<unsigned*>(scale_dev + 1);
  SCL_HIP(ctx, hipMemsetAsync(mx, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(k_absmax_bits, dim3((unsigned)std::min<int64_t>(rows, 256)), dim3(256), 0, st, src1, rows, K, ld, mx);
  hipLaunchKernelGGL(k_pick_scale, dim3(1), dim3(1), 0, st, mx, scale_dev);
  hipLaunchKernelGGL(k_split_image_pair_scaled, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0,
                     st, src1, src2, rows, K, ld, Kp, scale_dev, static_cast<_Float16*>(dst1), static_cast<_Float16*>(dst2));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_pair_scaled2(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, int half, void* dst1,
                             void* dst2, float* scale_dev) {
  if (rows <= 0) return SCLENS_OK;
  if (half <= 0 || K % (2 * half) != 0) return ctx->fail(SCLENS_ERR_ARG, "split_image_pair_scaled2: K must be a multiple of 2 * half");
  const int64_t Kp = round_up(K, 32);
  hipStream_t st = ctx->stream;
  SCL_HIP(ctx, hipMemsetAsync(scale_dev, 0, 4 * sizeof(float), st));
  hipLaunchKernelGGL(k_absmax_bits2, dim3((unsigned)std::min<int64_t>(rows, 256)), dim3(256), 0, st, src1, rows, K, ld, half,
                     reinterpret_cast<unsigned*>(scale_dev));
  hipLaunchKernelGGL(k_pick_scale2, dim3(1), dim3(1), 0, st, scale_dev);
  hipLaunchKernelGGL(k_split_image_pair_scaled2, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0,
                     st, src1, src2, rows, K, ld, Kp, half, scale_dev, static_cast<_Float16*>(dst1), static_cast<_Float16*>(dst2));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_pair_zmax(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, int half, void* dst1,
                          void* dst2, float* scale_dev, const unsigned* zmax_dev, int nz) {
  if (rows <= 0) return SCLENS_OK;
  if (half <= 0 || K % (2 * half) != 0 || nz < 1 || nz > 2) return ctx->fail(SCLENS_ERR_ARG, "split_image_pair_zmax: bad arguments");
  const int64_t Kp = round_up(K, 32);
  hipLaunchKernelGGL(k_split_image_pair_zmax, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0,
                     ctx->stream, src1, src2, rows, K, ld, Kp, half, zmax_dev, nz, scale_dev, static_cast<_Float16*>(dst1),
                     static_cast<_Float16*>(dst2));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

template <bool AF32>
static int launch_split_kernel(Ctx* ctx, const SplitUpdArgs& a, dim3 grid) {
  constexpr int LDS_BYTES = 2 * 2 * 256 * 128;
  if (!AF32 && ctx->opt.split_pipe != 0) {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm_split_kernel<false, true>), LDS_BYTES));
    hipLaunchKernelGGL((gemm_split_kernel<false, true>), grid, dim3(512), LDS_BYTES, ctx->stream, a);
  } else {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm_split_kernel<AF32, false>), LDS_BYTES));
    hipLaunchKernelGGL((gemm_split_kernel<AF32, false>), grid, dim3(512), LDS_BYTES, ctx->stream, a);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int gemm_split_update(Ctx* ctx, const void* Pimg, const float* sP, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K,
                      float* C, int64_t ldc, int lower, float post) {
  if (M <= 0 || N <= 0) return SCLENS_OK;
  if (lower && M != N) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_update: lower needs M == N");
  const int64_t bm = (M + 255) / 256, bn = (N + 255) / 256;
  int64_t nb = lower ? bm * (bm + 1) / 2 : bm * bn;
  if (nb > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_update: too many tiles");
  const int2* tiles = nullptr;
  if (K >= 2048 && nb >= 1500) SCL_TRY(big_tile_list(ctx, bm, bn, lower, &tiles, &nb));  // operand panels re-used out of the L2s
  const bool no_acc_init = ctx->opt.split_acc_init == 0;  // C added in the epilogue (tests compare the two forms)
  SplitUpdArgs a{static_cast<const _Float16*>(Pimg), static_cast<const _Float16*>(Qimg), sP, sQ, M, N, round_up(K, 32), C, ldc, lower,
                 (int)bn, post, tiles, 0, 0, 0, (!no_acc_init && fabsf(post) == 1.f) ? 1 : 0, nullptr, 0, 1.f};
  return launch_split_kernel<false>(ctx, a, dim3((unsigned)nb));
}

// C_s[M][N] = P Q' over the K-slice s (s < splits, k_chunk a multiple of 32), C_s at C + s * c_split_off with row pitch ldc: the
// split-K partials of a product with few output tiles on the fp16 matrix cores (the consumer sums the slabs in a fixed order)
int gemm_split_nt(Ctx* ctx, const void* Pimg, const float* sP, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K, float* C,
                  int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off) {
  if (M <= 0 || N <= 0) return SCLENS_OK;
  if (splits < 1 || (splits > 1 && k_chunk % 32 != 0)) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_nt: k_chunk must be a multiple of 32");
  const int64_t bm = (M + 255) / 256, bn = (N + 255) / 256, nb = bm * bn;
  if (nb > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_nt: too many tiles");
  SplitUpdArgs a{static_cast<const _Float16*>(Pimg), static_cast<const _Float16*>(Qimg), sP, sQ, M, N, round_up(K, 32), C, ldc, 0,
                 (int)bn, 1.0f, nullptr, splits > 1 ? k_chunk / 32 : round_up(K, 32) / 32, c_split_off, 1, 0, nullptr, 0, 1.f};
  return launch_split_kernel<false>(ctx, a, dim3((unsigned)nb, (unsigned)splits));
}

// the same with the first operand taken as it is: P [M][K] fp32 (row pitch ldp, K a multiple of 32), scaled by `p_scale` (a power of
// two that brings its entries into fp16 range) and split in registers by the kernel -- no image pass over P (the first
// back-transformation's W1 = Z Vm': the image of Z cost a read and a write of the vector block per group, profiles/r04_cfg4_kernel_stats.csv)
int gemm_split_nt_f32a(Ctx* ctx, const float* P, int64_t ldp, float p_scale, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K,
                       float* C, int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off) {
  if (M <= 0 || N <= 0) return SCLENS_OK;
  if (K % 32 != 0 || ldp % 4 != 0 || (reinterpret_cast<uintptr_t>(P) & 15u))
    return ctx->fail(SCLENS_ERR_ARG, "gemm_split_nt_f32a: K must be a multiple of 32 and P 16-byte aligned with a pitch that is a multiple of 4");
  if (splits < 1 || (splits > 1 && k_chunk % 32 != 0)) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_nt_f32a: k_chunk must be a multiple of 32");
  const int64_t bm = (M + 255) / 256, bn = (N + 255) / 256, nb = bm * bn;
  if (nb > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_nt_f32a: too many tiles");
  SplitUpdArgs a{nullptr, static_cast<const _Float16*>(Qimg), nullptr, sQ, M, N, K, C, ldc, 0,
                 (int)bn, 1.0f, nullptr, splits > 1 ? k_chunk / 32 : K / 32, c_split_off, 1, 0, P, ldp, p_scale};
  return launch_split_kernel<true>(ctx, a, dim3((unsigned)nb, (unsigned)splits));
}

// the split image of src [rows][K] under a FIXED power-of-two scale (operands whose entries are known to be at most 1 in magnitude:
// rows of orthogonal matrices, reflector blocks): no pass over the data for its largest entry. scale_dev[0] receives the scale.
int split_image_fixed(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* scale_dev, float scale) {
  if (rows <= 0) return SCLENS_OK;
  const int64_t Kp = round_up(K, 32);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_set_scale, dim3(1), dim3(1), 0, st, scale_dev, scale);
  hipLaunchKernelGGL(k_split_image_scaled, dim3((unsigned)((Kp + 255) / 256), (unsigned)std::min<int64_t>(rows, 65535)), dim3(256), 0, st,
                     src, rows, K, ld, Kp, scale_dev, static_cast<_Float16*>(dst));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_rows(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* inv_scale_dev) {
  if (rows <= 0) return SCLENS_OK;
  if (rows > 65535) return ctx->fail(SCLENS_ERR_ARG, "split_image_rows: too many rows");
  const int64_t Kp = round_up(K, 32);
  hipStream_t st = ctx->stream;
  SCL_WS(ctx, mx, unsigned, "gram.rowmax", rows);
  SCL_HIP(ctx, hipMemsetAsync(mx, 0, sizeof(unsigned) * (size_t)rows, st));
  hipLaunchKernelGGL(k_row_absmax_bits, dim3((unsigned)((K + 4095) / 4096), (unsigned)rows), dim3(256), 0, st, src, K, ld, mx);
  hipLaunchKernelGGL(k_split_image_rows, dim3((unsigned)((Kp + 255) / 256), (unsigned)rows), dim3(256), 0, st, src, K, ld, Kp, mx,
                     static_cast<_Float16*>(dst), inv_scale_dev);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int split_image_transposed(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, const float* scale_dev) {
  if (rows <= 0 || K <= 0) return SCLENS_OK;
  const int64_t gy = (round_up(rows, 32)) / 32;
  if (gy > 65535) return ctx->fail(SCLENS_ERR_ARG, "split_image_transposed: too many rows");
  hipLaunchKernelGGL(k_split_image_transposed, dim3((unsigned)((K + 63) / 64), (unsigned)gy), dim3(256), 0, ctx->stream, src, rows, K, ld,
                     round_up(rows, 32), scale_dev, static_cast<_Float16*>(dst));
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

int gemm_split_skinny(Ctx* ctx, const void* Aimg, const float* sA, const float* rowscale, int64_t M, const void* Bimg, const float* sB, int64_t N,
                      int64_t K, float* C, int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off, float post) {
  if (M <= 0 || N <= 0) return SCLENS_OK;
  if (M > 64) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_skinny: at most 64 rows");
  if (splits < 1 || (splits > 1 && k_chunk % 32 != 0)) return ctx->fail(SCLENS_ERR_ARG, "gemm_split_skinny: k_chunk must be a multiple of 32");
  const int64_t Kp = round_up(K, 32), bn = (N + 255) / 256;
  constexpr int LDS_BYTES = 2 * (64 * 128 + 256 * 128);
  SplitSkinnyArgs a{static_cast<const _Float16*>(Aimg), static_cast<const _Float16*>(Bimg), sA, sB, rowscale, M, N, Kp, C, ldc,
                    splits > 1 ? k_chunk / 32 : Kp / 32, c_split_off, post};
  SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm_split_skinny_kernel), LDS_BYTES));
  hipLaunchKernelGGL(gemm_split_skinny_kernel, dim3((unsigned)bn, (unsigned)splits), dim3(512), LDS_BYTES, ctx->stream, a);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// colabsmax[j] = max(colabsmax[j], max_i |A_i . B_j|) for the split images of A [M][K] and B [N][K]
int corr_colabsmax_split(Ctx* ctx, const void* Aimg, int64_t M, const void* Bimg, int64_t N, int64_t K, unsigned* colabsmax) {
  if (M <= 0 || N <= 0) return SCLENS_OK;
  const int64_t bm = (M + 255) / 256, bn = (N + 255) / 256;
  const int2* tiles = nullptr;
  int64_t nb = 0;
  SCL_TRY(big_tile_list(ctx, bm, bn, 0, &tiles, &nb));
  constexpr int LDS_BYTES = 2 * 2 * 256 * 128;
  SplitCorrArgs a{static_cast<const _Float16*>(Aimg), static_cast<const _Float16*>(Bimg), M, N, round_up(K, 32), colabsmax,
                  1.0f / 16777216.0f};
  if (ctx->opt.split_pipe != 0) {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(corr_split_kernel_p), LDS_BYTES));
    hipLaunchKernelGGL(corr_split_kernel_p, dim3((unsigned)nb), dim3(512), LDS_BYTES, ctx->stream, a, tiles);
  } else {
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(corr_split_kernel), LDS_BYTES));
    hipLaunchKernelGGL(corr_split_kernel, dim3((unsigned)nb), dim3(512), LDS_BYTES, ctx->stream, a, tiles);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
