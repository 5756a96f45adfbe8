// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-staged, used for every dense contraction
// of the path: the Gram ("Wishart") matrix (scLENS.jl:332-361), the rank-2k trailing updates and
// block-reflector applications of the eigensolver (scLENS.jl:375-387 -> cuSOLVER/LAPACK there),
// corr_mat (scLENS.jl:363-373, fused |.|-column-max of :742), the cell-side recovery GEMM
// (scLENS.jl:503-508, :556-558) and the gene basis (scLENS.jl:813-818).
//
// Block tile 128x128x16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// LDS image is [k][m] for both operands (row stride 132 floats), so a fragment read is 32
// consecutive floats per half-wave (conflict-free ds_read_b32); global->LDS goes through registers
// (prefetch of tile t+1 is issued before the MFMAs of tile t; one barrier per K-step).
#include <algorithm>

#include "common.h"

namespace scl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;

template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ base, int64_t ld, int64_t r,
                                     int64_t nr, int64_t c, int64_t nc) {
  // 4 consecutive elements (r, c..c+3) of a row-major matrix with nr x nc valid entries; 0 outside
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < nr && c < nc) {
    const float* p = base + r * ld + c;
    if (VEC && c + 3 < nc) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
      v[0] = p[0];
      if (c + 1 < nc) v[1] = p[1];
      if (c + 2 < nc) v[2] = p[2];
      if (c + 3 < nc) v[3] = p[3];
    }
  }
  return v;
}

// WM = 2: block tile 128 x 128 (waves 2 x 2); WM = 1: 64 x 256 (waves 1 x 4) for products with at most 64 rows (the block
// products of the subspace iteration: a 128-row tile would spend half of its MFMAs on zero rows)
template <bool QKC, bool VEC, int WM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a, int tiles_n) {
  constexpr int BMt = 64 * WM, BNt = 256 / WM, LDA = BMt + 4, LDB = BNt + 4;
  constexpr int RA = BMt / 64, RB = BNt / 64;  // 64-row groups of the staged tiles
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

  // ---- block -> tile, XCD-aware: blocks that share blockIdx%8 (one XCD's L2) get a contiguous
  //      run of tile ids, so neighbouring tiles (same P row panel) hit the same L2.
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int ti, tj;
  if (a.lower) {
    ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((unsigned)(ti + 1) * (unsigned)(ti + 2) / 2 <= t) ++ti;
    while ((unsigned)ti * (unsigned)(ti + 1) / 2 > t) --ti;
    tj = (int)(t - (unsigned)ti * (unsigned)(ti + 1) / 2);
  } else {
    ti = (int)(t / (unsigned)tiles_n);
    tj = (int)(t % (unsigned)tiles_n);
  }
  const int64_t m0 = (int64_t)ti * BMt, n0 = (int64_t)tj * BNt;
  if (a.splits > 1) {  // this block's K slice and private output (grid.y = slice)
    const int64_t koff = (int64_t)blockIdx.y * a.k_chunk;
    const int64_t kleft = a.K - koff;
    a.K = kleft < 0 ? 0 : (kleft < a.k_chunk ? kleft : a.k_chunk);
    a.P += koff;
    a.Q += QKC ? koff : koff * a.ldq;
    a.C += (int64_t)blockIdx.y * a.c_split_off;
  }

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = (WM == 2) ? (wid >> 1) : 0, wn = (WM == 2) ? (wid & 1) : wid, l31 = lane & 31, h = lane >> 5;

  // staging coordinates
  const int pr = tid >> 2, pkq = tid & 3;   // K-contiguous operand: rows pr + 64 g; k = 4*pkq..
  const int qk = tid >> 5, qnq = tid & 31;  // N-contiguous operand: k rows qk, qk+8; n = 4*qnq + 128 c..

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[RA], rb[RB];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int g = 0; g < RA; ++g) ra[g] = ld4<VEC>(a.P, a.ldp, m0 + pr + 64 * g, a.M, k0 + 4 * pkq, a.K);
    if (QKC) {
#pragma unroll
      for (int g = 0; g < RB; ++g) rb[g] = ld4<VEC>(a.Q, a.ldq, n0 + pr + 64 * g, a.N, k0 + 4 * pkq, a.K);
    } else {
#pragma unroll
      for (int c = 0; c < RB / 2; ++c) {
        rb[2 * c] = ld4<VEC>(a.Q, a.ldq, k0 + qk, a.K, n0 + 4 * qnq + 128 * c, a.N);
        rb[2 * c + 1] = ld4<VEC>(a.Q, a.ldq, k0 + qk + 8, a.K, n0 + 4 * qnq + 128 * c, a.N);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int g = 0; g < RA; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) As[buf][4 * pkq + i][pr + 64 * g] = ra[g][i];
    if (QKC) {
#pragma unroll
      for (int g = 0; g < RB; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[buf][4 * pkq + i][pr + 64 * g] = rb[g][i];
    } else {
#pragma unroll
      for (int c = 0; c < RB / 2; ++c) {
        *reinterpret_cast<f32x4*>(&Bs[buf][qk][4 * qnq + 128 * c]) = rb[2 * c];
        *reinterpret_cast<f32x4*>(&Bs[buf][qk + 8][4 * qnq + 128 * c]) = rb[2 * c + 1];
      }
    }
  };

  const int64_t nkt = (a.K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) gload((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[buf][kk + h][wm * 64 + l31];
      const float a1 = As[buf][kk + h][wm * 64 + 32 + l31];
      const float b0 = Bs[buf][kk + h][wn * 64 + l31];
      const float b1 = Bs[buf][kk + h][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue. D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (a.colabsmax) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(a.alpha * acc[i][j][e]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int64_t col = n0 + wn * 64 + j * 32 + l31;
      if (h == 0 && col < a.N) atomicMax(&a.colabsmax[col], __float_as_uint(mx));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.M && col < a.N) {
          if (a.lower) {
            if (col <= row) {
              float v = a.alpha * acc[i][j][e];
              if (a.beta != 0.f) v += a.beta * a.C[row * a.ldc + col];
              a.C[row * a.ldc + col] = v;
              if (col < row) a.C[col * a.ldc + row] = v;
            }
          } else {
            float v = a.alpha * acc[i][j][e];
            if (a.beta != 0.f) v += a.beta * a.C[row * a.ldc + col];
            a.C[row * a.ldc + col] = v;
          }
        }
      }
    }
}

// ---- large-tile NT kernel (both operands K-contiguous) ---------------------------------------------------------------------
// 256x256x32 block tile, 512 threads = 8 waves (2 x 4), each wave 128x64 = 4x2 MFMA tiles (128 accumulator registers, one
// workgroup per CU, two waves per SIMD). Operands go global -> LDS directly (`global_load_lds_dwordx4`, no VGPR staging), two
// LDS stages of 64 KB. One wave-instruction moves 8 rows x 128 B (full cache lines); the LDS image is lane-linear
// ([row][8 chunks of 16 B]) and chunk q of row r is stored at slot q ^ ((r >> 1) & 7) by permuting the SOURCE address, so
// that the fragment reads (`ds_read_b128`, 16-lane groups with 16 different rows) are bank-conflict free.
// k order inside an 8-deep group: MFMA step t takes k = t from the lanes 0-31 and k = 4 + t from the lanes 32-63 (lane half h
// reads chunk 2 j + h), so one 16-byte read per operand tile feeds four MFMA steps.
// The block -> tile map is a host-built list: blocks that share blockIdx % 8 (one XCD, one L2) walk through compact squares of
// tiles, so an XCD's concurrent tiles share operand panels in its L2.
constexpr int GB = 256, GK = 32;
// wave grid WM x WN, each wave TM x TN MFMA tiles: block tile BM = 32 WM TM = 256 rows by BN = 32 WN TN columns.
//   <2,4,4,2>: 256 x 256 (wave 128 x 64)  -- square products (Gram, rank-k updates, corr_mat)
//   <8,1,1,2>: 256 x 64  (wave 32 x 64)   -- skinny products with 64 columns (band reduction, back-transformations)
template <int WM, int WN, int TM, int TN>
struct BigCfg {
  static constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static constexpr int STAGE = (BM + BN) * GK;  // floats of one LDS stage (A image + B image)
  static constexpr int GA = BM / 8 / (WM * WN), GBq = BN / 8 / (WM * WN);  // 8-row staging groups per wave
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(512, 2) void gemm_nt_big(GemmArgs a, const int2* __restrict__ tiles, int tiles_n) {
  using Cfg = BigCfg<WM, WN, TM, TN>;
  static_assert(WM * WN == 8 && Cfg::BM == GB && Cfg::GA >= 1 && Cfg::GBq >= 1, "8 waves, 256 rows");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int2 tl;
  if (tiles) {
    tl = tiles[blockIdx.x];
  } else {  // no list (shapes that change from call to call): the XCD-contiguous run of tile ids of the 128x128 kernel
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    if (a.lower) {
      int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((unsigned)(ti + 1) * (unsigned)(ti + 2) / 2 <= t) ++ti;
      while ((unsigned)ti * (unsigned)(ti + 1) / 2 > t) --ti;
      tl = make_int2(ti, (int)(t - (unsigned)ti * (unsigned)(ti + 1) / 2));
    } else {
      tl = make_int2((int)(t / (unsigned)tiles_n), (int)(t % (unsigned)tiles_n));
    }
  }
  if (tl.x < 0) return;
  // Phase stagger (a.stagger_ns > 0): a product with a short contraction and a read-modify-write of C has three phases per tile
  // -- load C, multiply, store C -- and every workgroup has its CU to itself (128 KB of LDS), so all CUs start in step and STAY in
  // step (equal tiles): the whole chip loads, then the whole chip multiplies with HBM idle, then the whole chip stores; the
  // time per tile is the SUM of the HBM time and the MFMA time (rank-128 update at n' = 30 000: 77 us per tile against 27 us of
  // MFMA and 44 us of HBM at the fair share). Delaying the first workgroup of each CU by a different fraction of the tile period
  // spreads the phases, so that some CUs always use the memory system while the others multiply. Only the first wave of
  // workgroups waits (those that find an idle CU: blockIdx < 256); their successors inherit the offsets.
  if (a.stagger_ns > 0 && blockIdx.x < 256u && blockIdx.y == 0) {
    const unsigned slot = (blockIdx.x >> 3) & 31u;  // blocks b, b + 8, ... share an XCD: spread each XCD's 32 CUs
    const uint64_t wait_ticks = (uint64_t)a.stagger_ns * slot / 320u;  // s_memrealtime ticks at 100 MHz (10 ns); slot / 32 of the period
    if (threadIdx.x == 0) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  const int64_t m0 = (int64_t)tl.x * Cfg::BM, n0 = (int64_t)tl.y * Cfg::BN;
  if (a.splits > 1) {
    const int64_t koff = (int64_t)blockIdx.y * a.k_chunk;
    const int64_t kleft = a.K - koff;
    a.K = kleft < 0 ? 0 : (kleft < a.k_chunk ? kleft : a.k_chunk);
    a.P += koff;
    a.Q += koff;
    a.C += (int64_t)blockIdx.y * a.c_split_off;
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, l31 = lane & 31, h = lane >> 5;

  // staging: wave w moves the 8-row groups GA w .. GA w + GA - 1 of A and GBq w .. of B
  const int srow = lane >> 3, sq = lane & 7;
  const float* srcA[Cfg::GA];
  const float* srcB[Cfg::GBq];
  int64_t rowA[Cfg::GA], rowB[Cfg::GBq];
#pragma unroll
  for (int i = 0; i < Cfg::GA; ++i) {
    const int r = (wid * Cfg::GA + i) * 8 + srow;
    const int chunk = sq ^ ((r >> 1) & 7);
    int64_t ra = m0 + r;
    if (ra > a.M - 1) ra = a.M - 1;
    rowA[i] = ra;
    srcA[i] = a.P + ra * a.ldp + 4 * chunk;
  }
#pragma unroll
  for (int i = 0; i < Cfg::GBq; ++i) {
    const int r = (wid * Cfg::GBq + i) * 8 + srow;
    const int chunk = sq ^ ((r >> 1) & 7);
    int64_t rb = n0 + r;
    if (rb > a.N - 1) rb = a.N - 1;
    rowB[i] = rb;
    srcB[i] = a.Q + rb * a.ldq + 4 * chunk;
  }
  const int64_t nfull = a.K / GK, nkt = (a.K + GK - 1) / GK;
  auto stage = [&](int buf, int64_t kt) {
    float* As = lds + buf * Cfg::STAGE;
    float* Bs = As + Cfg::BM * GK;
    if (kt < nfull) {
#pragma unroll
      for (int i = 0; i < Cfg::GA; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(srcA[i] + kt * GK), (lds_void*)(As + (wid * Cfg::GA + i) * 256), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < Cfg::GBq; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(srcB[i] + kt * GK), (lds_void*)(Bs + (wid * Cfg::GBq + i) * 256), 16, 0, 0);
    } else {  // K tail: through registers, zero beyond K
#pragma unroll
      for (int i = 0; i < Cfg::GA; ++i) {
        const int g = wid * Cfg::GA + i, r = g * 8 + srow;
        const int chunk = sq ^ ((r >> 1) & 7);
        *reinterpret_cast<f32x4*>(As + g * 256 + lane * 4) = ld4<true>(a.P, a.ldp, rowA[i], a.M, kt * GK + 4 * chunk, a.K);
      }
#pragma unroll
      for (int i = 0; i < Cfg::GBq; ++i) {
        const int g = wid * Cfg::GBq + i, r = g * 8 + srow;
        const int chunk = sq ^ ((r >> 1) & 7);
        *reinterpret_cast<f32x4*>(Bs + g * 256 + lane * 4) = ld4<true>(a.Q, a.ldq, rowB[i], a.N, kt * GK + 4 * chunk, a.K);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // acc_init (alpha == beta == 1): C is loaded straight into the accumulators, all loads in flight together with the first
  // operand stage, so the epilogue is stores only. Without it every 32 x 32 tile of the epilogue pays a memory round trip of
  // its own (load 16, wait, store 16) with the matrix cores idle -- the workgroup is alone on its CU (128 KB of LDS).
  const bool acc_init = a.acc_init != 0;
  if (acc_init) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t col = n0 + wn * (32 * TN) + j * 32 + l31;
        const int64_t row0 = m0 + wm * (32 * TM) + i * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
          if (row < a.M && col < a.N && (!a.lower || col <= row)) acc[i][j][e] = a.C[row * a.ldc + col];
        }
      }
  }

  // fragment read offsets (floats) inside a stage: row r -> r * 32 + ((c ^ ((r >> 1) & 7)) << 2), c = 2 j8 + h
  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (32 * TM) + i * 32 + l31;
    offA[i] = r * 32;
    swA[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * (32 * TN) + j * 32 + l31;
    offB[j] = Cfg::BM * GK + r * 32;
    swB[j] = (r >> 1) & 7;
  }

  if (nkt > 0) stage(0, 0);
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const float* S = lds + buf * Cfg::STAGE;
    // fragments of the 8-deep group j8 + 1 are fetched before the MFMAs of group j8 (two named register sets)
    f32x4 av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(S + offA[i] + ((h ^ swA[i]) << 2));
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(S + offB[j] + ((h ^ swB[j]) << 2));
#pragma unroll
    for (int j8 = 0; j8 < 4; ++j8) {
      const int cur = j8 & 1, nxt = cur ^ 1;
      if (j8 + 1 < 4) {
        const int c = 2 * (j8 + 1) + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nxt][i] = *reinterpret_cast<const f32x4*>(S + offA[i] + ((c ^ swA[i]) << 2));
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nxt][j] = *reinterpret_cast<const f32x4*>(S + offB[j] + ((c ^ swB[j]) << 2));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of this group's MFMAs (hipcc would sink them behind)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][t], bv[cur][j][t], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue. D layout of the 32x32 MFMA: col = lane&31, rows (e&3) + 8*(e>>2) + 4*(lane>>5): four consecutive rows per
  //      register quad, so the mirrored (transposed) store of `lower` is one 16-byte store per quad.
  if (a.colabsmax) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(a.alpha * acc[i][j][e]));
      // rows beyond M hold copies of row M-1 (clamped loads): harmless for a maximum
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int64_t col = n0 + wn * (32 * TN) + j * 32 + l31;
      if (h == 0 && col < a.N) atomicMax(&a.colabsmax[col], __float_as_uint(mx));
    }
    return;
  }
  const bool vec_mirror = a.lower && (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * (32 * TN) + j * 32 + l31;
      const int64_t row0 = m0 + wm * (32 * TM) + i * 32 + 4 * h;
      // beta: all 16 loads of this 32 x 32 tile are issued before its first store (a store to C would otherwise fence every
      // later load of C: one memory round trip per element)
      f32x16 cin;
#pragma unroll
      for (int e = 0; e < 16; ++e) cin[e] = 0.f;
      if (a.beta != 0.f && !acc_init) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
          if (row < a.M && col < a.N && (!a.lower || col <= row)) cin[e] = a.C[row * a.ldc + col];
        }
      }
      f32x16 outv;
      if (acc_init) {
        outv = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) outv[e] = a.alpha * acc[i][j][e] + a.beta * cin[e];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
        if (row < a.M && col < a.N && (!a.lower || col <= row)) a.C[row * a.ldc + col] = outv[e];
      }
      if (a.lower && col < a.N) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t rowq = row0 + 8 * q;
          if (vec_mirror && rowq + 3 < a.M && col < rowq) {
            f32x4 v = {outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&a.C[col * a.ldc + rowq]) = v;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (rowq + u < a.M && col < rowq + u) a.C[col * a.ldc + rowq + u] = outv[4 * q + u];
          }
        }
      }
    }
}

// block -> tile list of gemm_nt_big: tiles ordered by squares of SQ x SQ tiles; chunk c of CH = SQ*SQ consecutive tiles goes
// to the blocks {b : b % 8 == c % 8} in order (blocks are dealt round-robin over the 8 XCDs), padded with (-1,-1)
int big_tile_list(Ctx* ctx, int64_t tm, int64_t tn, int lower, const int2** out, int64_t* nblocks) {
  constexpr int SQ = 6, CH = SQ * SQ;  // (a single tile column degenerates to runs of 36 row tiles per XCD)
  const std::string key = "gemm.tiles." + std::to_string(tm) + "x" + std::to_string(tn) + (lower ? "L" : "F");
  std::vector<int2> seq;
  for (int64_t si = 0; si < (tm + SQ - 1) / SQ; ++si)
    for (int64_t sj = 0; sj < (tn + SQ - 1) / SQ; ++sj) {
      if (lower && sj > si) break;
      for (int64_t i = si * SQ; i < std::min<int64_t>(tm, (si + 1) * SQ); ++i)
        for (int64_t j = sj * SQ; j < std::min<int64_t>(tn, (sj + 1) * SQ); ++j)
          if (!lower || j <= i) seq.push_back(make_int2((int)i, (int)j));
    }
  const int64_t nt = (int64_t)seq.size();
  const int64_t nb = round_up(nt, 8 * CH);
  *nblocks = nb;
  auto it = ctx->ws.find(key);
  if (it != ctx->ws.end()) {
    *out = static_cast<const int2*>(it->second.first);
    return SCLENS_OK;
  }
  std::vector<int2> list((size_t)nb);
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t loc = b >> 3, idx = ((loc / CH) * 8 + (b & 7)) * CH + loc % CH;
    list[(size_t)b] = idx < nt ? seq[(size_t)idx] : make_int2(-1, -1);
  }
  int2* dev = static_cast<int2*>(ctx->workspace(key, sizeof(int2) * (size_t)nb));
  if (!dev) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, hipMemcpyAsync(dev, list.data(), sizeof(int2) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `list` is a local
  *out = dev;
  return SCLENS_OK;
}

int gemm_f32(Ctx* ctx, const GemmArgs& a_in) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0) return SCLENS_OK;
  // C += P Q': the large-tile kernels start their accumulators from C (see acc_init in gemm_nt_big)
  if (a.alpha == 1.f && a.beta == 1.f && !a.colabsmax && a.splits <= 1) a.acc_init = 1;
  if (a.lower && a.M != a.N) return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: lower needs M == N");
  const int64_t tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
  const int64_t ntiles = a.lower ? tm * (tm + 1) / 2 : tm * tn;
  if (ntiles > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: too many tiles");
  const bool vec = ((reinterpret_cast<uintptr_t>(a.P) | reinterpret_cast<uintptr_t>(a.Q)) & 15u) == 0 &&
                   (a.ldp % 4 == 0) && (a.ldq % 4 == 0);
  if (a.splits > 1 && (a.k_chunk <= 0 || a.k_chunk % 16 != 0 || a.colabsmax || a.beta != 0.f))
    return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: bad split-K arguments");
  if (a.acc_init && (a.alpha != 1.f || a.beta != 1.f || a.colabsmax || a.splits > 1))
    return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: acc_init needs alpha == beta == 1, no split, no colabsmax");
  // ---- large-tile NT kernels
  const bool force_big = ctx->opt.gemm_force == 1;  // tests: the large-tile kernels on small shapes
  const bool big_ok = a.q_kcontig && vec && a.K >= GK && (a.splits <= 1 || a.k_chunk % GK == 0) && ctx->opt.gemm_force != 2;
  const int nsl = a.splits > 1 ? a.splits : 1;
  if (big_ok && a.N <= 64 && !a.lower && !a.colabsmax && (a.prefer_big || force_big)) {  // skinny: 256 x 64 tiles
    using Cfg = BigCfg<8, 1, 1, 2>;
    const int64_t bm = (a.M + GB - 1) / GB;
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm_nt_big<8, 1, 1, 2>), 2 * Cfg::STAGE * (int)sizeof(float)));
    hipLaunchKernelGGL((gemm_nt_big<8, 1, 1, 2>), dim3((unsigned)bm, (unsigned)nsl), dim3(512), 2 * Cfg::STAGE * sizeof(float),
                       ctx->stream, a, (const int2*)nullptr, 1);
    SCL_HIP(ctx, hipGetLastError());
    return SCLENS_OK;
  }
  const int64_t bm = (a.M + GB - 1) / GB, bn = (a.N + GB - 1) / GB;
  // the 256 x 256 kernel needs enough tiles for the 256 CUs (one workgroup each)
  // (measured at n = 10^4, 820 tiles, three concurrent decompositions: the 128x128 kernel, which shares a CU with the kernels
  // of the other streams, is as fast alone and 4 % faster in the mix; at n = 3 * 10^4 the large tiles are 11 % faster)
  const int64_t ntb = a.lower ? bm * (bm + 1) / 2 : bm * bn;
  const bool want_list = a.K >= 256 && ntb * nsl >= 1500;       // long contractions: tile list with compact squares per XCD
  const bool want_nolist = a.prefer_big && ntb * nsl >= 400;    // short K, shape changes per call: no list
  if (big_ok && (want_list || want_nolist || force_big) && a.N > 64) {
    using Cfg = BigCfg<2, 4, 4, 2>;
    const int2* tiles = nullptr;
    int64_t nb = ntb;
    if (want_list || (force_big && !a.prefer_big)) SCL_TRY(big_tile_list(ctx, bm, bn, a.lower, &tiles, &nb));
    SCL_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm_nt_big<2, 4, 4, 2>), 2 * Cfg::STAGE * (int)sizeof(float)));
    // stagger the CUs' phases for read-modify-write products with a short contraction and at least four rounds of tiles
    // (period ~ MFMA time of a tile, 0.21 us per unit of K at 2.4 GHz, + its HBM time)
    a.stagger_ns = 0;
    if (a.beta != 0.f && a.K <= 1024 && nsl == 1 && nb >= 4 * 256) a.stagger_ns = (int)(210.0 * (double)a.K + (a.lower ? 30000.0 : 20000.0));
    hipLaunchKernelGGL((gemm_nt_big<2, 4, 4, 2>), dim3((unsigned)nb, (unsigned)nsl), dim3(512), 2 * Cfg::STAGE * sizeof(float),
                       ctx->stream, a, tiles, (int)bn);
    SCL_HIP(ctx, hipGetLastError());
    return SCLENS_OK;
  }
  const unsigned gy = (unsigned)(a.splits > 1 ? a.splits : 1);
  if (a.M <= 64 && !a.lower) {  // 64 x 256 tiles
    const int64_t tn2 = (a.N + 255) / 256;
    dim3 grid((unsigned)tn2, gy), block(256);
    if (a.q_kcontig) {
      if (vec)
        hipLaunchKernelGGL((gemm_kernel<true, true, 1>), grid, block, 0, ctx->stream, a, (int)tn2);
      else
        hipLaunchKernelGGL((gemm_kernel<true, false, 1>), grid, block, 0, ctx->stream, a, (int)tn2);
    } else {
      if (vec)
        hipLaunchKernelGGL((gemm_kernel<false, true, 1>), grid, block, 0, ctx->stream, a, (int)tn2);
      else
        hipLaunchKernelGGL((gemm_kernel<false, false, 1>), grid, block, 0, ctx->stream, a, (int)tn2);
    }
    SCL_HIP(ctx, hipGetLastError());
    return SCLENS_OK;
  }
  dim3 grid((unsigned)ntiles, gy), block(256);
  if (a.q_kcontig) {
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<true, true, 2>), grid, block, 0, ctx->stream, a, (int)tn);
    else
      hipLaunchKernelGGL((gemm_kernel<true, false, 2>), grid, block, 0, ctx->stream, a, (int)tn);
  } else {
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<false, true, 2>), grid, block, 0, ctx->stream, a, (int)tn);
    else
      hipLaunchKernelGGL((gemm_kernel<false, false, 2>), grid, block, 0, ctx->stream, a, (int)tn);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
