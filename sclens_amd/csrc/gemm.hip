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
#include "common.h"

namespace scl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

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

template <bool QKC, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];

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
  const int64_t m0 = (int64_t)ti * BM, n0 = (int64_t)tj * BN;
  if (a.splits > 1) {  // this block's K slice and private output (grid.y = slice)
    const int64_t koff = (int64_t)blockIdx.y * a.k_chunk;
    const int64_t kleft = a.K - koff;
    a.K = kleft < 0 ? 0 : (kleft < a.k_chunk ? kleft : a.k_chunk);
    a.P += koff;
    a.Q += QKC ? koff : koff * a.ldq;
    a.C += (int64_t)blockIdx.y * a.c_split_off;
  }

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;

  // staging coordinates
  const int pr = tid >> 2, pkq = tid & 3;   // K-contiguous operand: rows pr, pr+64; k = 4*pkq..
  const int qk = tid >> 5, qnq = tid & 31;  // N-contiguous operand: k rows qk, qk+8; n = 4*qnq..

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra0, ra1, rb0, rb1;
  auto gload = [&](int64_t k0) {
    ra0 = ld4<VEC>(a.P, a.ldp, m0 + pr, a.M, k0 + 4 * pkq, a.K);
    ra1 = ld4<VEC>(a.P, a.ldp, m0 + pr + 64, a.M, k0 + 4 * pkq, a.K);
    if (QKC) {
      rb0 = ld4<VEC>(a.Q, a.ldq, n0 + pr, a.N, k0 + 4 * pkq, a.K);
      rb1 = ld4<VEC>(a.Q, a.ldq, n0 + pr + 64, a.N, k0 + 4 * pkq, a.K);
    } else {
      rb0 = ld4<VEC>(a.Q, a.ldq, k0 + qk, a.K, n0 + 4 * qnq, a.N);
      rb1 = ld4<VEC>(a.Q, a.ldq, k0 + qk + 8, a.K, n0 + 4 * qnq, a.N);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[buf][4 * pkq + i][pr] = ra0[i];
      As[buf][4 * pkq + i][pr + 64] = ra1[i];
    }
    if (QKC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Bs[buf][4 * pkq + i][pr] = rb0[i];
        Bs[buf][4 * pkq + i][pr + 64] = rb1[i];
      }
    } else {
      *reinterpret_cast<f32x4*>(&Bs[buf][qk][4 * qnq]) = rb0;
      *reinterpret_cast<f32x4*>(&Bs[buf][qk + 8][4 * qnq]) = rb1;
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

int gemm_f32(Ctx* ctx, const GemmArgs& a) {
  if (a.M <= 0 || a.N <= 0) return SCLENS_OK;
  if (a.lower && a.M != a.N) return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: lower needs M == N");
  const int64_t tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
  const int64_t ntiles = a.lower ? tm * (tm + 1) / 2 : tm * tn;
  if (ntiles > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: too many tiles");
  const bool vec = ((reinterpret_cast<uintptr_t>(a.P) | reinterpret_cast<uintptr_t>(a.Q)) & 15u) == 0 &&
                   (a.ldp % 4 == 0) && (a.ldq % 4 == 0);
  if (a.splits > 1 && (a.k_chunk <= 0 || a.k_chunk % 16 != 0 || a.colabsmax || a.beta != 0.f))
    return ctx->fail(SCLENS_ERR_ARG, "gemm_f32: bad split-K arguments");
  dim3 grid((unsigned)ntiles, (unsigned)(a.splits > 1 ? a.splits : 1)), block(256);
  if (a.q_kcontig) {
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, ctx->stream, a, (int)tn);
    else
      hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, ctx->stream, a, (int)tn);
  } else {
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, ctx->stream, a, (int)tn);
    else
      hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, ctx->stream, a, (int)tn);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
