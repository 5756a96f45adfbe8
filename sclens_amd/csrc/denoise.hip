// get_denoised_df (scLENS.jl:889-931) on the device: SURVEY 8(f-2), the immediate consumer of sclens()'s result.
//   d_mean = pca_n1 * gene_basis[sig_id,:] * sqrt(M)                      (one fp32 MFMA GEMM, K = |sig_id|)
//   r = ((d_mean + cent_) * (norm_tgc / mean(norm_tgc)) * mat2_std + mat2_mean            (inverse normalisation)
//   r = max(exp(r) - 1, 0);  r ./= sum(r, dims=2);  r .*= mean(TGC)
// Output is N x M column-major (cells x genes), i.e. row-major [gene][cell] on the device: every access is coalesced
// over cells and the per-cell sums are a loop over genes inside one thread.
#include <cmath>
#include <vector>

#include "common.h"

namespace scl {

// in place on D[M][ldd] (gene-major); one thread per cell
__global__ __launch_bounds__(256) void k_denoise_finish(float* __restrict__ D, int64_t N, int64_t M, int64_t ldd,
                                                        const float* __restrict__ cent, const float* __restrict__ rowscale,
                                                        const float* __restrict__ stdv, const float* __restrict__ mean,
                                                        float sqrtM, float mean_tgc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float rs = rowscale[i];
  double sum = 0.0;
  for (int64_t j = 0; j < M; ++j) {
    float v = (D[j * ldd + i] * sqrtM + cent[j]) * rs;
    v = v * stdv[j] + mean[j];
    v = expf(v) - 1.f;
    v = v < 0.f ? 0.f : v;
    D[j * ldd + i] = v;
    sum += (double)v;
  }
  const float f = (float)((double)mean_tgc / sum);
  for (int64_t j = 0; j < M; ++j) D[j * ldd + i] *= f;
}

int denoise_host(Ctx* ctx, const float* pca_n1, int64_t N, int64_t s, const float* gene_basis_sig, int64_t M,
                 const double* tgc, const double* mat2_mean, const double* mat2_std, const double* norm_tgc,
                 const double* cent, float* out) {
  if (!pca_n1 || !gene_basis_sig || !out || N <= 0 || M <= 0 || s <= 0 || !tgc || !mat2_mean || !mat2_std || !norm_tgc || !cent)
    return ctx->fail(SCLENS_ERR_ARG, "get_denoised: bad arguments");
  StageTimer tm(ctx, "denoise");
  hipStream_t st = ctx->stream;
  const int64_t ldn = round_up(N, 32), lds = round_up(s, 4);
  SCL_WS(ctx, dQ, float, "dn.Q", s * ldn);   // pca_n1 column-major N x s == row-major [s][N]
  SCL_WS(ctx, dP, float, "dn.P", M * lds);   // gene_basis_sig^T: [M][s]
  SCL_WS(ctx, dD, float, "dn.D", M * ldn);
  SCL_WS(ctx, dv, float, "dn.v", 3 * M + N);
  SCL_HIP(ctx, hipMemsetAsync(dQ, 0, sizeof(float) * s * ldn, st));
  SCL_HIP(ctx, hipMemcpy2DAsync(dQ, sizeof(float) * ldn, pca_n1, sizeof(float) * N, sizeof(float) * N, s, hipMemcpyHostToDevice, st));
  std::vector<float> hP((size_t)M * lds, 0.f), hv((size_t)3 * M + N);
  for (int64_t q = 0; q < s; ++q)
    for (int64_t j = 0; j < M; ++j) hP[j * lds + q] = gene_basis_sig[q * M + j];  // input: s rows of M genes
  double ml = 0.0, mt = 0.0;
  for (int64_t i = 0; i < N; ++i) { ml += norm_tgc[i]; mt += tgc[i]; }
  ml /= (double)N; mt /= (double)N;
  for (int64_t j = 0; j < M; ++j) { hv[j] = (float)cent[j]; hv[M + j] = (float)mat2_std[j]; hv[2 * M + j] = (float)mat2_mean[j]; }
  for (int64_t i = 0; i < N; ++i) hv[3 * M + i] = (float)(norm_tgc[i] / ml);
  SCL_HIP(ctx, hipMemcpyAsync(dP, hP.data(), sizeof(float) * hP.size(), hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemcpyAsync(dv, hv.data(), sizeof(float) * hv.size(), hipMemcpyHostToDevice, st));
  GemmArgs g{};
  g.P = dP; g.Q = dQ; g.C = dD;
  g.M = M; g.N = N; g.K = s;
  g.ldp = lds; g.ldq = ldn; g.ldc = ldn;
  g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
  SCL_TRY(gemm_f32(ctx, g));
  hipLaunchKernelGGL(k_denoise_finish, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dD, N, M, ldn, dv, dv + 3 * M,
                     dv + M, dv + 2 * M, (float)std::sqrt((double)M), (float)mt);
  SCL_HIP(ctx, hipGetLastError());
  SCL_HIP(ctx, hipMemcpy2DAsync(out, sizeof(float) * N, dD, sizeof(float) * ldn, sizeof(float) * N, M, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  return SCLENS_OK;
}

}  // namespace scl
