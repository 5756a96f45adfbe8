// Normalisation of a sparse count matrix into the dense, scaled fp32 matrix the Gram kernel consumes.
// Device restatement of `logn_scale(pre_scale(x))` (scLENS.jl:650-652 -> proj_l :607, log1p,
// zscore_with_l2 :596-605, scaled_gdata "cent" :300-305) and of its inline Float64 twin for the
// data matrix (scLENS.jl:676-696, which also yields rec_vals). The reference densifies on the host
// with 2-3 N x M Float64 temporaries; here all statistics are O(nnz) passes over a fixed sparse
// pattern (CSC + CSR views of the same entries) and the dense matrix is written exactly once:
//     X_ij = s_i (Z_ij - mu_j) - cent_j ,   Z_ij = log1p(x_ij / TGC_i) / std_j ,  s_i = mean(l)/l_i
// (the identity of scLENS.jl:601-603 / :688-690 for l_i). centering="median" (scLENS.jl:653-654: scaled_gdata "median"
// then norm_l :608) is the same expression with mu_j = median_j / std_j and cent = 0. Statistics are accumulated in fp64 in a
// fixed order (deterministic); the dense output is fp32.
//
// The pattern is the union of the stored counts and the zero-candidate positions (scLENS.jl:668-673);
// a perturbed matrix (scLENS.jl:735, :774) is just another value array over the same pattern, so a
// perturbation costs one scatter of ones instead of a sparse(...) rebuild.
#include "common.h"
#include "pattern.h"
#include "rng.h"

namespace scl {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- K1: row sums (CSR view), one wave per row ---------------------------------------------------
// A value array over a pattern with CSR companions carries its values a second time in CSR slot order (val + nU): streamed.
__global__ __launch_bounds__(256) void k_row_sums(PatternDev p, const float* __restrict__ val,
                                                  double* __restrict__ tgc) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= p.N) return;
  double s = 0.0;
  const int64_t b = p.rowptr[row], e = p.rowptr[row + 1];
  // four loads of a lane in flight per trip (the lane still adds its entries in ascending order: same sums as the plain loop)
  if (p.base_val_csr) {
    const float* vc = val + p.nU;
    for (int64_t q0 = b + lane; q0 < e; q0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (q0 + 64 * u < e) ? vc[q0 + 64 * u] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + 64 * u < e) s += (double)v[u];
    }
  } else {
    for (int64_t q0 = b + lane; q0 < e; q0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (q0 + 64 * u < e) ? val[p.csr2csc[q0 + 64 * u]] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + 64 * u < e) s += (double)v[u];
    }
  }
  s = wsum(s);
  if (lane == 0) tgc[row] = s;
}

// ---- K2: lg = log1p(x / TGC_row) per stored entry (CSC order) + column mean/std -------------------
// f32path = 1: the closure path (Float32 proj_l + log1p, Float32 std)
__global__ __launch_bounds__(256) void k_col_stats(PatternDev p, const float* __restrict__ val,
                                                   const double* __restrict__ tgc, int f32path,
                                                   double* __restrict__ lg, double* __restrict__ mean,
                                                   double* __restrict__ stdv, double* __restrict__ mu) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  const int64_t b = p.colptr[col], e = p.colptr[col + 1];
  double s = 0.0;
  // four entries of a lane per trip: their loads, then their gathers of the cell totals, are in flight together (a trip of the plain
  // loop was one dependent chain load -> gather -> log1p -> store: 6.7 ms for 8 GB at 100 000 x 30 000); the lane still takes its
  // entries in ascending order, so the sums are the ones of the plain loop
  for (int64_t q0 = b + lane; q0 < e; q0 += 256) {
    float v[4];
    int32_t r[4];
    double t[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = q0 + 64 * u < e;
      v[u] = in ? val[q0 + 64 * u] : 0.f;
      r[u] = in ? p.row[q0 + 64 * u] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = tgc[r[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      l[u] = 0.0;
      if (v[u] != 0.f) {
        if (f32path) {
          const float inv = 1.0f / (float)t[u];
          l[u] = (double)log1pf(inv * v[u]);
        } else {
          l[u] = log1p((double)v[u] / t[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + 64 * u < e) {
        lg[q0 + 64 * u] = l[u];
        s += l[u];
      }
  }
  s = wsum(s);
  const double m = s / (double)p.N;
  double s2 = 0.0;
  for (int64_t q0 = b + lane; q0 < e; q0 += 256) {
    double l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) l[u] = (q0 + 64 * u < e) ? lg[q0 + 64 * u] : m;  // written by this lane above
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + 64 * u < e) {
        const double dlt = l[u] - m;
        s2 += dlt * dlt;
      }
  }
  s2 = wsum(s2);
  s2 += (double)(p.N - (e - b)) * m * m;  // implicit zeros
  double sd = sqrt(s2 / (double)(p.N - 1));
  if (f32path) sd = (double)(float)sd;
  if (lane == 0) {
    mean[col] = f32path ? (double)(float)m : m;
    stdv[col] = sd;
    mu[col] = (s / sd) / (double)p.N;  // mean of the std-scaled column
  }
}

// ---- median centring (scLENS.jl:653-654 -> scaled_gdata(position_="median") :291-298, :328) ----------------------
// mu_j := median_i(lg_ij) / std_j over ALL N cells (the reference densifies first, so implicit zeros count). One block
// per gene. Sorted column = (N - nz) zeros, then the nz positives ascending: genes whose positives do not reach the
// middle have median 0 (the usual case at 90 % sparsity); for the others the one or two middle order statistics are
// found exactly by an MSB-first radix select on the bit pattern of the positive values (8 bits per pass, histogram in
// LDS with integer atomics: order-independent, deterministic). Even N: mean of the two middle values (`middle`).
__device__ __forceinline__ double radix_select_pos(const double* __restrict__ v, int64_t len, int64_t k, int* hist,
                                                   unsigned long long* sh) {
  // k-th smallest (0-based) among the entries v[0..len) that are > 0
  unsigned long long prefix = 0ull, mask = 0ull;
  for (int pass = 7; pass >= 0; --pass) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int64_t q = threadIdx.x; q < len; q += blockDim.x) {
      const double x = v[q];
      if (x > 0.0) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(x);
        if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> (8 * pass)) & 255ull)], 1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t cum = 0;
      int b = 0;
      for (; b < 255; ++b) {
        if (cum + hist[b] > k) break;
        cum += hist[b];
      }
      sh[0] = (unsigned long long)b;
      sh[1] = (unsigned long long)(k - cum);
    }
    __syncthreads();
    prefix |= sh[0] << (8 * pass);
    mask |= 255ull << (8 * pass);
    k = (int64_t)sh[1];
    __syncthreads();
  }
  return __longlong_as_double((long long)prefix);
}

__global__ __launch_bounds__(256) void k_col_median(PatternDev p, const double* __restrict__ lg,
                                                    const double* __restrict__ stdv, int f32path,
                                                    double* __restrict__ mu) {
  __shared__ int hist[256];
  __shared__ unsigned long long sh[2];
  __shared__ int64_t cnt_s[4];
  const int64_t col = blockIdx.x;
  const int64_t b = p.colptr[col], len = p.colptr[col + 1] - b;
  int64_t nzl = 0;
  for (int64_t q = threadIdx.x; q < len; q += 256) nzl += lg[b + q] > 0.0 ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nzl += __shfl_xor(nzl, o);
  if ((threadIdx.x & 63) == 0) cnt_s[threadIdx.x >> 6] = nzl;
  __syncthreads();
  const int64_t nz = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
  const int64_t zeros = p.N - nz;
  const int64_t r2 = p.N / 2, r1 = (p.N & 1) ? r2 : r2 - 1;  // 0-based ranks of the middle element(s)
  double med = 0.0;
  if (r2 >= zeros) {  // block-uniform branch
    const double v2 = radix_select_pos(lg + b, len, r2 - zeros, hist, sh);
    double v1 = v2;
    if (r1 != r2) v1 = (r1 >= zeros) ? radix_select_pos(lg + b, len, r1 - zeros, hist, sh) : 0.0;
    med = f32path ? (double)((float)v1 * 0.5f + (float)v2 * 0.5f) : 0.5 * v1 + 0.5 * v2;
    if (r1 == r2) med = v2;
  }
  if (threadIdx.x == 0) mu[col] = med / stdv[col];
}

// ---- deterministic reductions of a vector: out[0] = sum(f(v)) ---------------------------------------
// mode 0: sum v^2, mode 1: sum v, mode 2: sum 1/v
__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ v, int64_t n, int mode,
                                                 double* __restrict__ out) {
  __shared__ double sw[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double x = v[i];
    s += (mode == 0) ? x * x : (mode == 1 ? x : 1.0 / x);
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sw[w];
    out[0] = t;
  }
}

// ---- K3: row L2 norms of the centred row (CSR view) ------------------------------------------------
// With CSR companions the scaled entry is RECOMPUTED from the CSR-ordered value (the same expression k_col_stats evaluates,
// hence the same bits) instead of gathering lg through csr2csc: the kernel streams 4 + 4 bytes per slot.
__global__ __launch_bounds__(256) void k_row_norms(PatternDev p, const float* __restrict__ val, const double* __restrict__ tgc, int f32path,
                                                   const double* __restrict__ lg, const double* __restrict__ stdv,
                                                   const double* __restrict__ mu, const double* __restrict__ mu2sum,
                                                   double* __restrict__ l2) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= p.N) return;
  double zz = 0.0, zm = 0.0;
  if (p.base_val_csr) {
    const float* vc = val + p.nU;
    const double t = tgc[row];
    const float inv = 1.0f / (float)t;
    for (int64_t q = p.rowptr[row] + lane; q < p.rowptr[row + 1]; q += 64) {
      const float v = vc[q];
      if (v == 0.f) continue;
      const int64_t c = p.csrcol[q];
      const double l = f32path ? (double)log1pf(inv * v) : log1p((double)v / t);
      const double z = l / stdv[c];
      zz += z * z;
      zm += z * mu[c];
    }
  } else {
    for (int64_t q = p.rowptr[row] + lane; q < p.rowptr[row + 1]; q += 64) {
      const int64_t pos = p.csr2csc[q];
      const int64_t c = p.csrcol[q];
      const double z = lg[pos] / stdv[c];
      zz += z * z;
      zm += z * mu[c];
    }
  }
  zz = wsum(zz);
  zm = wsum(zm);
  if (lane == 0) l2[row] = sqrt(zz - 2.0 * zm + mu2sum[0]);
}

// s_i = mean(l) / l_i
__global__ void k_row_scale(const double* __restrict__ l2, int64_t N, const double* __restrict__ lsum,
                            double* __restrict__ srow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) srow[i] = (lsum[0] / (double)N) / l2[i];
}

// ---- K4: cent_j = mean_i s_i (Z_ij - mu_j) (CSC view) ----------------------------------------------
__global__ __launch_bounds__(256) void k_col_cent(PatternDev p, const double* __restrict__ lg,
                                                  const double* __restrict__ stdv, const double* __restrict__ mu,
                                                  const double* __restrict__ srow, const double* __restrict__ ssum,
                                                  double* __restrict__ cent) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  double s = 0.0;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64) s += srow[p.row[q]] * lg[q];
  s = wsum(s);
  if (lane == 0) cent[col] = (s / stdv[col] - mu[col] * ssum[0]) / (double)p.N;
}

// ---- K5: dense write. cells_major = 1: B[i][j] (ld over genes); 0: B[j][i] (ld over cells) ---------
__global__ __launch_bounds__(256) void k_dense_fill(int64_t N, int64_t M, int cells_major,
                                                    const double* __restrict__ srow, const double* __restrict__ mu,
                                                    const double* __restrict__ cent, float* __restrict__ B,
                                                    int64_t ldb) {
  // grid.y = B row, grid.x over the contiguous dimension
  const int64_t r = blockIdx.y;
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t nc = cells_major ? M : N;
  if (c >= nc) return;
  float out[4];
  if (cells_major) {
    const double s = srow[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (c + e < nc) ? (float)(-s * mu[c + e] - cent[c + e]) : 0.f;
  } else {
    const double m = mu[r], ce = cent[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (c + e < nc) ? (float)(-srow[c + e] * m - ce) : 0.f;
  }
  float* dst = B + r * ldb + c;
  if (c + 3 < ldb) {
    *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
  } else {
    for (int e = 0; e < 4 && c + e < ldb; ++e) dst[e] = out[e];
  }
}

__global__ __launch_bounds__(256) void k_dense_scatter(PatternDev p, int cells_major, const float* __restrict__ val,
                                                       const double* __restrict__ lg, const double* __restrict__ stdv,
                                                       const double* __restrict__ mu, const double* __restrict__ cent,
                                                       const double* __restrict__ srow, float* __restrict__ B,
                                                       int64_t ldb) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= p.M) return;
  const double isd = 1.0 / stdv[col], m = mu[col], ce = cent[col];
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64) {
    if (val[q] == 0.f) continue;
    const int64_t r = p.row[q];
    const float x = (float)(srow[r] * (lg[q] * isd - m) - ce);
    if (cells_major) B[r * ldb + col] = x;
    else B[col * ldb + r] = x;
  }
}

// ---- K5 fused (round 3, gene-major B[j][i] only): one workgroup per gene walks the cells in chunks of DF_CH: the chunk's background
// -s_i mu_j - cent_j is formed in LDS, the gene's stored entries that fall into the chunk overwrite their slots with the same
// expression k_dense_scatter uses, and the chunk leaves as 16-byte stores: the 4 N M bytes are written ONCE (fill + scatter wrote
// them, then re-wrote 64-byte sectors around each of the nnz scattered dwords: 19 ms for the two kernels at 100 000 x 30 000,
// `profiles/r03_cfg4_kernel_stats_mid2.csv`). A column of the pattern holds the counts with ascending cells and THEN the zero
// candidates in draw order, so a running position q replaces any search as long as the entries keep coming in order: q advances
// over the PREFIX of entries that lie in the current chunk; whatever is left when the last chunk has been written (the unordered
// tail) is scattered as before, behind a barrier (same workgroup, same addresses: the chunk stores have completed).
constexpr int DF_CH = 4096;
__global__ __launch_bounds__(256) void k_dense_fused(PatternDev p, const float* __restrict__ val, const double* __restrict__ lg,
                                                     const double* __restrict__ stdv, const double* __restrict__ mu,
                                                     const double* __restrict__ cent, const double* __restrict__ srow,
                                                     float* __restrict__ B, int64_t ldb) {
  __shared__ __attribute__((aligned(16))) float chunk[DF_CH];
  __shared__ int wlead[4];
  const int64_t col = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double isd = 1.0 / stdv[col], m = mu[col], ce = cent[col];
  int64_t q = p.colptr[col];
  const int64_t qend = p.colptr[col + 1];
  float* dst = B + col * ldb;
  for (int64_t c0 = 0; c0 < ldb; c0 += DF_CH) {
    const int64_t cend = (c0 + DF_CH < ldb) ? c0 + DF_CH : ldb;
#pragma unroll 4
    for (int c = tid; c < DF_CH; c += 256) {
      const int64_t i = c0 + c;
      chunk[c] = (i < p.N) ? (float)(-srow[i] * m - ce) : 0.f;
    }
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
      if (tid < cnt && val[idx] != 0.f) chunk[r - c0] = (float)(srow[r] * (lg[idx] * isd - m) - ce);
      __syncthreads();
      q += cnt;
      if (cnt < 256) break;
    }
    for (int c = 4 * tid; c0 + c < cend; c += 1024) {
      if (c0 + c + 3 < cend)
        *reinterpret_cast<float4*>(dst + c0 + c) = *reinterpret_cast<const float4*>(chunk + c);
      else
        for (int e = 0; e < 4 && c0 + c + e < cend; ++e) dst[c0 + c + e] = chunk[c + e];
    }
    __syncthreads();
  }
  for (int64_t idx = q + tid; idx < qend; idx += 256) {  // the entries that did not come in cell order
    if (val[idx] == 0.f) continue;
    const int64_t r = p.row[idx];
    dst[r] = (float)(srow[r] * (lg[idx] * isd - m) - ce);
  }
}

int scale_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, ScaleStats* out) {
  const int64_t N = p.N, M = p.M;
  SCL_WS(ctx, tgc, double, "sc.tgc", N);
  SCL_WS(ctx, lg, double, "sc.lg", p.nU);
  SCL_WS(ctx, mean, double, "sc.mean", M);
  SCL_WS(ctx, stdv, double, "sc.std", M);
  SCL_WS(ctx, mu, double, "sc.mu", M);
  SCL_WS(ctx, l2, double, "sc.l2", N);
  SCL_WS(ctx, srow, double, "sc.srow", N);
  SCL_WS(ctx, cent, double, "sc.cent", M);
  SCL_WS(ctx, red, double, "sc.red", 8);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_row_sums, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc);
  hipLaunchKernelGGL(k_col_stats, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, val, tgc, f32path, lg, mean,
                     stdv, mu);
  if (centering == 1)  // median centring: the column offset is the median instead of the mean
    hipLaunchKernelGGL(k_col_median, dim3((unsigned)M), dim3(256), 0, st, p, lg, stdv, f32path, mu);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, mu, M, 0, red + 0);  // ||mu||^2
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc, f32path, lg, stdv, mu, red + 0, l2);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, l2, N, 1, red + 1);  // sum l
  hipLaunchKernelGGL(k_row_scale, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, l2, N, red + 1, srow);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, srow, N, 1, red + 2);  // sum s
  if (centering == 1) {  // norm_l(scaled_gdata(.,"median")) has no final centring (scLENS.jl:654)
    SCL_HIP(ctx, hipMemsetAsync(cent, 0, sizeof(double) * M, st));
  } else {
    hipLaunchKernelGGL(k_col_cent, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, lg, stdv, mu, srow, red + 2,
                       cent);
  }
  SCL_HIP(ctx, hipGetLastError());
  *out = ScaleStats{tgc, lg, mean, stdv, mu, l2, srow, cent, red};
  return SCLENS_OK;
}

int scale_to_dense(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, int cells_major,
                   float* B, int64_t ldb, ScaleVecs* keep) {
  return scale_to_dense_stats(ctx, p, val, f32path, centering, cells_major, B, ldb, keep, nullptr);
}
int scale_to_dense_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, int cells_major, float* B, int64_t ldb,
                         ScaleVecs* keep, ScaleStats* out) {
  StageTimer tm(ctx, "scale");
  const int64_t N = p.N, M = p.M;
  ScaleStats ss;
  SCL_TRY(scale_stats(ctx, p, val, f32path, centering, &ss));
  if (out) *out = ss;
  const double *tgc = ss.tgc, *lg = ss.lg, *mean = ss.mean, *stdv = ss.stdv, *mu = ss.mu, *l2 = ss.l2, *srow = ss.srow,
               *cent = ss.cent;
  hipStream_t st = ctx->stream;
  const int64_t nr = cells_major ? N : M, nc = cells_major ? M : N;
  if (nr > 65535LL * 65535LL) return ctx->fail(SCLENS_ERR_ARG, "scale_to_dense: too many rows");
  const bool fused_ok = ctx->opt.dense_fused != 0;  // 0: the separate fill + scatter kernels (same values)
  if (!B) {
    // statistics only (the caller forms the Gram matrix from the sparse structure and does not need the dense matrix)
  } else if (!cells_major && fused_ok && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15u) == 0) {
    hipLaunchKernelGGL(k_dense_fused, dim3((unsigned)M), dim3(256), 0, st, p, val, lg, stdv, mu, cent, srow, B, ldb);
  } else {
  // grid.y is limited to 65535: loop over row slabs
  for (int64_t r0 = 0; r0 < nr; r0 += 65535) {
    const int64_t rows = (nr - r0 < 65535) ? nr - r0 : 65535;
    hipLaunchKernelGGL(k_dense_fill, dim3((unsigned)((nc + 1023) / 1024), (unsigned)rows), dim3(256), 0, st, N, M,
                       cells_major, cells_major ? srow + r0 : srow, cells_major ? mu : mu + r0,
                       cells_major ? cent : cent + r0, B + r0 * ldb, ldb);
  }
  hipLaunchKernelGGL(k_dense_scatter, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, cells_major, val, lg, stdv,
                     mu, cent, srow, B, ldb);
  }
  SCL_HIP(ctx, hipGetLastError());
  if (keep) {  // rec_vals of the data matrix (scLENS.jl:676-696) -> host
    SCL_HIP(ctx, hipMemcpyAsync(keep->tgc, tgc, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_mean, mean, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_std, stdv, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->norm_tgc, l2, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->cent, cent, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
  }
  return SCLENS_OK;
}

// ---- row-sharded variant (atlas configuration): this process holds N_local of the N_global cells -------
// Per-cell quantities (TGC, l_i, s_i) are local; the statistics that span all cells are formed as local partial sums, summed
// over the ranks by the caller-supplied all-reduce (ShardReduce; RCCL through torch.distributed, or gloo in the tests)
// and finished identically on every rank: three all-reduces of O(M) doubles and one scalar per normalisation.
__global__ __launch_bounds__(256) void k_sh_col_sum(PatternDev p, const float* __restrict__ val,
                                                    const double* __restrict__ tgc, int f32path, double* __restrict__ lg,
                                                    double* __restrict__ sum_cnt, int64_t M) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= M) return;
  const int64_t b = p.colptr[col], e = p.colptr[col + 1];
  double s = 0.0;
  for (int64_t q = b + lane; q < e; q += 64) {
    const float v = val[q];
    double l = 0.0;
    if (v != 0.f) {
      const int64_t r = p.row[q];
      if (f32path) {
        const float inv = 1.0f / (float)tgc[r];
        l = (double)log1pf(inv * v);
      } else {
        l = log1p((double)v / tgc[r]);
      }
    }
    lg[q] = l;
    s += l;
  }
  s = wsum(s);
  if (lane == 0) {
    sum_cnt[col] = s;
    sum_cnt[M + col] = (double)(e - b);  // local slots of the column (exact in fp64)
  }
}
__global__ __launch_bounds__(256) void k_sh_col_var(PatternDev p, const double* __restrict__ lg,
                                                    const double* __restrict__ sum_cnt, double n_global,
                                                    double* __restrict__ s2out, int64_t M) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= M) return;
  const double m = sum_cnt[col] / n_global;
  double s2 = 0.0;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64) {
    const double dlt = lg[q] - m;
    s2 += dlt * dlt;
  }
  s2 = wsum(s2);
  if (lane == 0) s2out[col] = s2;
}
__global__ void k_sh_col_fin(const double* __restrict__ sum_cnt, const double* __restrict__ s2, double n_global, int f32path,
                             int64_t M, double* __restrict__ mean, double* __restrict__ stdv, double* __restrict__ mu) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= M) return;
  const double s = sum_cnt[col], m = s / n_global;
  const double tot = s2[col] + (n_global - sum_cnt[M + col]) * m * m;  // implicit zeros of all ranks
  double sd = sqrt(tot / (n_global - 1.0));
  if (f32path) sd = (double)(float)sd;
  mean[col] = f32path ? (double)(float)m : m;
  stdv[col] = sd;
  mu[col] = (s / sd) / n_global;
}
__global__ void k_sh_row_scale(const double* __restrict__ l2, int64_t n_local, double n_global,
                               const double* __restrict__ lsum, double* __restrict__ srow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_local) srow[i] = (lsum[0] / n_global) / l2[i];
}
__global__ __launch_bounds__(256) void k_sh_col_cent_part(PatternDev p, const double* __restrict__ lg,
                                                          const double* __restrict__ srow, double* __restrict__ part,
                                                          int64_t M) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= M) return;
  double s = 0.0;
  for (int64_t q = p.colptr[col] + lane; q < p.colptr[col + 1]; q += 64) s += srow[p.row[q]] * lg[q];
  s = wsum(s);
  if (lane == 0) part[col] = s;
}
__global__ void k_sh_col_cent_fin(const double* __restrict__ part, const double* __restrict__ stdv,
                                  const double* __restrict__ mu, double n_global, int64_t M, double* __restrict__ cent) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col < M) cent[col] = (part[col] / stdv[col] - mu[col] * part[M]) / n_global;  // part[M] = sum of s over all cells
}

// statistics of the row-sharded normalisation: per-cell vectors cover the local cells, per-gene vectors are global (identical on
// every rank after the all-reduces)
int scale_stats_sharded(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const ShardReduce& sh, ScaleStats* out) {
  const int64_t N = p.N, M = p.M;  // N = local cells
  const double ng = (double)sh.N_global;
  SCL_WS(ctx, tgc, double, "sc.tgc", N);
  SCL_WS(ctx, lg, double, "sc.lg", p.nU);
  SCL_WS(ctx, mean, double, "sc.mean", M);
  SCL_WS(ctx, stdv, double, "sc.std", M);
  SCL_WS(ctx, mu, double, "sc.mu", M);
  SCL_WS(ctx, l2, double, "sc.l2", N);
  SCL_WS(ctx, srow, double, "sc.srow", N);
  SCL_WS(ctx, cent, double, "sc.cent", M);
  SCL_WS(ctx, red, double, "sc.red", 8);
  SCL_WS(ctx, part, double, "sc.part", 2 * M + 8);
  SCL_WS(ctx, s2, double, "sc.s2", M);
  hipStream_t st = ctx->stream;
  const unsigned gc4 = (unsigned)((M + 3) / 4), gcm = (unsigned)((M + 255) / 256);
  hipLaunchKernelGGL(k_row_sums, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc);  // all genes are local
  hipLaunchKernelGGL(k_sh_col_sum, dim3(gc4), dim3(256), 0, st, p, val, tgc, f32path, lg, part, M);
  SCL_TRY(sh.sum(ctx, part, 2 * M, 0));
  hipLaunchKernelGGL(k_sh_col_var, dim3(gc4), dim3(256), 0, st, p, lg, part, ng, s2, M);
  SCL_TRY(sh.sum(ctx, s2, M, 0));
  hipLaunchKernelGGL(k_sh_col_fin, dim3(gcm), dim3(256), 0, st, part, s2, ng, f32path, M, mean, stdv, mu);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, mu, M, 0, red + 0);  // ||mu||^2 (replicated)
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc, f32path, lg, stdv, mu, red + 0, l2);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, l2, N, 1, red + 1);  // local sum of l
  SCL_TRY(sh.sum(ctx, red + 1, 1, 0));
  hipLaunchKernelGGL(k_sh_row_scale, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, l2, N, ng, red + 1, srow);
  hipLaunchKernelGGL(k_sh_col_cent_part, dim3(gc4), dim3(256), 0, st, p, lg, srow, part, M);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, srow, N, 1, part + M);  // local sum of s, appended
  SCL_TRY(sh.sum(ctx, part, M + 1, 0));
  hipLaunchKernelGGL(k_sh_col_cent_fin, dim3(gcm), dim3(256), 0, st, part, stdv, mu, ng, M, cent);
  SCL_HIP(ctx, hipGetLastError());
  *out = ScaleStats{tgc, lg, mean, stdv, mu, l2, srow, cent, red};
  return SCLENS_OK;
}

int scale_to_dense_sharded(Ctx* ctx, const PatternDev& p, const float* val, int f32path, float* B, int64_t ldb,
                           ScaleVecs* keep, const ShardReduce& sh) {
  StageTimer tm(ctx, "scale");
  const int64_t N = p.N, M = p.M;
  ScaleStats ss;
  SCL_TRY(scale_stats_sharded(ctx, p, val, f32path, sh, &ss));
  const double *tgc = ss.tgc, *lg = ss.lg, *mean = ss.mean, *stdv = ss.stdv, *mu = ss.mu, *l2 = ss.l2, *srow = ss.srow, *cent = ss.cent;
  hipStream_t st = ctx->stream;
  const unsigned gc4 = (unsigned)((M + 3) / 4);
  if (M > 65535LL * 65535LL) return ctx->fail(SCLENS_ERR_ARG, "scale_to_dense: too many rows");
  for (int64_t r0 = 0; r0 < M; r0 += 65535) {  // genes-major: B[j][i_local]
    const int64_t rows = (M - r0 < 65535) ? M - r0 : 65535;
    hipLaunchKernelGGL(k_dense_fill, dim3((unsigned)((N + 1023) / 1024), (unsigned)rows), dim3(256), 0, st, N, M, 0, srow,
                       mu + r0, cent + r0, B + r0 * ldb, ldb);
  }
  hipLaunchKernelGGL(k_dense_scatter, dim3(gc4), dim3(256), 0, st, p, 0, val, lg, stdv, mu, cent, srow, B, ldb);
  SCL_HIP(ctx, hipGetLastError());
  if (keep) {
    SCL_HIP(ctx, hipMemcpyAsync(keep->tgc, tgc, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_mean, mean, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_std, stdv, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->norm_tgc, l2, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipMemcpyAsync(keep->cent, cent, sizeof(double) * M, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
  }
  return SCLENS_OK;
}

// ---- chunked variant (cells > genes, ALL cells on this device but visited in chunks of rows; session.hip, "chunked session") -----
// The same statistics as the row-sharded path, with the sums over ranks replaced by sums over the chunks of ONE session: a matrix
// whose scaled form (4 N M bytes: 120 GB at 1 000 000 x 30 000) does not fit is normalised and contracted chunk by chunk,
//     B'B = sum over chunks of B_g' B_g                       (the identity behind scLENS.jl:332-361 for dims = 2)
// Three passes over the chunks instead of the row-sharded path's five exchange points: (1) per-gene sums, (2) per-gene squared
// deviations (two-pass variance, as everywhere else), (3) the cells' norms l_i together with the Gram contribution, for which the
// global mean of the norms c = mean(l) is not needed yet: with U_ij = (Z_ij - mu_j) / l_i (unit rows) the scaled matrix is
// B = c U - 1 cent', cent_j = (c / N) sum_i U_ij, hence B'B = c^2 U'U - N cent cent' -- the chunk contributes U_g'U_g and
// sum_i U_ij, and c, cent and the rank-one term are applied once at the end in fp64.
__global__ void k_axpy_f64(const double* __restrict__ x, int64_t n, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
__global__ void k_inv_f64(const double* __restrict__ x, int64_t n, double num, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = num / x[i];
}
static int chunk_lg(Ctx* ctx, const PatternDev& p, const float* val, int f32path, double** tgc_out, double** lg_out, double* part) {
  SCL_WS(ctx, tgc, double, "sc.tgc", p.N);
  SCL_WS(ctx, lg, double, "sc.lg", p.nU);
  hipLaunchKernelGGL(k_row_sums, dim3((unsigned)((p.N + 3) / 4)), dim3(256), 0, ctx->stream, p, val, tgc);
  hipLaunchKernelGGL(k_sh_col_sum, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, ctx->stream, p, val, tgc, f32path, lg, part, p.M);
  SCL_HIP(ctx, hipGetLastError());
  *tgc_out = tgc;
  *lg_out = lg;
  return SCLENS_OK;
}
// pass 1: acc[0..M) += per-gene sums of lg over this chunk, acc[M..2M) += its slots per gene
int chunk_pass_sum(Ctx* ctx, const PatternDev& p, const float* val, int f32path, double* acc) {
  SCL_WS(ctx, part, double, "sc.part", 2 * p.M + 8);
  double *tgc, *lg;
  SCL_TRY(chunk_lg(ctx, p, val, f32path, &tgc, &lg, part));
  hipLaunchKernelGGL(k_axpy_f64, dim3((unsigned)((2 * p.M + 255) / 256)), dim3(256), 0, ctx->stream, part, 2 * p.M, acc);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}
// pass 2: acc_s2[j] += sum over this chunk's slots of (lg - mean_j)^2, mean_j = acc[j] / n_global
int chunk_pass_var(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* acc, double n_global, double* acc_s2) {
  SCL_WS(ctx, part, double, "sc.part", 2 * p.M + 8);
  SCL_WS(ctx, s2, double, "sc.s2", p.M);
  double *tgc, *lg;
  SCL_TRY(chunk_lg(ctx, p, val, f32path, &tgc, &lg, part));
  hipLaunchKernelGGL(k_sh_col_var, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, ctx->stream, p, lg, acc, n_global, s2, p.M);
  hipLaunchKernelGGL(k_axpy_f64, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, ctx->stream, s2, p.M, acc_s2);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}
// mean / std / mu of all cells from the accumulated sums; red[0] = ||mu||^2
int chunk_stats_finish(Ctx* ctx, int64_t M, const double* acc, const double* acc_s2, double n_global, int f32path, double* mean,
                       double* stdv, double* mu, double* red) {
  hipLaunchKernelGGL(k_sh_col_fin, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, acc, acc_s2, n_global, f32path, M, mean, stdv, mu);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, ctx->stream, mu, M, 0, red + 0);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}
// pass 3 / dense pass. The chunk's cells' norms l (-> l2_out, device, p.N doubles, optional), then the dense block
//   B[j][i] = srow_i (Z_ij - mu_j) - cent_j      with srow_i = num / l_i
// num = 1, cent = zeros: the unit-row form U of pass 3, together with accT[0..M) += sum_i lg_ij / l_i, accT[M] += sum_i 1 / l_i,
// accT[M + 1] += sum_i l_i (accT != nullptr); num = c = mean(l), cent = the finished centring vector: the scaled matrix itself.
int chunk_dense(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* stdv, const double* mu, const double* red,
                double num, const double* cent, double* accT, float* B, int64_t ldb, double** tgc_out, double** l2_out, double** lg_out,
                double** srow_out) {
  const int64_t N = p.N, M = p.M;
  SCL_WS(ctx, part, double, "sc.part", 2 * M + 8);
  SCL_WS(ctx, l2, double, "sc.l2", N);
  SCL_WS(ctx, srow, double, "sc.srow", N);
  double *tgc, *lg;
  SCL_TRY(chunk_lg(ctx, p, val, f32path, &tgc, &lg, part));
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p, val, tgc, f32path, lg, stdv, mu, red + 0, l2);
  hipLaunchKernelGGL(k_inv_f64, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, l2, N, num, srow);
  if (accT) {
    hipLaunchKernelGGL(k_sh_col_cent_part, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, p, lg, srow, part, M);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, srow, N, 1, part + M);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, l2, N, 1, part + M + 1);
    hipLaunchKernelGGL(k_axpy_f64, dim3((unsigned)((M + 2 + 255) / 256)), dim3(256), 0, st, part, M + 2, accT);
  }
  if (B) {
    if (ldb % 4 != 0 || (reinterpret_cast<uintptr_t>(B) & 15u) != 0) return ctx->fail(SCLENS_ERR_ARG, "chunk_dense: unaligned block");
    hipLaunchKernelGGL(k_dense_fused, dim3((unsigned)M), dim3(256), 0, st, p, val, lg, stdv, mu, cent, srow, B, ldb);
  }
  SCL_HIP(ctx, hipGetLastError());
  if (tgc_out) *tgc_out = tgc;
  if (l2_out) *l2_out = l2;
  if (lg_out) *lg_out = lg;
  if (srow_out) *srow_out = srow;
  return SCLENS_OK;
}
// cent_j = (c T_j / std_j - mu_j c T_M) / N with c = T_{M+1} / N (all on the device: no host round trip), then
// A = c^2 A - (N / divisor) cent cent' (A holds sum_g U_g'U_g / divisor; fp64 per element, symmetric in (j, k))
__global__ void k_chunk_cent_fin(const double* __restrict__ T, const double* __restrict__ stdv, const double* __restrict__ mu,
                                 double n_global, int64_t M, double* __restrict__ cent) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const double c = T[M + 1] / n_global;
  cent[j] = (c * T[j] / stdv[j] - mu[j] * c * T[M]) / n_global;
}
__global__ void k_chunk_gram_fin(float* __restrict__ A, int64_t n, int64_t lda, const double* __restrict__ T, const double* __restrict__ cent_rows,
                                 const double* __restrict__ cent, double n_global, double nd) {
  const int64_t j = blockIdx.y;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double c = T[n + 1] / n_global;
  A[j * lda + k] = (float)(c * c * (double)A[j * lda + k] - nd * (cent_rows[j] * cent[k]));
}
int chunk_gram_finish(Ctx* ctx, float* A, int64_t n, int64_t lda, const double* T, const double* stdv, const double* mu, double n_global,
                      double divisor, double* cent) {
  hipLaunchKernelGGL(k_chunk_cent_fin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, T, stdv, mu, n_global, n, cent);
  for (int64_t r0 = 0; r0 < n; r0 += 65535) {
    const int64_t rows = std::min<int64_t>(65535, n - r0);
    hipLaunchKernelGGL(k_chunk_gram_fin, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream, A + r0 * lda, n, lda, T,
                       cent + r0, cent, n_global, n_global / divisor);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// ---- value arrays over the pattern ------------------------------------------------------------------
// out[q] = binary ? (base[q] != 0) : base[q]     (candidate slots have base 0); the CSR-ordered copy behind it likewise
__global__ void k_val_init(const float* __restrict__ base, const float* __restrict__ base_csr, int64_t nU, int binary,
                           float* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nU) {
    const float b = base[q];
    out[q] = binary ? (b != 0.f ? 1.f : 0.f) : b;
    if (base_csr) {
      const float c = base_csr[q];
      out[nU + q] = binary ? (c != 0.f ? 1.f : 0.f) : c;
    }
  }
}
// out[cand_pos[idx[t] - cand_off]] = 1 for the sampled candidates this pattern holds: indices outside [cand_off, cand_off + ncand)
// belong to other ranks of a row-sharded session (or are out of range) and are ignored rather than dereferenced
__global__ void k_val_set_ones(const uint32_t* __restrict__ idx, int64_t m, const int64_t* __restrict__ cand_pos,
                               const int64_t* __restrict__ cand_pos_csr, int64_t nU, int64_t cand_off, int64_t ncand,
                               float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) {
    const int64_t g = (int64_t)idx[t] - cand_off;
    if (g >= 0 && g < ncand) {
      const int64_t pos = cand_pos[g];
      if (pos >= 0) {  // -1: the candidate's cell belongs to another rank (row-sharded session, global list)
        out[pos] = 1.f;
        if (cand_pos_csr) out[nU + cand_pos_csr[g]] = 1.f;
      }
    }
  }
}

int make_values(Ctx* ctx, const PatternDev& p, const float* base_val, int binary, const uint32_t* idx_dev, int64_t m,
                float* out) {
  hipLaunchKernelGGL(k_val_init, dim3((unsigned)((p.nU + 255) / 256)), dim3(256), 0, ctx->stream, base_val, p.base_val_csr, p.nU,
                     binary, out);
  if (m > 0 && p.ncand > 0)
    hipLaunchKernelGGL(k_val_set_ones, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, idx_dev, m,
                       p.cand_pos, p.cand_pos_csr, p.nU, p.cand_off, p.ncand, out);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

__global__ void k_val_set_ones_feistel(FeistelPerm perm, int64_t m, const int64_t* __restrict__ cand_pos,
                                       const int64_t* __restrict__ cand_pos_csr, int64_t nU, int64_t cand_off, int64_t ncand,
                                       float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) {
    const int64_t g = (int64_t)feistel_apply(perm, (uint64_t)t) - cand_off;
    if (g >= 0 && g < ncand) {
      const int64_t pos = cand_pos[g];
      if (pos >= 0) {
        out[pos] = 1.f;
        if (cand_pos_csr) out[nU + cand_pos_csr[g]] = 1.f;
      }
    }
  }
}

int make_values_seeded(Ctx* ctx, const PatternDev& p, const float* base_val, int binary, uint64_t seed, int64_t m,
                       float* out) {
  if (m < 0 || m > p.population()) return ctx->fail(SCLENS_ERR_ARG, "make_values_seeded: bad sample size");
  hipLaunchKernelGGL(k_val_init, dim3((unsigned)((p.nU + 255) / 256)), dim3(256), 0, ctx->stream, base_val, p.base_val_csr, p.nU,
                     binary, out);
  if (m > 0) {
    // the permutation is over the GLOBAL candidate list: every rank of a row-sharded session evaluates the same sample and
    // keeps the part that falls into its own window
    const FeistelPerm perm = feistel_make((uint64_t)p.population(), seed);
    hipLaunchKernelGGL(k_val_set_ones_feistel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, perm, m,
                       p.cand_pos, p.cand_pos_csr, p.nU, p.cand_off, p.ncand, out);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// ---- small helpers ----------------------------------------------------------------------------------
__global__ void k_fill(float* p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int fill_f32(Ctx* ctx, float* p, int64_t n, float v) {
  if (n <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, n, v);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

__global__ __launch_bounds__(256) void k_normalize_rows(float* A, int64_t rows, int64_t cols, int64_t ld) {
  __shared__ double sw[4];
  const int64_t r = blockIdx.x;
  float* a = A + r * ld;
  double s = 0.0;
  for (int64_t c = threadIdx.x; c < cols; c += 256) s += (double)a[c] * (double)a[c];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  const double inv = 1.0 / sqrt(sw[0] + sw[1] + sw[2] + sw[3]);
  for (int64_t c = threadIdx.x; c < cols; c += 256) a[c] = (float)((double)a[c] * inv);
}
int normalize_rows_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld) {
  if (rows <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_normalize_rows, dim3((unsigned)rows), dim3(256), 0, ctx->stream, A, rows, cols, ld);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// squared row norms (fp64) and the matching rescale: normalize_rows_f32 in two halves, so that a row-sharded session can
// sum the squared norms over the ranks in between
__global__ __launch_bounds__(256) void k_row_sqnorms(const float* A, int64_t cols, int64_t ld, double* out) {
  __shared__ double sw[4];
  const float* a = A + (int64_t)blockIdx.x * ld;
  double s = 0.0;
  for (int64_t c = threadIdx.x; c < cols; c += 256) s += (double)a[c] * (double)a[c];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ void k_scale_rows_rsqrt(float* A, int64_t cols, int64_t ld, const double* sq) {
  const int64_t r = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cols) A[r * ld + c] = (float)((double)A[r * ld + c] / sqrt(sq[r]));
}
int row_sqnorms_f32(Ctx* ctx, const float* A, int64_t rows, int64_t cols, int64_t ld, double* out) {
  if (rows <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_row_sqnorms, dim3((unsigned)rows), dim3(256), 0, ctx->stream, A, cols, ld, out);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}
int scale_rows_rsqrt_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld, const double* sq) {
  if (rows <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_scale_rows_rsqrt, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream, A,
                     cols, ld, sq);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

__global__ void k_scale_rows(float* A, int64_t rows, int64_t cols, int64_t ld, const float* s) {
  const int64_t r = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cols) A[r * ld + c] *= s[r];
}
int scale_rows_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld, const float* s) {
  if (rows <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream,
                     A, rows, cols, ld, s);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

}  // namespace scl
