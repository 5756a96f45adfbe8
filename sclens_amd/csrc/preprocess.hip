// QC filtering of a raw count matrix on the device: the reference's `preprocess` (scLENS.jl:160-236), the step
// immediately before sclens(). The reference materialises Boolean matrices (`X .!= 0`), row/column sums
// and a fancy-indexed copy `X[fc_idx, fg_idx]` on the host; here the O(nnz) passes run over the uploaded CSC and only
// the O(N + M) mask / sort / prefix-sum logic stays on the host:
//   pass 1  per gene : cells expressing it, total counts                       (k_pp_gene_stats)
//   pass 2  per cell : genes expressed, total, mitochondrial and ribosomal counts (k_pp_cell_stats, integer + fp64 atomics)
//   masks            : fg_idx (:185-189), fc_idx (:191-215)                      (host, from the downloaded vectors)
//   pass 3  per kept gene : entries and total over the kept cells                (k_pp_gene_kept) -> drop all-zero genes
//                      (:219-220), order by mean ascending, ties by original order (:223 sortperm is stable)
//   pass 4  gather   : filtered CSC in the new gene order, cells renumbered      (k_pp_gather, order-preserving)
// Sums are accumulated in fp64: for count data (values with <= 24 significant bits within a 2^29 dynamic range) every
// partial sum is exact, so the atomics' order does not matter and the result equals the reference's Float32 sums
// wherever those are exact. Ratios are formed in Float32 and compared in Float64 as Julia does (:201, :207).
#include <algorithm>
#include <numeric>

#include "common.h"
#include "pattern.h"

namespace scl {

__device__ __forceinline__ double pp_wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int64_t pp_wsum_i(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per gene: n_cell_counts (:181), n_cell_counts_sum (:182)
__global__ __launch_bounds__(256) void k_pp_gene_stats(int64_t M, const int64_t* __restrict__ colptr,
                                                       const float* __restrict__ val, int64_t* __restrict__ cnt,
                                                       double* __restrict__ sum) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= M) return;
  int64_t c = 0;
  double s = 0.0;
  for (int64_t q = colptr[col] + lane; q < colptr[col + 1]; q += 64) {
    const float v = val[q];
    c += (v != 0.f);
    s += (double)v;
  }
  c = pp_wsum_i(c);
  s = pp_wsum(s);
  if (lane == 0) {
    cnt[col] = c;
    sum[col] = s;
  }
}

// one wave per gene, scattering into the per-cell accumulators: n_gene_counts (:188), n_gene_counts_sum (:189),
// mitochondrial (:198) and ribosomal (:204) totals
__global__ __launch_bounds__(256) void k_pp_cell_stats(int64_t M, const int64_t* __restrict__ colptr,
                                                       const int32_t* __restrict__ row, const float* __restrict__ val,
                                                       const uint8_t* __restrict__ is_mito,
                                                       const uint8_t* __restrict__ is_ribo, int* __restrict__ cnt,
                                                       double* __restrict__ sum, double* __restrict__ mito,
                                                       double* __restrict__ ribo) {
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= M) return;
  const bool mt = is_mito && is_mito[col], rb = is_ribo && is_ribo[col];
  for (int64_t q = colptr[col] + lane; q < colptr[col + 1]; q += 64) {
    const float v = val[q];
    if (v == 0.f) continue;
    const int32_t r = row[q];
    atomicAdd(&cnt[r], 1);
    atomicAdd(&sum[r], (double)v);
    if (mt) atomicAdd(&mito[r], (double)v);
    if (rb) atomicAdd(&ribo[r], (double)v);
  }
}

// one wave per kept gene: entries and total over the kept cells
__global__ __launch_bounds__(256) void k_pp_gene_kept(int64_t nkeep, const int64_t* __restrict__ genes,
                                                      const int64_t* __restrict__ colptr, const int32_t* __restrict__ row,
                                                      const float* __restrict__ val, const int32_t* __restrict__ rowmap,
                                                      int64_t* __restrict__ cnt, double* __restrict__ sum) {
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= nkeep) return;
  const int64_t col = genes[g];
  int64_t c = 0;
  double s = 0.0;
  for (int64_t q = colptr[col] + lane; q < colptr[col + 1]; q += 64) {
    const float v = val[q];
    if (v != 0.f && rowmap[row[q]] >= 0) {
      c += 1;
      s += (double)v;
    }
  }
  c = pp_wsum_i(c);
  s = pp_wsum(s);
  if (lane == 0) {
    cnt[g] = c;
    sum[g] = s;
  }
}

// one wave per output gene: order-preserving compaction of the kept entries (rows stay ascending)
__global__ __launch_bounds__(256) void k_pp_gather(int64_t nout, const int64_t* __restrict__ genes,
                                                   const int64_t* __restrict__ colptr, const int32_t* __restrict__ row,
                                                   const float* __restrict__ val, const int32_t* __restrict__ rowmap,
                                                   const int64_t* __restrict__ out_colptr, int32_t* __restrict__ out_row,
                                                   float* __restrict__ out_val) {
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= nout) return;
  const int64_t col = genes[g];
  const int64_t b = colptr[col], e = colptr[col + 1];
  int64_t dst = out_colptr[g];
  for (int64_t q0 = b; q0 < e; q0 += 64) {
    const int64_t q = q0 + lane;
    float v = 0.f;
    int32_t nr = -1;
    if (q < e) {
      v = val[q];
      nr = rowmap[row[q]];
    }
    const bool keep = (q < e) && v != 0.f && nr >= 0;
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int64_t pos = dst + __popcll(m & ((1ull << lane) - 1ull));
      out_row[pos] = nr;
      out_val[pos] = v;
    }
    dst += __popcll(m);
  }
}

int preprocess_stats(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                     const uint8_t* is_mito, const uint8_t* is_ribo, const PpParams& P, uint8_t* keep_cell,
                     int64_t* gene_order, int64_t* n_cells, int64_t* n_genes, int64_t* nnz_out) {
  ctx->pp.valid = false;
  if (N <= 0 || M <= 0 || !colptr || !rowval || !nzval || !keep_cell || !gene_order || !n_cells || !n_genes || !nnz_out)
    return ctx->fail(SCLENS_ERR_ARG, "preprocess: bad arguments");
  if (N > 0x7fffffffLL) return ctx->fail(SCLENS_ERR_ARG, "preprocess: more than 2^31-1 cells");
  const int64_t nnz = colptr[M];
  if (colptr[0] != 0 || nnz < 0) return ctx->fail(SCLENS_ERR_ARG, "preprocess: colptr must start at 0");
  for (int64_t j = 0; j < M; ++j)
    if (colptr[j + 1] < colptr[j]) return ctx->fail(SCLENS_ERR_ARG, "preprocess: colptr must be non-decreasing");
  for (int64_t q = 0; q < nnz; ++q)
    if (rowval[q] < 0 || rowval[q] >= N) return ctx->fail(SCLENS_ERR_ARG, "preprocess: row index out of range");
  hipStream_t st = ctx->stream;
  SCL_WS(ctx, d_colptr, int64_t, "pp.colptr", M + 1);
  SCL_WS(ctx, d_row, int32_t, "pp.row", nnz);
  SCL_WS(ctx, d_val, float, "pp.val", nnz);
  SCL_WS(ctx, d_mito, uint8_t, "pp.ismito", M);
  SCL_WS(ctx, d_ribo, uint8_t, "pp.isribo", M);
  SCL_WS(ctx, d_gcnt, int64_t, "pp.gcnt", M);
  SCL_WS(ctx, d_gsum, double, "pp.gsum", M);
  SCL_WS(ctx, d_ccnt, int, "pp.ccnt", N);
  SCL_WS(ctx, d_csum, double, "pp.csum", 3 * N);  // total | mito | ribo
  SCL_HIP(ctx, hipMemcpyAsync(d_colptr, colptr, sizeof(int64_t) * (M + 1), hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemcpyAsync(d_row, rowval, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemcpyAsync(d_val, nzval, sizeof(float) * nnz, hipMemcpyHostToDevice, st));
  if (is_mito) SCL_HIP(ctx, hipMemcpyAsync(d_mito, is_mito, M, hipMemcpyHostToDevice, st));
  if (is_ribo) SCL_HIP(ctx, hipMemcpyAsync(d_ribo, is_ribo, M, hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemsetAsync(d_ccnt, 0, sizeof(int) * N, st));
  SCL_HIP(ctx, hipMemsetAsync(d_csum, 0, sizeof(double) * 3 * N, st));
  const unsigned gcols = (unsigned)((M + 3) / 4);
  hipLaunchKernelGGL(k_pp_gene_stats, dim3(gcols), dim3(256), 0, st, M, d_colptr, d_val, d_gcnt, d_gsum);
  hipLaunchKernelGGL(k_pp_cell_stats, dim3(gcols), dim3(256), 0, st, M, d_colptr, d_row, d_val,
                     is_mito ? d_mito : nullptr, is_ribo ? d_ribo : nullptr, d_ccnt, d_csum, d_csum + N, d_csum + 2 * N);
  SCL_HIP(ctx, hipGetLastError());
  std::vector<int64_t> gcnt(M);
  std::vector<double> gsum(M), csum(3 * N);
  std::vector<int> ccnt(N);
  SCL_HIP(ctx, hipMemcpyAsync(gcnt.data(), d_gcnt, sizeof(int64_t) * M, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(gsum.data(), d_gsum, sizeof(double) * M, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(ccnt.data(), d_ccnt, sizeof(int) * N, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(csum.data(), d_csum, sizeof(double) * 3 * N, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));

  // ---- masks (scLENS.jl:183-215). Sums are Float32 in the reference: round before comparing
  std::vector<int64_t> kept_genes;
  for (int64_t j = 0; j < M; ++j) {
    const float s = (float)gsum[j];
    if ((double)s > P.min_tp_g && (double)s < P.max_tp_g && gcnt[j] >= P.min_cells_per_gene) kept_genes.push_back(j);
  }
  std::vector<int32_t> rowmap(N, -1);
  int64_t nc = 0;
  for (int64_t i = 0; i < N; ++i) {
    const float s = (float)csum[i];
    bool ok = (double)s > P.min_tp_c && (double)s < P.max_tp_c && ccnt[i] >= P.min_genes_per_cell;
    if (ok && P.mito_percent != 0.0) ok = (double)((float)csum[N + i] / s) < P.mito_percent / 100.0;      // NaN -> false
    if (ok && P.ribo_percent != 0.0) ok = (double)((float)csum[2 * N + i] / s) < P.ribo_percent / 100.0;
    if (ok && P.max_genes_per_cell != 0) ok = ccnt[i] < P.max_genes_per_cell;
    keep_cell[i] = ok ? 1 : 0;
    if (ok) rowmap[i] = (int32_t)nc++;
  }
  *n_cells = nc;
  *n_genes = 0;
  *nnz_out = 0;
  ctx->pp = Ctx::PpState{};
  if (nc == 0 || kept_genes.empty()) return SCLENS_OK;  // "There is no high quality cells and genes" (:233)

  // ---- pass 3: kept genes over kept cells
  const int64_t nk = (int64_t)kept_genes.size();
  SCL_WS(ctx, d_rowmap, int32_t, "pp.rowmap", N);
  SCL_WS(ctx, d_genes, int64_t, "pp.genes", M);
  SCL_WS(ctx, d_kcnt, int64_t, "pp.kcnt", M);
  SCL_WS(ctx, d_ksum, double, "pp.ksum", M);
  SCL_HIP(ctx, hipMemcpyAsync(d_rowmap, rowmap.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemcpyAsync(d_genes, kept_genes.data(), sizeof(int64_t) * nk, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_pp_gene_kept, dim3((unsigned)((nk + 3) / 4)), dim3(256), 0, st, nk, d_genes, d_colptr, d_row, d_val,
                     d_rowmap, d_kcnt, d_ksum);
  SCL_HIP(ctx, hipGetLastError());
  std::vector<int64_t> kcnt(nk);
  std::vector<double> ksum(nk);
  SCL_HIP(ctx, hipMemcpyAsync(kcnt.data(), d_kcnt, sizeof(int64_t) * nk, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(ksum.data(), d_ksum, sizeof(double) * nk, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  // drop genes that are all-zero over the kept cells (:219-220), order by mean ascending (:223; stable)
  std::vector<int64_t> idx;
  std::vector<float> mean(nk);
  for (int64_t g = 0; g < nk; ++g) {
    if ((float)ksum[g] != 0.f) idx.push_back(g);
    mean[g] = (float)ksum[g] / (float)nc;
  }
  std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return mean[a] < mean[b]; });
  const int64_t ng = (int64_t)idx.size();
  std::vector<int64_t> out_genes(ng), out_colptr(ng + 1, 0);
  for (int64_t t = 0; t < ng; ++t) {
    out_genes[t] = kept_genes[idx[t]];
    out_colptr[t + 1] = out_colptr[t] + kcnt[idx[t]];
    gene_order[t] = out_genes[t];
  }
  *n_genes = ng;
  *nnz_out = out_colptr[ng];
  if (ng == 0) return SCLENS_OK;
  SCL_WS(ctx, d_ocol, int64_t, "pp.ocolptr", M + 1);
  SCL_HIP(ctx, hipMemcpyAsync(d_genes, out_genes.data(), sizeof(int64_t) * ng, hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipMemcpyAsync(d_ocol, out_colptr.data(), sizeof(int64_t) * (ng + 1), hipMemcpyHostToDevice, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));  // host vectors go out of scope
  ctx->pp.valid = true;
  ctx->pp.N = N;
  ctx->pp.M = M;
  ctx->pp.n_cells = nc;
  ctx->pp.n_genes = ng;
  ctx->pp.nnz_out = out_colptr[ng];
  return SCLENS_OK;
}

// pass 4: the filtered matrix of the preceding preprocess_stats call, into caller-allocated CSC arrays
int preprocess_gather(Ctx* ctx, int64_t* out_colptr, int32_t* out_rowval, float* out_nzval) {
  if (!ctx->pp.valid) return ctx->fail(SCLENS_ERR_STATE, "preprocess_gather: no preceding preprocess call with a non-empty result");
  if (!out_colptr || !out_rowval || !out_nzval) return ctx->fail(SCLENS_ERR_ARG, "preprocess_gather: null output");
  const int64_t ng = ctx->pp.n_genes, nnz = ctx->pp.nnz_out;
  hipStream_t st = ctx->stream;
  auto ws = [&](const char* name) { return ctx->ws.at(name).first; };
  const int64_t* d_colptr = static_cast<const int64_t*>(ws("pp.colptr"));
  const int32_t* d_row = static_cast<const int32_t*>(ws("pp.row"));
  const float* d_val = static_cast<const float*>(ws("pp.val"));
  const int32_t* d_rowmap = static_cast<const int32_t*>(ws("pp.rowmap"));
  const int64_t* d_genes = static_cast<const int64_t*>(ws("pp.genes"));
  const int64_t* d_ocol = static_cast<const int64_t*>(ws("pp.ocolptr"));
  SCL_WS(ctx, d_orow, int32_t, "pp.orow", nnz);
  SCL_WS(ctx, d_oval, float, "pp.oval", nnz);
  hipLaunchKernelGGL(k_pp_gather, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, st, ng, d_genes, d_colptr, d_row, d_val,
                     d_rowmap, d_ocol, d_orow, d_oval);
  SCL_HIP(ctx, hipGetLastError());
  SCL_HIP(ctx, hipMemcpyAsync(out_colptr, d_ocol, sizeof(int64_t) * (ng + 1), hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(out_rowval, d_orow, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(out_nzval, d_oval, sizeof(float) * nnz, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  ctx->pp.valid = false;
  for (const char* name : {"pp.colptr", "pp.row", "pp.val", "pp.orow", "pp.oval"}) ctx->release(name);  // the large ones
  return SCLENS_OK;
}

// pass 4': the filtered matrix stays in HBM as a Counts object (the hand-over preprocess -> sclens of SURVEY 8f-3: what
// preprocess_gather would send to the host, 8 bytes per stored entry, and session_create would send straight back)
int preprocess_keep(Ctx* ctx, Counts** out) {
  if (!ctx->pp.valid) return ctx->fail(SCLENS_ERR_STATE, "preprocess_keep: no preceding preprocess call with a non-empty result");
  const int64_t ng = ctx->pp.n_genes, nnz = ctx->pp.nnz_out;
  hipStream_t st = ctx->stream;
  auto ws = [&](const char* name) { return ctx->ws.at(name).first; };
  const int64_t* d_colptr = static_cast<const int64_t*>(ws("pp.colptr"));
  const int32_t* d_row = static_cast<const int32_t*>(ws("pp.row"));
  const float* d_val = static_cast<const float*>(ws("pp.val"));
  const int32_t* d_rowmap = static_cast<const int32_t*>(ws("pp.rowmap"));
  const int64_t* d_genes = static_cast<const int64_t*>(ws("pp.genes"));
  const int64_t* d_ocol = static_cast<const int64_t*>(ws("pp.ocolptr"));
  Counts* c = new Counts();
  c->device = ctx->device; c->N = ctx->pp.n_cells; c->M = ng; c->nnz = nnz;
  if (pool_malloc((void**)&c->colptr, sizeof(int64_t) * (ng + 1)) != hipSuccess ||
      pool_malloc((void**)&c->row, sizeof(int32_t) * (nnz > 4 ? nnz : 4)) != hipSuccess ||
      pool_malloc((void**)&c->val, sizeof(float) * (nnz > 4 ? nnz : 4)) != hipSuccess) {
    counts_free(c, ctx);
    return ctx->fail(SCLENS_ERR_OOM, "preprocess_keep: out of device memory");
  }
  hipLaunchKernelGGL(k_pp_gather, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, st, ng, d_genes, d_colptr, d_row, d_val,
                     d_rowmap, d_ocol, c->row, c->val);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(c->colptr, d_ocol, sizeof(int64_t) * (ng + 1), hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    counts_free(c, ctx);
    return ctx->fail(SCLENS_ERR_HIP, std::string("preprocess_keep: ") + hipGetErrorString(e));
  }
  ctx->pp.valid = false;
  for (const char* name : {"pp.colptr", "pp.row", "pp.val"}) ctx->release(name);  // the raw matrix
  *out = c;
  return SCLENS_OK;
}

}  // namespace scl
