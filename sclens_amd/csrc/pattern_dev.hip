// The sparse pattern of a sclens() call built ON THE DEVICE: the zero-candidate draw of scLENS.jl:668-673
// (R1: nnz uniform (i, j) pairs, minus the stored entries, first occurrences in draw order) and the union pattern
// "stored counts + candidates" with its CSR view, from the counts' CSC alone. Replaces the host passes of pattern_build
// (session.hip) and sclens_draw_zero_candidates (rng.cpp) for sessions that hold all cells: the host uploads 8 nnz bytes of CSC
// instead of 28 bytes per union slot, and nothing of it runs on host threads (which a multi-GPU job shares between its ranks).
// Deterministic: stable radix sorts (rocPRIM), integer atomics, scans; the arrays are bit-identical to the host builder's
// (tests/test_gpu_pattern.py) and the candidate list to the host generator's (same counter-based draw, rng.h).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "pattern.h"
#include "rng.h"

namespace scl {

namespace {

struct DevTmp {  // temporaries of one build, freed on every exit path
  std::vector<void*> p;
  hipStream_t stream = nullptr;  // the building context's stream (set by the first get)
  ~DevTmp() {
    for (void* q : p) pool_free(q, stream);
  }
  template <typename T>
  T* get(Ctx* ctx, size_t count) {
    void* q = nullptr;
    stream = ctx->stream;
    if (pool_malloc(&q, std::max<size_t>(sizeof(T) * count, 16)) != hipSuccess) {
      ctx->fail(SCLENS_ERR_OOM, "pattern_build_device: out of device memory");
      return nullptr;
    }
    p.push_back(q);
    return static_cast<T*>(q);
  }
};

template <typename T>
T* keep(Ctx* ctx, PatternOwner* o, size_t count) {
  void* q = nullptr;
  if (pool_malloc(&q, std::max<size_t>(sizeof(T) * count, 16)) != hipSuccess) {
    ctx->fail(SCLENS_ERR_OOM, "pattern_build_device: out of device memory");
    return nullptr;
  }
  o->allocs.push_back(q);
  return static_cast<T*>(q);
}

int bits_for(uint64_t count) {  // bits needed for values in [0, count)
  int b = 1;
  while (b < 64 && (1ull << b) < count) ++b;
  return b;
}

// ---- R1 on the device -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nz_bitmap(const int64_t* __restrict__ cp, const int32_t* __restrict__ rv, int64_t N, int64_t M,
                                                   unsigned* __restrict__ bits, int* __restrict__ bad) {
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= M) return;
  for (int64_t s = cp[j] + (threadIdx.x & 63); s < cp[j + 1]; s += 64) {
    const int64_t i = rv[s];
    if (i < 0 || i >= N) {
      *bad = 1;
      continue;
    }
    const uint64_t key = (uint64_t)i + (uint64_t)j * (uint64_t)N;
    atomicOr(&bits[key >> 5], 1u << (key & 31));
  }
}
// key of draw t, or `invalid` when the pair is a stored entry
template <typename K>
__global__ void k_r1_keys(uint64_t seed, int64_t nnz, uint64_t N, uint64_t M, const unsigned* __restrict__ bits, K invalid,
                          K* __restrict__ keys, unsigned* __restrict__ tval) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  uint64_t i, j;
  r1_draw(seed, (uint64_t)t, N, M, &i, &j);
  const uint64_t key = i + j * N;
  keys[t] = ((bits[key >> 5] >> (key & 31)) & 1u) ? invalid : (K)key;
  tval[t] = (unsigned)t;
}
// first element of every run of equal valid keys in the (stably) sorted order = the first occurrence in draw order
template <typename K>
__global__ void k_r1_flag(const K* __restrict__ keys, const unsigned* __restrict__ tval, int64_t nnz, K invalid,
                          unsigned* __restrict__ flag) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nnz) return;
  const K k = keys[q];
  if (k != invalid && (q == 0 || keys[q - 1] != k)) flag[tval[q]] = 1u;
}
__global__ void k_r1_emit(uint64_t seed, int64_t nnz, uint64_t N, uint64_t M, const unsigned* __restrict__ flag,
                          const unsigned* __restrict__ pos, uint32_t* __restrict__ z1, uint32_t* __restrict__ z2) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz || !flag[t]) return;
  uint64_t i, j;
  r1_draw(seed, (uint64_t)t, N, M, &i, &j);
  z1[pos[t]] = (uint32_t)i;
  z2[pos[t]] = (uint32_t)j;
}

template <typename K>
int r1_draw_device(Ctx* ctx, DevTmp& tmp, uint64_t seed, int64_t nnz, int64_t N, int64_t M, const unsigned* bits, uint32_t* z1,
                   uint32_t* z2, int64_t* ncand) {
  hipStream_t st = ctx->stream;
  const uint64_t cells = (uint64_t)N * (uint64_t)M;
  const int nb = bits_for(cells);
  const K invalid = (K)((nb >= (int)(8 * sizeof(K))) ? ~(K)0 : ((K)1 << nb));  // sorts behind every valid key
  K* k0 = tmp.get<K>(ctx, nnz);
  K* k1 = tmp.get<K>(ctx, nnz);
  unsigned* v0 = tmp.get<unsigned>(ctx, nnz);
  unsigned* v1 = tmp.get<unsigned>(ctx, nnz);
  unsigned* flag = tmp.get<unsigned>(ctx, nnz + 1);
  unsigned* pos = tmp.get<unsigned>(ctx, nnz + 1);
  if (!k0 || !k1 || !v0 || !v1 || !flag || !pos) return SCLENS_ERR_OOM;
  const unsigned gb = (unsigned)((nnz + 255) / 256);
  hipLaunchKernelGGL((k_r1_keys<K>), dim3(gb), dim3(256), 0, st, seed, nnz, (uint64_t)N, (uint64_t)M, bits, invalid, k0, v0);
  size_t bytes = 0;
  const unsigned end_bit = (unsigned)std::min<int>(nb + 1, 8 * (int)sizeof(K));
  SCL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, k0, k1, v0, v1, (size_t)nnz, 0u, end_bit, st));
  void* ws = tmp.get<char>(ctx, bytes);
  if (!ws) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, rocprim::radix_sort_pairs(ws, bytes, k0, k1, v0, v1, (size_t)nnz, 0u, end_bit, st));
  SCL_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned) * (nnz + 1), st));
  hipLaunchKernelGGL((k_r1_flag<K>), dim3(gb), dim3(256), 0, st, k1, v1, nnz, invalid, flag);
  size_t b2 = 0;
  SCL_HIP(ctx, rocprim::exclusive_scan(nullptr, b2, flag, pos, 0u, (size_t)nnz + 1, rocprim::plus<unsigned>(), st));
  void* ws2 = tmp.get<char>(ctx, b2);
  if (!ws2) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, rocprim::exclusive_scan(ws2, b2, flag, pos, 0u, (size_t)nnz + 1, rocprim::plus<unsigned>(), st));
  hipLaunchKernelGGL(k_r1_emit, dim3(gb), dim3(256), 0, st, seed, nnz, (uint64_t)N, (uint64_t)M, flag, pos, z1, z2);
  unsigned total = 0;
  SCL_HIP(ctx, hipMemcpyAsync(&total, pos + nnz, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  *ncand = (int64_t)total;
  return SCLENS_OK;
}

// ---- R1 for a BLOCK of cells (row-sharded sessions) -----------------------------------------------------------
// Draw t of the global list is a pure function of (seed, t) (r1_draw), so the rank that holds the cells [row0, row0 + Nl) can
// take its own candidates out of the global draw sequence without seeing anybody else's: every rank evaluates all nnz_global
// draws (two splitmix64 outputs each: ~10 ms per 10^9 on this device) in chunks, keeps those that land in its cells, and removes
// stored entries and repeats among them exactly as the global procedure would (a repeat has the same cell, hence the same
// rank). The local list is the global first-occurrence list restricted to the block, in the same order.
template <typename K>
__global__ void k_r1_block_hits(uint64_t seed, uint64_t t0, int64_t cnt, uint64_t Ng, uint64_t M, uint64_t row0, uint64_t Nl,
                                unsigned* __restrict__ flag) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  uint64_t i, j;
  r1_draw(seed, t0 + (uint64_t)q, Ng, M, &i, &j);
  flag[q] = (i >= row0 && i < row0 + Nl) ? 1u : 0u;
}
template <typename K>
__global__ void k_r1_block_emit(uint64_t seed, uint64_t t0, int64_t cnt, uint64_t Ng, uint64_t M, uint64_t row0, uint64_t Nl,
                                const unsigned* __restrict__ flag, const unsigned* __restrict__ pos, int64_t base,
                                const unsigned* __restrict__ bits, K invalid, K* __restrict__ keys, unsigned* __restrict__ seq) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt || !flag[q]) return;
  uint64_t i, j;
  r1_draw(seed, t0 + (uint64_t)q, Ng, M, &i, &j);
  const uint64_t key = (i - row0) + j * Nl;  // local key
  const int64_t o = base + pos[q];
  keys[o] = ((bits[key >> 5] >> (key & 31)) & 1u) ? invalid : (K)key;
  seq[o] = (unsigned)o;  // position in the (draw-ordered) local hit list
}
template <typename K>
__global__ void k_r1_block_out(const K* __restrict__ keys_in_draw_order, const unsigned* __restrict__ flag, const unsigned* __restrict__ pos,
                               int64_t nh, uint64_t Nl, uint32_t* __restrict__ z1, uint32_t* __restrict__ z2) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nh || !flag[q]) return;
  const uint64_t key = (uint64_t)keys_in_draw_order[q];
  z1[pos[q]] = (uint32_t)(key % Nl);  // LOCAL cell index
  z2[pos[q]] = (uint32_t)(key / Nl);
}

// z1 (local cell indices) / z2 of the block's candidates; *z1o / *z2o are allocated here (kept in `out`); *ncand = their number
template <typename K>
int r1_draw_device_block(Ctx* ctx, PatternOwner* out, uint64_t seed, int64_t nnz_global, int64_t Ng, int64_t M, int64_t row0, int64_t Nl,
                         const unsigned* bits, uint32_t** z1o, uint32_t** z2o, int64_t* ncand) {
  hipStream_t st = ctx->stream;
  const uint64_t cells = (uint64_t)Nl * (uint64_t)M;
  const int nb = bits_for(cells);
  const K invalid = (K)((nb >= (int)(8 * sizeof(K))) ? ~(K)0 : ((K)1 << nb));
  // expected hits: nnz_global * Nl / Ng; capacity with 6 sigma + slack, grown on overflow (never observed)
  const double expect = (double)nnz_global * (double)Nl / (double)Ng;
  int64_t cap = (int64_t)(expect + 6.0 * std::sqrt(expect + 1.0) + 4096.0);
  const int64_t CH = (int64_t)1 << 26;  // draws per chunk
  DevTmp tmp;
  unsigned* cflag = tmp.get<unsigned>(ctx, CH + 1);
  unsigned* cpos = tmp.get<unsigned>(ctx, CH + 1);
  K* keys = tmp.get<K>(ctx, cap);
  unsigned* seq = tmp.get<unsigned>(ctx, cap);
  if (!cflag || !cpos || !keys || !seq) return SCLENS_ERR_OOM;
  size_t sb = 0;
  SCL_HIP(ctx, rocprim::exclusive_scan(nullptr, sb, cflag, cpos, 0u, (size_t)CH + 1, rocprim::plus<unsigned>(), st));
  void* sws = tmp.get<char>(ctx, sb);
  if (!sws) return SCLENS_ERR_OOM;
  int64_t nh = 0;
  for (int64_t t0 = 0; t0 < nnz_global; t0 += CH) {
    const int64_t cnt = std::min<int64_t>(CH, nnz_global - t0);
    const unsigned gb = (unsigned)((cnt + 255) / 256);
    SCL_HIP(ctx, hipMemsetAsync(cflag + cnt, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL((k_r1_block_hits<K>), dim3(gb), dim3(256), 0, st, seed, (uint64_t)t0, cnt, (uint64_t)Ng, (uint64_t)M, (uint64_t)row0,
                       (uint64_t)Nl, cflag);
    SCL_HIP(ctx, rocprim::exclusive_scan(sws, sb, cflag, cpos, 0u, (size_t)cnt + 1, rocprim::plus<unsigned>(), st));
    unsigned got = 0;
    SCL_HIP(ctx, hipMemcpyAsync(&got, cpos + cnt, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
    if (nh + (int64_t)got > cap) return ctx->fail(SCLENS_ERR_HIP, "r1_draw_device_block: more local draws than 6 sigma above expectation");
    hipLaunchKernelGGL((k_r1_block_emit<K>), dim3(gb), dim3(256), 0, st, seed, (uint64_t)t0, cnt, (uint64_t)Ng, (uint64_t)M, (uint64_t)row0,
                       (uint64_t)Nl, cflag, cpos, nh, bits, invalid, keys, seq);
    nh += (int64_t)got;
  }
  *ncand = 0;
  *z1o = *z2o = nullptr;
  if (nh == 0) return SCLENS_OK;
  if (nh >= 0xFFFFFFF0ll) return ctx->fail(SCLENS_ERR_ARG, "r1_draw_device_block: more than 2^32 local draws");
  // first occurrences among the valid keys: stable sort by key (ties keep draw order), run heads, back to draw order by `seq`
  K* k1 = tmp.get<K>(ctx, nh);
  unsigned* v1 = tmp.get<unsigned>(ctx, nh);
  unsigned* flag = tmp.get<unsigned>(ctx, nh + 1);
  unsigned* pos = tmp.get<unsigned>(ctx, nh + 1);
  if (!k1 || !v1 || !flag || !pos) return SCLENS_ERR_OOM;
  size_t bytes = 0;
  const unsigned end_bit = (unsigned)std::min<int>(nb + 1, 8 * (int)sizeof(K));
  SCL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, keys, k1, seq, v1, (size_t)nh, 0u, end_bit, st));
  void* ws = tmp.get<char>(ctx, bytes);
  if (!ws) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, rocprim::radix_sort_pairs(ws, bytes, keys, k1, seq, v1, (size_t)nh, 0u, end_bit, st));
  SCL_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned) * (nh + 1), st));
  const unsigned gh = (unsigned)((nh + 255) / 256);
  hipLaunchKernelGGL((k_r1_flag<K>), dim3(gh), dim3(256), 0, st, k1, v1, nh, invalid, flag);  // flag[seq] = 1 for run heads
  size_t b2 = 0;
  SCL_HIP(ctx, rocprim::exclusive_scan(nullptr, b2, flag, pos, 0u, (size_t)nh + 1, rocprim::plus<unsigned>(), st));
  void* ws2 = tmp.get<char>(ctx, b2);
  if (!ws2) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, rocprim::exclusive_scan(ws2, b2, flag, pos, 0u, (size_t)nh + 1, rocprim::plus<unsigned>(), st));
  unsigned total = 0;
  SCL_HIP(ctx, hipMemcpyAsync(&total, pos + nh, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  *ncand = (int64_t)total;
  if (total == 0) return SCLENS_OK;
  uint32_t* z1 = keep<uint32_t>(ctx, out, total);
  uint32_t* z2 = keep<uint32_t>(ctx, out, total);
  if (!z1 || !z2) return SCLENS_ERR_OOM;
  hipLaunchKernelGGL((k_r1_block_out<K>), dim3(gh), dim3(256), 0, st, keys, flag, pos, nh, (uint64_t)Nl, z1, z2);
  SCL_HIP(ctx, hipStreamSynchronize(st));  // `tmp` dies with this frame
  *z1o = z1;
  *z2o = z2;
  return SCLENS_OK;
}

// ---- union pattern ----------------------------------------------------------------------------------------------------------
__global__ void k_count_u32(const uint32_t* __restrict__ key, int64_t n, uint32_t limit, unsigned* __restrict__ cnt, int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t k = key[t];
  if (k >= limit) {
    *bad = 1;
    return;
  }
  atomicAdd(&cnt[k], 1u);
}
__global__ void k_check_lt(const uint32_t* __restrict__ key, int64_t n, uint32_t limit, int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && key[t] >= limit) *bad = 1;
}
__global__ void k_col_totals(const int64_t* __restrict__ cp, const unsigned* __restrict__ cc, int64_t M, int64_t* __restrict__ tot) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M) tot[j] = (cp[j + 1] - cp[j]) + (int64_t)cc[j];
  if (j == M) tot[j] = 0;
}
__global__ __launch_bounds__(256) void k_fill_counts(const int64_t* __restrict__ cp, const int32_t* __restrict__ rv, const float* __restrict__ nz,
                                                     const int64_t* __restrict__ ucol, int64_t N, int64_t M, int32_t* __restrict__ urow,
                                                     float* __restrict__ uval, int* __restrict__ bad) {
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= M) return;
  const int64_t b = cp[j], e = cp[j + 1], u = ucol[j];
  for (int64_t s = b + (threadIdx.x & 63); s < e; s += 64) {
    const int32_t r = rv[s];
    if (r < 0 || r >= N) *bad = 1;
    urow[u + (s - b)] = r;
    uval[u + (s - b)] = nz[s];
  }
}
__global__ void k_iota_u32(unsigned* __restrict__ v, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) v[t] = (unsigned)t;
}
// candidates sorted by gene (stable: list order inside a gene): slot of candidate t
__global__ void k_place_cands(const uint32_t* __restrict__ kg, const unsigned* __restrict__ vt, int64_t ncand, const int64_t* __restrict__ cp,
                              const int64_t* __restrict__ ucol, const int64_t* __restrict__ cstart, const uint32_t* __restrict__ z1,
                              int64_t* __restrict__ cpos, int32_t* __restrict__ urow, float* __restrict__ uval) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= ncand) return;
  const int64_t j = kg[q], t = vt[q];
  const int64_t pos = ucol[j] + (cp[j + 1] - cp[j]) + (q - cstart[j]);
  cpos[t] = pos;
  urow[pos] = (int32_t)z1[t];
  uval[pos] = 0.f;
}
__global__ void k_copy_rows_u32(const int32_t* __restrict__ urow, int64_t nU, unsigned* __restrict__ key, unsigned* __restrict__ cnt) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nU) return;
  const unsigned r = (unsigned)urow[q];
  key[q] = r;
  atomicAdd(&cnt[r], 1u);
}
__global__ void k_widen_cnt(const unsigned* __restrict__ cnt, int64_t n, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int64_t)cnt[i];
  if (i == n) out[i] = 0;
}
// CSR slot s holds CSC slot v[s]; its column = the gene whose slot range contains it
__global__ void k_csr_finish(const unsigned* __restrict__ v, int64_t nU, const int64_t* __restrict__ ucol, int64_t M,
                             int64_t* __restrict__ c2c, int32_t* __restrict__ ccol) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nU) return;
  const int64_t q = (int64_t)v[s];
  int64_t lo = 0, hi = M;  // largest j with ucol[j] <= q
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (ucol[mid] <= q) lo = mid;
    else hi = mid;
  }
  c2c[s] = q;
  ccol[s] = (int32_t)lo;
}

int scan_i64(Ctx* ctx, DevTmp& tmp, const int64_t* in, int64_t* out, size_t n) {
  size_t bytes = 0;
  SCL_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), ctx->stream));
  void* ws = tmp.get<char>(ctx, bytes);
  if (!ws) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, rocprim::exclusive_scan(ws, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), ctx->stream));
  return SCLENS_OK;
}

}  // namespace

namespace {
__global__ void k_csr_gather_base(const float* __restrict__ base, const int64_t* __restrict__ c2c, int64_t nU, float* __restrict__ base_csr,
                                  int64_t* __restrict__ inv) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nU) return;
  const int64_t q = c2c[s];
  base_csr[s] = base[q];
  inv[q] = s;
}
__global__ void k_csr_cand_pos(const int64_t* __restrict__ cpos, const int64_t* __restrict__ inv, int64_t ncand, int64_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ncand) return;
  const int64_t q = cpos[t];
  out[t] = q >= 0 ? inv[q] : -1;
}
}  // namespace

int pattern_add_csr_companions(Ctx* ctx, PatternOwner* out) {
  const bool off = ctx->opt.val_csr == 0;
  PatternDev& d = out->dev;
  // only patterns with candidates: those serve the S + P decompositions of the search and the ensemble; a counts-only pattern
  // (data / null / binarised matrix: one or two decompositions each) would pay more for the companions than it saves
  if (off || d.nU <= 0 || d.ncand <= 0 || !d.csr2csc || !out->base_val) return SCLENS_OK;
  hipStream_t st = ctx->stream;
  float* bc = keep<float>(ctx, out, d.nU);
  int64_t* cpc = keep<int64_t>(ctx, out, std::max<int64_t>(d.ncand, 1));
  DevTmp tmp;
  int64_t* inv = tmp.get<int64_t>(ctx, d.nU);
  if (!bc || !cpc || !inv) return SCLENS_ERR_OOM;
  hipLaunchKernelGGL(k_csr_gather_base, dim3((unsigned)((d.nU + 255) / 256)), dim3(256), 0, st, out->base_val, d.csr2csc, d.nU, bc, inv);
  if (d.ncand > 0)
    hipLaunchKernelGGL(k_csr_cand_pos, dim3((unsigned)((d.ncand + 255) / 256)), dim3(256), 0, st, d.cand_pos, inv, d.ncand, cpc);
  SCL_HIP(ctx, hipGetLastError());
  SCL_HIP(ctx, hipStreamSynchronize(st));  // `tmp` dies with this frame
  d.base_val_csr = bc;
  d.cand_pos_csr = cpc;
  return SCLENS_OK;
}

// draw != 0: the candidate list is drawn on the device from `seed` (z1_h / z2_h ignored); else it is uploaded from the host
// (ncand entries). The candidate list stays on the device in out->z1_dev / z2_dev (sclens_hip_pattern_candidates downloads it).
// nnz_dev >= 0: colptr / rowval / nzval are DEVICE arrays of a count matrix with nnz_dev stored entries (a sclens_hip_counts
// handle: the output of the QC filter that stayed in HBM) and are read in place; otherwise host arrays, uploaded.
int pattern_build_device(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t ncand,
                         const uint32_t* z1_h, const uint32_t* z2_h, int draw, uint64_t seed, PatternOwner* out, int64_t nnz_dev,
                         const BlockDraw* blk) {
  const bool src_dev = nnz_dev >= 0;
  if (blk && (!draw || blk->N_global < N || blk->row0 < 0 || blk->row0 + N > blk->N_global || blk->nnz_global < 0 ||
              blk->nnz_global >= 0xFFFFFFF0ll))
    return ctx->fail(SCLENS_ERR_ARG, "pattern_build_device: bad block-draw arguments");
  if (N <= 0 || M <= 0 || !colptr || (!src_dev && colptr[M] > 0 && (!rowval || !nzval)) || (src_dev && nnz_dev > 0 && (!rowval || !nzval)) ||
      (!draw && ncand > 0 && (!z1_h || !z2_h)))
    return ctx->fail(SCLENS_ERR_ARG, "pattern_build_device: bad arguments");
  const int64_t nnz = src_dev ? nnz_dev : colptr[M];
  if (nnz >= 0xFFFFFFF0ll || N >= 0x7FFFFFFFll || M >= 0x7FFFFFFFll)
    return ctx->fail(SCLENS_ERR_ARG, "pattern_build_device: more than 2^32 stored entries");
  hipStream_t st = ctx->stream;
  DevTmp tmp;
  const int64_t* cp = colptr;
  const int32_t* rv = rowval;
  const float* nz = nzval;
  int* bad = tmp.get<int>(ctx, 4);
  if (!bad) return SCLENS_ERR_OOM;
  if (!src_dev) {
    int64_t* cpu = tmp.get<int64_t>(ctx, M + 1);
    int32_t* rvu = tmp.get<int32_t>(ctx, nnz);
    float* nzu = tmp.get<float>(ctx, nnz);
    if (!cpu || !rvu || !nzu) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemcpyAsync(cpu, colptr, sizeof(int64_t) * (M + 1), hipMemcpyHostToDevice, st));
    if (nnz > 0) {
      SCL_HIP(ctx, hipMemcpyAsync(rvu, rowval, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, st));
      SCL_HIP(ctx, hipMemcpyAsync(nzu, nzval, sizeof(float) * nnz, hipMemcpyHostToDevice, st));
    }
    cp = cpu; rv = rvu; nz = nzu;
  }
  SCL_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int) * 4, st));
  const unsigned gcol = (unsigned)((M + 3) / 4);
  // ---- candidates
  uint32_t *z1 = nullptr, *z2 = nullptr;
  if (draw && blk) {  // this rank's part of the global draw (row-sharded session): z1 = LOCAL cell indices
    ncand = 0;
    const uint64_t cells = (uint64_t)N * (uint64_t)M;
    unsigned* bits = tmp.get<unsigned>(ctx, (cells + 31) / 32 + 1);
    if (!bits) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemsetAsync(bits, 0, sizeof(unsigned) * ((cells + 31) / 32 + 1), st));
    if (nnz > 0) hipLaunchKernelGGL(k_nz_bitmap, dim3(gcol), dim3(256), 0, st, cp, rv, N, M, bits, bad);
    if (cells < 0xFFFFFFF0ull)
      SCL_TRY((r1_draw_device_block<uint32_t>(ctx, out, seed, blk->nnz_global, blk->N_global, M, blk->row0, N, bits, &z1, &z2, &ncand)));
    else
      SCL_TRY((r1_draw_device_block<uint64_t>(ctx, out, seed, blk->nnz_global, blk->N_global, M, blk->row0, N, bits, &z1, &z2, &ncand)));
  } else if (draw) {
    ncand = 0;
    if (nnz > 0) {
      z1 = keep<uint32_t>(ctx, out, nnz);
      z2 = keep<uint32_t>(ctx, out, nnz);
      const uint64_t cells = (uint64_t)N * (uint64_t)M;
      unsigned* bits = tmp.get<unsigned>(ctx, (cells + 31) / 32 + 1);
      if (!z1 || !z2 || !bits) return SCLENS_ERR_OOM;
      SCL_HIP(ctx, hipMemsetAsync(bits, 0, sizeof(unsigned) * ((cells + 31) / 32 + 1), st));
      hipLaunchKernelGGL(k_nz_bitmap, dim3(gcol), dim3(256), 0, st, cp, rv, N, M, bits, bad);
      DevTmp t2;  // the draw's sort buffers go before the union build allocates its own
      if (cells < 0xFFFFFFF0ull)
        SCL_TRY((r1_draw_device<uint32_t>(ctx, t2, seed, nnz, N, M, bits, z1, z2, &ncand)));
      else
        SCL_TRY((r1_draw_device<uint64_t>(ctx, t2, seed, nnz, N, M, bits, z1, z2, &ncand)));
    }
  } else if (ncand > 0) {
    z1 = keep<uint32_t>(ctx, out, ncand);
    z2 = keep<uint32_t>(ctx, out, ncand);
    if (!z1 || !z2) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemcpyAsync(z1, z1_h, sizeof(uint32_t) * ncand, hipMemcpyHostToDevice, st));
    SCL_HIP(ctx, hipMemcpyAsync(z2, z2_h, sizeof(uint32_t) * ncand, hipMemcpyHostToDevice, st));
  }
  // ---- per-gene totals, slot ranges
  unsigned* cc = tmp.get<unsigned>(ctx, M + 1);
  int64_t* tot = tmp.get<int64_t>(ctx, M + 1);
  int64_t* cstart = tmp.get<int64_t>(ctx, M + 1);
  int64_t* ucol = keep<int64_t>(ctx, out, M + 1);
  if (!cc || !tot || !cstart || !ucol) return SCLENS_ERR_OOM;
  SCL_HIP(ctx, hipMemsetAsync(cc, 0, sizeof(unsigned) * (M + 1), st));
  if (ncand > 0) {
    hipLaunchKernelGGL(k_count_u32, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, z2, ncand, (uint32_t)M, cc, bad);
    hipLaunchKernelGGL(k_check_lt, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, z1, ncand, (uint32_t)N, bad + 1);
  }
  hipLaunchKernelGGL(k_col_totals, dim3((unsigned)((M + 256) / 256)), dim3(256), 0, st, cp, cc, M, tot);
  SCL_TRY(scan_i64(ctx, tmp, tot, ucol, (size_t)M + 1));
  hipLaunchKernelGGL(k_widen_cnt, dim3((unsigned)((M + 256) / 256)), dim3(256), 0, st, cc, M, tot);
  SCL_TRY(scan_i64(ctx, tmp, tot, cstart, (size_t)M + 1));
  int64_t nU = 0;
  int hbad[4] = {0, 0, 0, 0};
  SCL_HIP(ctx, hipMemcpyAsync(&nU, ucol + M, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipMemcpyAsync(hbad, bad, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  if (hbad[0] || hbad[1]) return ctx->fail(SCLENS_ERR_ARG, hbad[0] ? "row / gene index out of range" : "candidate index out of range");
  if (nU != nnz + ncand) return ctx->fail(SCLENS_ERR_HIP, "pattern_build_device: slot count mismatch");
  if (nU >= 0xFFFFFFF0ll) return ctx->fail(SCLENS_ERR_ARG, "pattern_build_device: more than 2^32 union slots");
  // ---- CSC fill
  int32_t* urow = keep<int32_t>(ctx, out, nU);
  float* uval = keep<float>(ctx, out, nU);
  int64_t* cpos = keep<int64_t>(ctx, out, ncand);
  if (!urow || !uval || !cpos) return SCLENS_ERR_OOM;
  hipLaunchKernelGGL(k_fill_counts, dim3(gcol), dim3(256), 0, st, cp, rv, nz, ucol, N, M, urow, uval, bad);
  if (ncand > 0) {
    DevTmp t2;
    unsigned* v0 = t2.get<unsigned>(ctx, ncand);
    unsigned* v1 = t2.get<unsigned>(ctx, ncand);
    uint32_t* k1 = t2.get<uint32_t>(ctx, ncand);
    if (!v0 || !v1 || !k1) return SCLENS_ERR_OOM;
    hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, v0, ncand);
    size_t bytes = 0;
    const unsigned eb = (unsigned)bits_for((uint64_t)M);
    SCL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, z2, k1, v0, v1, (size_t)ncand, 0u, eb, st));
    void* ws = t2.get<char>(ctx, bytes);
    if (!ws) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, rocprim::radix_sort_pairs(ws, bytes, z2, k1, v0, v1, (size_t)ncand, 0u, eb, st));
    hipLaunchKernelGGL(k_place_cands, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, k1, v1, ncand, cp, ucol, cstart, z1, cpos,
                       urow, uval);
    SCL_HIP(ctx, hipStreamSynchronize(st));
  }
  // a stored row index outside [0, N) must not reach k_copy_rows_u32 (it counts rows through the unchecked index)
  SCL_HIP(ctx, hipMemcpyAsync(hbad, bad, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  if (hbad[0]) return ctx->fail(SCLENS_ERR_ARG, "row index out of range");
  // ---- CSR view
  int64_t* rptr = keep<int64_t>(ctx, out, N + 1);
  int64_t* c2c = keep<int64_t>(ctx, out, nU);
  int32_t* ccol = keep<int32_t>(ctx, out, nU);
  if (!rptr || !c2c || !ccol) return SCLENS_ERR_OOM;
  {
    DevTmp t2;
    unsigned* rc = t2.get<unsigned>(ctx, N + 1);
    int64_t* rt = t2.get<int64_t>(ctx, N + 1);
    unsigned* k0 = t2.get<unsigned>(ctx, nU);
    unsigned* k1 = t2.get<unsigned>(ctx, nU);
    unsigned* v0 = t2.get<unsigned>(ctx, nU);
    unsigned* v1 = t2.get<unsigned>(ctx, nU);
    if (!rc || !rt || !k0 || !k1 || !v0 || !v1) return SCLENS_ERR_OOM;
    SCL_HIP(ctx, hipMemsetAsync(rc, 0, sizeof(unsigned) * (N + 1), st));
    const unsigned gU = (unsigned)((nU + 255) / 256);
    if (nU > 0) {
      hipLaunchKernelGGL(k_copy_rows_u32, dim3(gU), dim3(256), 0, st, urow, nU, k0, rc);
      hipLaunchKernelGGL(k_iota_u32, dim3(gU), dim3(256), 0, st, v0, nU);
    }
    hipLaunchKernelGGL(k_widen_cnt, dim3((unsigned)((N + 256) / 256)), dim3(256), 0, st, rc, N, rt);
    SCL_TRY(scan_i64(ctx, t2, rt, rptr, (size_t)N + 1));
    if (nU > 0) {
      size_t bytes = 0;
      const unsigned eb = (unsigned)bits_for((uint64_t)N);
      SCL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, k0, k1, v0, v1, (size_t)nU, 0u, eb, st));
      void* ws = t2.get<char>(ctx, bytes);
      if (!ws) return SCLENS_ERR_OOM;
      SCL_HIP(ctx, rocprim::radix_sort_pairs(ws, bytes, k0, k1, v0, v1, (size_t)nU, 0u, eb, st));
      hipLaunchKernelGGL(k_csr_finish, dim3(gU), dim3(256), 0, st, v1, nU, ucol, M, c2c, ccol);
    }
    SCL_HIP(ctx, hipGetLastError());
    SCL_HIP(ctx, hipMemcpyAsync(hbad, bad, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    SCL_HIP(ctx, hipStreamSynchronize(st));
    if (hbad[0]) return ctx->fail(SCLENS_ERR_ARG, "row index out of range");
  }
  out->dev.N = N; out->dev.M = M; out->dev.nU = nU; out->dev.ncand = ncand;
  out->dev.colptr = ucol; out->dev.row = urow; out->dev.rowptr = rptr; out->dev.csr2csc = c2c; out->dev.csrcol = ccol;
  out->dev.cand_pos = cpos;
  out->base_val = uval;
  out->z1_dev = z1;
  out->z2_dev = z2;
  SCL_TRY(pattern_add_csr_companions(ctx, out));
  return SCLENS_OK;
}

}  // namespace scl
