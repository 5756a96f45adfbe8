// Host-side random draws of sclens(): R1 zero candidates (scLENS.jl:668-673), R2 null matrix
// (scLENS.jl:701 -> :261-289, :239-248; intent as documented in DESIGN.md), R4/R5 index samples (same Feistel
// permutation the device uses). Plain C++ (no HIP). xoshiro256** seeded by splitmix64.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/sclens_hip.h"
#include "rng.h"

namespace {
struct Xo {
  uint64_t s[4];
  explicit Xo(uint64_t seed) { for (auto& v : s) v = scl::splitmix64(seed); }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  uint64_t below(uint64_t n) {  // unbiased (Lemire)
    uint64_t x = next();
    __uint128_t m = (__uint128_t)x * n;
    uint64_t l = (uint64_t)m;
    if (l < n) {
      const uint64_t t = (0 - n) % n;
      while (l < t) { x = next(); m = (__uint128_t)x * n; l = (uint64_t)m; }
    }
    return (uint64_t)(m >> 64);
  }
};
}  // namespace

extern "C" {

int sclens_draw_zero_candidates(int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, uint64_t seed,
                                uint32_t* z1, uint32_t* z2, int64_t* count) {
  if (N <= 0 || M <= 0 || !colptr || !z1 || !z2 || !count) return SCLENS_ERR_ARG;
  const int64_t nnz = colptr[M];
  const uint64_t cells = (uint64_t)N * (uint64_t)M;
  std::vector<uint64_t> taken((cells + 63) / 64, 0);  // stored entries + already drawn pairs
  for (int64_t j = 0; j < M; ++j)
    for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
      const uint64_t key = (uint64_t)rowval[q] + (uint64_t)j * (uint64_t)N;
      taken[key >> 6] |= 1ull << (key & 63);
    }
  Xo rng(seed);
  int64_t c = 0;
  for (int64_t t = 0; t < nnz; ++t) {  // nnz uniform (i, j) draws with replacement (scLENS.jl:669)
    const uint64_t i = rng.below((uint64_t)N), j = rng.below((uint64_t)M);
    const uint64_t key = i + j * (uint64_t)N;
    uint64_t& w = taken[key >> 6];
    const uint64_t bit = 1ull << (key & 63);
    if (w & bit) continue;  // stored entry (setdiff, :671) or duplicate draw (first occurrence kept)
    w |= bit;
    z1[c] = (uint32_t)i;
    z2[c] = (uint32_t)j;
    ++c;
  }
  *count = c;
  return SCLENS_OK;
}

int sclens_draw_null_matrix(int64_t N, int64_t M, const int64_t* colptr, const float* nzval, uint64_t seed,
                            int32_t* out_rowval, float* out_nzval) {
  if (N <= 0 || M <= 0 || !colptr || !nzval || !out_rowval || !out_nzval) return SCLENS_ERR_ARG;
  const int64_t nnz = colptr[M];
  Xo rng(seed ^ 0xA5A5A5A5DEADBEEFull);
  std::copy(nzval, nzval + nnz, out_nzval);
  for (int64_t i = nnz - 1; i > 0; --i) std::swap(out_nzval[i], out_nzval[rng.below((uint64_t)i + 1)]);  // shuffle (:275)
  std::vector<uint64_t> mark((N + 63) / 64);
  for (int64_t j = 0; j < M; ++j) {  // per gene: the same number of entries at uniformly drawn distinct cells (:247)
    const int64_t b = colptr[j], c = colptr[j + 1] - b;
    if (c <= 0) continue;
    if (c > N) return SCLENS_ERR_ARG;
    std::fill(mark.begin(), mark.end(), 0);
    if (2 * c <= N) {
      int64_t got = 0;
      while (got < c) {
        const uint64_t r = rng.below((uint64_t)N);
        if (mark[r >> 6] & (1ull << (r & 63))) continue;
        mark[r >> 6] |= 1ull << (r & 63);
        ++got;
      }
    } else {  // dense column: draw the complement
      int64_t got = 0;
      while (got < N - c) {
        const uint64_t r = rng.below((uint64_t)N);
        if (mark[r >> 6] & (1ull << (r & 63))) continue;
        mark[r >> 6] |= 1ull << (r & 63);
        ++got;
      }
      for (auto& w : mark) w = ~w;
    }
    int64_t q = b;
    for (int64_t r = 0; r < N && q < b + c; ++r)
      if (mark[r >> 6] & (1ull << (r & 63))) out_rowval[q++] = (int32_t)r;  // ascending rows
  }
  return SCLENS_OK;
}

int sclens_sample_without_replacement(uint64_t len, int64_t m, uint64_t seed, uint32_t* out) {
  if (!out || m < 0 || (uint64_t)m > len || len == 0 || len > 0xFFFFFFFFull) return SCLENS_ERR_ARG;
  const scl::FeistelPerm p = scl::feistel_make(len, seed);
  for (int64_t t = 0; t < m; ++t) out[t] = (uint32_t)scl::feistel_apply(p, (uint64_t)t);
  return SCLENS_OK;
}

}  // extern "C"
