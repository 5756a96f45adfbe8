// Host-side random draws of sclens(): R1 zero candidates (scLENS.jl:668-673), R2 null matrix
// (scLENS.jl:701 -> :261-289, :239-248; intent as documented in DESIGN.md), R4/R5 index samples (same Feistel
// permutation the device uses). Plain C++ (no HIP). xoshiro256** seeded by splitmix64.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
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

// R1 (scLENS.jl:668-673): nnz uniform (i, j) draws with replacement, minus the stored entries, duplicates removed, in
// first-occurrence order. Draw t is a pure function of (seed, t) (counter-based: two splitmix64 outputs mapped to [0, N) and
// [0, M) by a 128-bit multiply), so any number of host threads produces the same list: thread b owns a range of genes, walks
// ALL draws, keeps those of its genes that its private bitmap (stored entries + earlier draws) has not seen, and the per-thread
// survivor lists (each ascending in t) are merged back into draw order.
using scl::r1_draw;

static int host_threads(int64_t work_items) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(scl::host_parallelism(), work_items / 500000 + 1));
}

int sclens_draw_zero_candidates(int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, uint64_t seed,
                                uint32_t* z1, uint32_t* z2, int64_t* count) {
  if (N <= 0 || M <= 0 || !colptr || !z1 || !z2 || !count) return SCLENS_ERR_ARG;
  const int64_t nnz = colptr[M];
  const int T = (int)std::min<int64_t>(host_threads(nnz), M);
  struct Hit { uint32_t t, i, j; };
  std::vector<std::vector<Hit>> hits(T);
  auto owner = [&](int b) {
    const int64_t j0 = M * b / T, j1 = M * (b + 1) / T;
    std::vector<uint64_t> taken(((uint64_t)(j1 - j0) * (uint64_t)N + 63) / 64, 0);  // stored entries + already drawn pairs
    for (int64_t j = j0; j < j1; ++j)
      for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
        const uint64_t key = (uint64_t)rowval[q] + (uint64_t)(j - j0) * (uint64_t)N;
        taken[key >> 6] |= 1ull << (key & 63);
      }
    std::vector<Hit>& out = hits[b];
    out.reserve((size_t)(nnz / T + nnz / (8 * T) + 16));
    for (int64_t t = 0; t < nnz; ++t) {
      uint64_t i, j;
      r1_draw(seed, (uint64_t)t, (uint64_t)N, (uint64_t)M, &i, &j);
      if ((int64_t)j < j0 || (int64_t)j >= j1) continue;
      const uint64_t key = i + (j - (uint64_t)j0) * (uint64_t)N;
      uint64_t& w = taken[key >> 6];
      const uint64_t bit = 1ull << (key & 63);
      if (w & bit) continue;  // stored entry (setdiff, :671) or duplicate draw (first occurrence kept)
      w |= bit;
      out.push_back(Hit{(uint32_t)t, (uint32_t)i, (uint32_t)j});
    }
  };
  {
    std::vector<std::thread> th;
    for (int b = 1; b < T; ++b) th.emplace_back(owner, b);
    owner(0);
    for (auto& x : th) x.join();
  }
  // merge by draw index: thread c writes the survivors with t in [nnz c / T, nnz (c + 1) / T)
  std::vector<int64_t> offs(T + 1, 0);
  std::vector<std::vector<size_t>> lo(T, std::vector<size_t>(T + 1));
  for (int b = 0; b < T; ++b) {
    for (int c = 0; c <= T; ++c) {
      const uint32_t tb = (uint32_t)std::min<int64_t>(nnz * c / T, 0xFFFFFFFFll);
      lo[b][c] = (c == T) ? hits[b].size()
                          : (size_t)(std::lower_bound(hits[b].begin(), hits[b].end(), tb, [](const Hit& h, uint32_t v) { return h.t < v; }) -
                                     hits[b].begin());
    }
  }
  for (int c = 0; c < T; ++c) {
    int64_t n_c = 0;
    for (int b = 0; b < T; ++b) n_c += (int64_t)(lo[b][c + 1] - lo[b][c]);
    offs[c + 1] = offs[c] + n_c;
  }
  auto merger = [&](int c) {
    std::vector<size_t> cur(T);
    for (int b = 0; b < T; ++b) cur[b] = lo[b][c];
    int64_t o = offs[c];
    while (o < offs[c + 1]) {  // T-way merge by t (T <= 64: linear scan of the heads)
      int best = -1;
      uint32_t bt = 0xFFFFFFFFu;
      for (int b = 0; b < T; ++b)
        if (cur[b] < lo[b][c + 1] && (best < 0 || hits[b][cur[b]].t < bt)) {
          best = b;
          bt = hits[b][cur[b]].t;
        }
      const Hit& h = hits[best][cur[best]++];
      z1[o] = h.i;
      z2[o] = h.j;
      ++o;
    }
  };
  {
    std::vector<std::thread> th;
    for (int c = 1; c < T; ++c) th.emplace_back(merger, c);
    merger(0);
    for (auto& x : th) x.join();
  }
  *count = offs[T];
  return SCLENS_OK;
}

int sclens_draw_null_matrix(int64_t N, int64_t M, const int64_t* colptr, const float* nzval, uint64_t seed,
                            int32_t* out_rowval, float* out_nzval) {
  if (N <= 0 || M <= 0 || !colptr || !nzval || !out_rowval || !out_nzval) return SCLENS_ERR_ARG;
  const int64_t nnz = colptr[M];
  for (int64_t j = 0; j < M; ++j)
    if (colptr[j + 1] - colptr[j] > N || colptr[j + 1] < colptr[j]) return SCLENS_ERR_ARG;
  const int T = host_threads(nnz);
  // (1) the stored values in random order (scLENS.jl:275), as a two-level shuffle: every value goes to one of NBK buckets
  //     chosen uniformly -- round 4: by a COUNTER-based draw, bucket(q) = top byte of splitmix64(key + q), so that the pass runs on
  //     all threads (chunks of q with per-chunk histograms, a prefix sum, a stable scatter: the result is the one the sequential
  //     loop over q gives and does not depend on the thread count; the sequential generator of rounds 1-3 made this pass 0.4 of the
  //     0.6 s the null matrix takes at 100 000 x 30 000, which the null decomposition of sclens() waits for) --, then each bucket --
  //     small enough for the cache -- gets its own Fisher-Yates shuffle from its own generator (in parallel). Concatenated, that is
  //     a uniform random permutation that depends on the seed only.
  // (2) per gene the same number of entries at uniformly drawn distinct cells (:247): gene j has its own generator
  //     seeded from (seed, j)
  constexpr int NBK = 256;
  std::vector<int64_t> bstart(NBK + 1, 0);
  {
    auto mix = [](uint64_t x) { return scl::splitmix64(x); };  // (by value: the generator form advances its argument)
    const uint64_t key = mix(seed ^ 0xA5A5A5A5DEADBEEFull);
    std::vector<uint8_t> bucket((size_t)nnz);
    std::vector<std::vector<int64_t>> hist(T, std::vector<int64_t>(NBK, 0));
    auto count = [&](int t) {
      std::vector<int64_t>& h = hist[t];
      for (int64_t q = nnz * t / T; q < nnz * (t + 1) / T; ++q) {
        const uint8_t b = (uint8_t)(mix(key + 0x9E3779B97F4A7C15ull * (uint64_t)q) >> 56);
        bucket[q] = b;
        h[b] += 1;
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(count, t);
      count(0);
      for (auto& x : th) x.join();
    }
    for (int b = 0; b < NBK; ++b) {
      int64_t c = 0;
      for (int t = 0; t < T; ++t) c += hist[t][b];
      bstart[b + 1] = bstart[b] + c;
    }
    // start of chunk t inside bucket b: the bucket's start + the counts of the earlier chunks (order of q preserved)
    for (int b = 0; b < NBK; ++b) {
      int64_t o = bstart[b];
      for (int t = 0; t < T; ++t) {
        const int64_t c = hist[t][b];
        hist[t][b] = o;
        o += c;
      }
    }
    auto scatter = [&](int t) {
      std::vector<int64_t>& cur = hist[t];
      for (int64_t q = nnz * t / T; q < nnz * (t + 1) / T; ++q) out_nzval[cur[bucket[q]]++] = nzval[q];
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(scatter, t);
      scatter(0);
      for (auto& x : th) x.join();
    }
  }
  auto work = [&](int t) {
    for (int b = NBK * t / T; b < NBK * (t + 1) / T; ++b) {
      uint64_t sb = seed ^ (0xC2B2AE3D27D4EB4Full * (uint64_t)(b + 1));
      Xo rb(scl::splitmix64(sb));
      float* v = out_nzval + bstart[b];
      for (int64_t i = bstart[b + 1] - bstart[b] - 1; i > 0; --i) std::swap(v[i], v[rb.below((uint64_t)i + 1)]);
    }
    std::vector<uint64_t> mark((N + 63) / 64);
    const int64_t j0 = M * t / T, j1 = M * (t + 1) / T;
    for (int64_t j = j0; j < j1; ++j) {
      const int64_t b = colptr[j], c = colptr[j + 1] - b;
      if (c <= 0) continue;
      uint64_t sj = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1);
      Xo rng(scl::splitmix64(sj));
      std::fill(mark.begin(), mark.end(), 0);
      const bool complement = 2 * c > N;  // dense gene: draw the cells that are NOT expressed
      const int64_t want = complement ? N - c : c;
      int64_t got = 0;
      while (got < want) {
        const uint64_t r = rng.below((uint64_t)N);
        if (mark[r >> 6] & (1ull << (r & 63))) continue;
        mark[r >> 6] |= 1ull << (r & 63);
        ++got;
      }
      int64_t q = b;
      for (size_t wi = 0; wi < mark.size(); ++wi) {  // ascending cells, word by word
        uint64_t w = complement ? ~mark[wi] : mark[wi];
        if (wi == mark.size() - 1 && (N & 63)) w &= (1ull << (N & 63)) - 1;
        while (w) {
          out_rowval[q++] = (int32_t)(wi * 64 + (size_t)__builtin_ctzll(w));
          w &= w - 1;
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return SCLENS_OK;
}

int sclens_sample_without_replacement(uint64_t len, int64_t m, uint64_t seed, uint32_t* out) {
  if (!out || m < 0 || (uint64_t)m > len || len == 0 || len > 0xFFFFFFFFull) return SCLENS_ERR_ARG;
  const scl::FeistelPerm p = scl::feistel_make(len, seed);
  for (int64_t t = 0; t < m; ++t) out[t] = (uint32_t)scl::feistel_apply(p, (uint64_t)t);
  return SCLENS_OK;
}

}  // extern "C"
