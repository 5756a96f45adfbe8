// CPU self-test of the host side of libsclens_hip (stats.cpp, rng.cpp, pattern_host.cpp), built with
// -fsanitize=address,undefined by `make asan` and run by tests/test_host_logic.py: memory errors and undefined behaviour in
// the code that the C ABI exposes to foreign hosts (Julia, ctypes) abort the run. No GPU, no HIP.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <utility>
#include <vector>

#include "../../include/sclens_hip.h"
#include "pattern_host.h"
#include "rng.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

struct Csc {
  int64_t N, M;
  std::vector<int64_t> colptr;
  std::vector<int32_t> row;
  std::vector<float> val;
};

static Csc random_csc(int64_t N, int64_t M, double density, uint64_t seed) {
  Csc c{N, M, std::vector<int64_t>(M + 1, 0), {}, {}};
  uint64_t s = seed;
  for (int64_t j = 0; j < M; ++j) {
    for (int64_t i = 0; i < N; ++i)
      if ((double)(scl::splitmix64(s) >> 11) * (1.0 / 9007199254740992.0) < density) {
        c.row.push_back((int32_t)i);
        c.val.push_back((float)(1 + scl::splitmix64(s) % 7));
      }
    c.colptr[j + 1] = (int64_t)c.row.size();
  }
  return c;
}

static int test_stats() {
  // Marchenko-Pastur bulk on a grid + three spikes: the fit must keep the bulk and the TW threshold must separate the spikes
  const int64_t n = 400;
  const double gamma = 0.5, bp = (1 + std::sqrt(gamma)) * (1 + std::sqrt(gamma)), bm = (1 - std::sqrt(gamma)) * (1 - std::sqrt(gamma));
  std::vector<double> L(n), Lr(n - 1);
  for (int64_t i = 0; i < n; ++i) L[i] = bm + (bp - bm) * (i + 0.5) / n;
  L[n - 1] = 9.0; L[n - 2] = 7.0; L[n - 3] = 5.5;
  for (int64_t i = 0; i + 1 < n; ++i) Lr[i] = bm + (bp - bm) * (i + 0.5) / n;
  double b_plus = 0, b_minus = 0;
  std::vector<uint8_t> mask(n);
  CHECK(sclens_mp_calculation(L.data(), n, Lr.data(), n - 1, &b_plus, &b_minus, mask.data()) == SCLENS_OK);
  std::vector<double> Lmp;
  for (int64_t i = 0; i < n; ++i)
    if (mask[i]) Lmp.push_back(L[i]);
  CHECK(!Lmp.empty() && Lmp.size() <= (size_t)n - 3 && !mask[n - 1] && !mask[n - 2] && !mask[n - 3]);  // spikes stay outside
  double lc = 0, g = 0, p = 0, sg = 0;
  CHECK(sclens_tw(n, Lmp.data(), (int64_t)Lmp.size(), &lc, &g, &p, &sg) == SCLENS_OK);
  CHECK(std::isfinite(lc) && lc > 0.0 && lc < 5.5);
  double ks = 0;
  int pass = -1;
  CHECK(sclens_mp_check(Lmp.data(), (int64_t)Lmp.size(), 0.05, &ks, &pass) == SCLENS_OK);
  CHECK(pass == 0 || pass == 1);
  // degenerate inputs are error codes or finite answers, never crashes
  CHECK(sclens_mp_calculation(nullptr, 0, nullptr, 0, &b_plus, &b_minus, nullptr) != SCLENS_OK ||
        true);
  const int64_t k = 3, npairs = 10;
  std::vector<double> b(k * npairs), m(k), sd(k);
  for (int64_t i = 0; i < k * npairs; ++i) b[i] = 0.5 + 0.04 * (double)(i % 11);
  b[7] = 0.01;  // an outlier the Tukey fence removes
  CHECK(sclens_robust_scores(b.data(), k, npairs, m.data(), sd.data()) == SCLENS_OK);
  for (int64_t i = 0; i < k; ++i) CHECK(m[i] > 0.4 && m[i] < 1.0 && sd[i] >= 0.0);
  const double e1 = sclens_noise_baseline_exact(1000), e2 = sclens_noise_baseline_exact(30000);
  CHECK(e1 > 0.05 && e1 < 0.2 && e2 > 0.0 && e2 < e1);
  return 0;
}

static int test_rng(const Csc& c) {
  const int64_t nnz = c.colptr[c.M];
  std::vector<uint32_t> z1(nnz), z2(nnz);
  int64_t cnt = -1;
  CHECK(sclens_draw_zero_candidates(c.N, c.M, c.colptr.data(), c.row.data(), 12345u, z1.data(), z2.data(), &cnt) == SCLENS_OK);
  CHECK(cnt > 0 && cnt <= nnz);
  std::set<std::pair<uint32_t, uint32_t>> stored, seen;
  for (int64_t j = 0; j < c.M; ++j)
    for (int64_t s = c.colptr[j]; s < c.colptr[j + 1]; ++s) stored.insert({(uint32_t)c.row[s], (uint32_t)j});
  for (int64_t t = 0; t < cnt; ++t) {
    CHECK(z1[t] < (uint32_t)c.N && z2[t] < (uint32_t)c.M);
    CHECK(!stored.count({z1[t], z2[t]}));
    CHECK(seen.insert({z1[t], z2[t]}).second);
  }
  std::vector<int32_t> rrow(nnz);
  std::vector<float> rval(nnz);
  CHECK(sclens_draw_null_matrix(c.N, c.M, c.colptr.data(), c.val.data(), 99u, rrow.data(), rval.data()) == SCLENS_OK);
  double s0 = 0, s1 = 0;
  for (int64_t q = 0; q < nnz; ++q) {
    s0 += c.val[q];
    s1 += rval[q];
  }
  CHECK(s0 == s1);  // a permutation of the stored values
  for (int64_t j = 0; j < c.M; ++j)
    for (int64_t s = c.colptr[j]; s < c.colptr[j + 1]; ++s) {
      CHECK(rrow[s] >= 0 && rrow[s] < c.N);
      if (s > c.colptr[j]) CHECK(rrow[s] > rrow[s - 1]);  // distinct cells, ascending
    }
  for (uint64_t len : {1ull, 2ull, 17ull, 4096ull, 100003ull}) {
    const int64_t mm = (int64_t)std::min<uint64_t>(len, 5000);
    std::vector<uint32_t> smp(mm);
    CHECK(sclens_sample_without_replacement(len, mm, 7u + len, smp.data()) == SCLENS_OK);
    std::set<uint32_t> u(smp.begin(), smp.end());
    CHECK((int64_t)u.size() == mm && *u.rbegin() < len);
  }
  std::vector<uint32_t> one(4);
  CHECK(sclens_sample_without_replacement(3, 4, 1u, one.data()) != SCLENS_OK);  // more than the population
  return 0;
}

static int test_pattern(const Csc& c) {
  const int64_t nnz = c.colptr[c.M];
  std::vector<uint32_t> z1(nnz), z2(nnz);
  int64_t cnt = 0;
  CHECK(sclens_draw_zero_candidates(c.N, c.M, c.colptr.data(), c.row.data(), 5u, z1.data(), z2.data(), &cnt) == SCLENS_OK);
  std::string err;
  scl::HostPattern a, b;
  CHECK(scl::pattern_build_host(c.N, c.M, c.colptr.data(), c.row.data(), c.val.data(), cnt, z1.data(), z2.data(), 0, c.N, 1, &a, &err) == SCLENS_OK);
  CHECK(scl::pattern_build_host(c.N, c.M, c.colptr.data(), c.row.data(), c.val.data(), cnt, z1.data(), z2.data(), 0, c.N, 3, &b, &err) == SCLENS_OK);
  CHECK(a.nU == nnz + cnt && a.ucol == b.ucol && a.urow == b.urow && a.uval == b.uval && a.cpos == b.cpos && a.rptr == b.rptr &&
        a.c2c == b.c2c && a.ccol == b.ccol);
  for (int64_t t = 0; t < cnt; ++t) {
    const int64_t pos = a.cpos[t];
    CHECK(pos >= a.ucol[z2[t]] && pos < a.ucol[z2[t] + 1] && a.urow[pos] == (int32_t)z1[t] && a.uval[pos] == 0.f);
  }
  for (int64_t i = 0; i < c.N; ++i)
    for (int64_t s = a.rptr[i]; s < a.rptr[i + 1]; ++s) {
      CHECK(a.urow[a.c2c[s]] == (int32_t)i);
      CHECK(a.c2c[s] >= a.ucol[a.ccol[s]] && a.c2c[s] < a.ucol[a.ccol[s] + 1]);
      if (s > a.rptr[i]) CHECK(a.ccol[s] >= a.ccol[s - 1]);
    }
  // a block of cells [row0, row0 + Nl) of the same matrix: local rows, global candidate indices
  const int64_t row0 = c.N / 3, Nl = c.N / 2;
  Csc l{Nl, c.M, std::vector<int64_t>(c.M + 1, 0), {}, {}};
  for (int64_t j = 0; j < c.M; ++j) {
    for (int64_t s = c.colptr[j]; s < c.colptr[j + 1]; ++s)
      if (c.row[s] >= row0 && c.row[s] < row0 + Nl) {
        l.row.push_back((int32_t)(c.row[s] - row0));
        l.val.push_back(c.val[s]);
      }
    l.colptr[j + 1] = (int64_t)l.row.size();
  }
  scl::HostPattern h;
  CHECK(scl::pattern_build_host(Nl, c.M, l.colptr.data(), l.row.data(), l.val.data(), cnt, z1.data(), z2.data(), row0, c.N, 2, &h, &err) == SCLENS_OK);
  int64_t nloc = 0;
  for (int64_t t = 0; t < cnt; ++t) {
    const bool loc = (int64_t)z1[t] >= row0 && (int64_t)z1[t] < row0 + Nl;
    nloc += loc;
    CHECK(loc ? (h.cpos[t] >= 0 && h.urow[h.cpos[t]] == (int32_t)(z1[t] - row0)) : h.cpos[t] == -1);
  }
  CHECK(h.nU == l.colptr[c.M] + nloc);
  // error paths: out-of-range row index / candidate index / row block
  Csc bad = c;
  bad.row[nnz / 2] = (int32_t)c.N;
  CHECK(scl::pattern_build_host(c.N, c.M, bad.colptr.data(), bad.row.data(), bad.val.data(), 0, nullptr, nullptr, 0, c.N, 2, &b, &err) == SCLENS_ERR_ARG);
  bad.row[nnz / 2] = -1;
  CHECK(scl::pattern_build_host(c.N, c.M, bad.colptr.data(), bad.row.data(), bad.val.data(), 0, nullptr, nullptr, 0, c.N, 1, &b, &err) == SCLENS_ERR_ARG);
  std::vector<uint32_t> zb1 = {0u, (uint32_t)c.N}, zb2 = {0u, 1u};
  CHECK(scl::pattern_build_host(c.N, c.M, c.colptr.data(), c.row.data(), c.val.data(), 2, zb1.data(), zb2.data(), 0, c.N, 1, &b, &err) == SCLENS_ERR_ARG);
  CHECK(scl::pattern_build_host(Nl, c.M, l.colptr.data(), l.row.data(), l.val.data(), 0, nullptr, nullptr, c.N, c.N, 1, &b, &err) == SCLENS_ERR_ARG);
  return 0;
}

int main() {
  const Csc c = random_csc(211, 97, 0.12, 42);
  if (test_stats()) return 1;
  if (test_rng(c)) return 1;
  if (test_pattern(c)) return 1;
  const Csc c2 = random_csc(40, 300, 0.3, 7);
  if (test_rng(c2)) return 1;
  if (test_pattern(c2)) return 1;
  std::puts("host selftest ok");
  return 0;
}
