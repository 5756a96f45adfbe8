// Host builder of the sparse pattern (see pattern_host.h); moved out of session.hip so that it has no HIP dependency.
#include "pattern_host.h"

#include <algorithm>
#include <thread>

#include "../../include/sclens_hip.h"
#include "rng.h"

namespace scl {

int pattern_build_host(int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t ncand,
                       const uint32_t* z1, const uint32_t* z2, int64_t row0, int64_t N_global, int threads, HostPattern* out,
                       std::string* err) {
  if (N <= 0 || M <= 0 || !colptr || (colptr[M] > 0 && (!rowval || !nzval)) || (ncand > 0 && (!z1 || !z2)) || !out || !err) {
    if (err) *err = "pattern_build_host: bad arguments";
    return SCLENS_ERR_ARG;
  }
  if (N_global <= 0) N_global = N;
  if (row0 < 0 || row0 + N > N_global) {
    *err = "pattern_build_host: bad row range";
    return SCLENS_ERR_ARG;
  }
  const int64_t nnz = colptr[M];
  // candidates whose cell lies in [row0, row0 + N) are local; the others keep their list index with slot -1
  auto local = [&](int64_t t) { return (int64_t)z1[t] >= row0 && (int64_t)z1[t] < row0 + N; };
  std::vector<int64_t>& ucol = out->ucol;
  ucol.assign(M + 1, 0);
  for (int64_t j = 0; j < M; ++j) ucol[j + 1] = colptr[j + 1] - colptr[j];
  int64_t ncl = 0;
  for (int64_t t = 0; t < ncand; ++t) {
    if (z2[t] >= (uint64_t)M || z1[t] >= (uint64_t)N_global) { *err = "candidate index out of range"; return SCLENS_ERR_ARG; }
    if (local(t)) {
      ucol[z2[t] + 1] += 1;
      ncl += 1;
    }
  }
  const int64_t nU = nnz + ncl;
  for (int64_t j = 0; j < M; ++j) ucol[j + 1] += ucol[j];
  out->nU = nU;
  std::vector<int32_t>& urow = out->urow;
  std::vector<float>& uval = out->uval;
  std::vector<int64_t>& cpos = out->cpos;
  urow.assign(nU, 0);
  uval.assign(nU, 0.f);
  cpos.assign(ncand, 0);
  // host threads: every pass below is split by a range of genes or of cells whose owner scans the whole input and handles
  // its own part, so the result does not depend on the thread count
  const int T = threads > 0 ? threads : (int)std::max<int64_t>(1, std::min<int64_t>(host_parallelism(), nU / 1000000 + 1));
  std::vector<int> bad(T, 0);
  auto run = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto& x : th) x.join();
  };
  // CSC fill: stored counts and the candidate cursors of the owner's genes (candidate order inside a gene = list order)
  run([&](int t) {
    const int64_t j0 = M * t / T, j1 = M * (t + 1) / T;
    for (int64_t j = j0; j < j1; ++j) {
      int64_t q = ucol[j];
      for (int64_t s = colptr[j]; s < colptr[j + 1]; ++s, ++q) {
        if (rowval[s] < 0 || rowval[s] >= N) { bad[t] = 1; return; }
        urow[q] = rowval[s];
        uval[q] = nzval[s];
      }
    }
    std::vector<int64_t> cur(j1 - j0);
    for (int64_t j = j0; j < j1; ++j) cur[j - j0] = ucol[j] + (colptr[j + 1] - colptr[j]);
    for (int64_t c = 0; c < ncand; ++c) {
      const int64_t j = (int64_t)z2[c];
      if (j < j0 || j >= j1) continue;
      if (local(c)) {
        const int64_t pos = cur[j - j0]++;
        cpos[c] = pos;
        urow[pos] = (int32_t)((int64_t)z1[c] - row0);
      } else {
        cpos[c] = -1;
      }
    }
  });
  for (int b : bad)
    if (b) { *err = "row index out of range"; return SCLENS_ERR_ARG; }
  // CSR view: thread t owns the row range [r0, r1): it scans every slot but only counts / places its own rows, so the
  // order inside a row (ascending column, slot order inside a column) does not depend on the thread count
  std::vector<int64_t>& rptr = out->rptr;
  rptr.assign(N + 1, 0);
  run([&](int t) {
    const int32_t r0 = (int32_t)(N * t / T), r1 = (int32_t)(N * (t + 1) / T);
    for (int64_t q = 0; q < nU; ++q) {
      const int32_t r = urow[q];
      if (r >= r0 && r < r1) rptr[r + 1] += 1;
    }
  });
  for (int64_t i = 0; i < N; ++i) rptr[i + 1] += rptr[i];
  std::vector<int64_t>& c2c = out->c2c;
  std::vector<int32_t>& ccol = out->ccol;
  c2c.assign(nU, 0);
  ccol.assign(nU, 0);
  {
    std::vector<int64_t> rc(rptr.begin(), rptr.end() - 1);
    run([&](int t) {
      const int32_t r0 = (int32_t)(N * t / T), r1 = (int32_t)(N * (t + 1) / T);
      for (int64_t j = 0; j < M; ++j)
        for (int64_t q = ucol[j]; q < ucol[j + 1]; ++q) {
          const int32_t r = urow[q];
          if (r < r0 || r >= r1) continue;
          const int64_t s = rc[r]++;
          c2c[s] = q;
          ccol[s] = (int32_t)j;
        }
    });
  }
  return SCLENS_OK;
}

}  // namespace scl
