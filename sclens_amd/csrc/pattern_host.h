// Host arrays of the sparse pattern (counts UNION zero candidates, scLENS.jl:664-673) for sessions that hold a block of cells:
// pure C++ (no HIP), so the builder is also compiled and run under AddressSanitizer / UBSan on the CPU (`make asan`).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace scl {

struct HostPattern {
  int64_t nU = 0;
  std::vector<int64_t> ucol;   // [M+1]  CSC: per gene the stored counts (rows ascending), then the local candidates (list order)
  std::vector<int32_t> urow;   // [nU]   local cell index
  std::vector<float> uval;     // [nU]   stored counts, 0 in candidate slots
  std::vector<int64_t> cpos;   // [ncand] CSC slot of candidate t, -1 when its cell is not in [row0, row0 + N)
  std::vector<int64_t> rptr;   // [N+1]  CSR view
  std::vector<int64_t> c2c;    // [nU]   CSC slot of each CSR slot
  std::vector<int32_t> ccol;   // [nU]
};

// returns 0 or SCLENS_ERR_ARG (message in *err). z1 holds GLOBAL cell indices of which [row0, row0 + N) are local.
// `threads` <= 0: host_parallelism(). The result does not depend on the thread count.
int pattern_build_host(int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t ncand,
                       const uint32_t* z1, const uint32_t* z2, int64_t row0, int64_t N_global, int threads, HostPattern* out,
                       std::string* err);

}  // namespace scl
