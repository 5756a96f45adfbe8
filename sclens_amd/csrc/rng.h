// Counter-based randomness shared by host (rng.cpp) and device (scale.hip): a keyed bijection of [0, len) built from a
// balanced Feistel network with cycle walking. Evaluating it at t = 0..m-1 yields m distinct, uniformly scattered
// indices -- a sample without replacement (scLENS.jl:731, :772: `sample(1:len, m, replace=false)`) that needs no
// memory, no host round trip, and is identical on host and device.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

#if defined(__HIPCC__)
#define SCL_HD __host__ __device__
#else
#define SCL_HD
#endif

namespace scl {

SCL_HD inline uint32_t mix32(uint32_t x) {  // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
SCL_HD inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// R1 (scLENS.jl:669): draw t of the zero-candidate list as a pure function of (seed, t): two splitmix64 outputs mapped to
// [0, N) x [0, M) by the high half of a 64 x 64 bit product. Shared by the host generator (rng.cpp) and the device one
// (pattern_dev.hip): both produce the same list.
SCL_HD inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((__uint128_t)a * b) >> 64);
#endif
}
SCL_HD inline void r1_draw(uint64_t seed, uint64_t t, uint64_t N, uint64_t M, uint64_t* i, uint64_t* j) {
  uint64_t s = seed + 0x632BE59BD9B4E019ull * (t + 1);
  const uint64_t a = splitmix64(s), b = splitmix64(s);
  *i = mulhi64(a, N);
  *j = mulhi64(b, M);
}

struct FeistelPerm {
  uint64_t len;
  uint32_t half_bits, half_mask;
  uint32_t key[6];
};

inline FeistelPerm feistel_make(uint64_t len, uint64_t seed) {
  FeistelPerm p;
  p.len = len;
  uint32_t bits = 2;
  while ((1ull << bits) < len) ++bits;
  if (bits & 1u) ++bits;
  p.half_bits = bits / 2;
  p.half_mask = (1u << p.half_bits) - 1u;
  uint64_t s = seed ^ 0xD1B54A32D192ED03ull;
  for (int i = 0; i < 6; ++i) p.key[i] = (uint32_t)(splitmix64(s) >> 16);
  return p;
}

SCL_HD inline uint64_t feistel_apply(const FeistelPerm& p, uint64_t t) {
  uint64_t x = t;
  do {  // cycle walking: the domain 2^(2*half_bits) is < 4*len, so <= 4 expected iterations
    uint32_t L = (uint32_t)(x >> p.half_bits) & p.half_mask, R = (uint32_t)x & p.half_mask;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const uint32_t F = mix32(R * 0x9E3779B1u + p.key[i]) & p.half_mask;
      const uint32_t nl = R;
      R = L ^ F;
      L = nl;
    }
    x = ((uint64_t)L << p.half_bits) | R;
  } while (x >= p.len);
  return x;
}

// Host threads a parallel host pass may use: the CPUs this process can actually run on -- the cgroup quota (containers:
// 256 visible CPUs with a quota of 16 is what the GPU box hands out), capped at 64; SCLENS_HIP_HOST_THREADS overrides.
inline int host_parallelism() {
  if (const char* e = getenv("SCLENS_HIP_HOST_THREADS")) return std::max(1, std::min(64, atoi(e)));
  int T = (int)std::thread::hardware_concurrency();
  if (T <= 0) T = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
    long long q = 0, per = 0;
    if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) T = std::min<int>(T, (int)((q + per - 1) / per));
    fclose(f);
  } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    long long q = -1, per = 100000;
    if (fscanf(f1, "%lld", &q) != 1) q = -1;
    fclose(f1);
    if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(f2, "%lld", &per) != 1) per = 100000;
      fclose(f2);
    }
    if (q > 0 && per > 0) T = std::min<int>(T, (int)((q + per - 1) / per));
  }
  return std::max(1, std::min(T, 64));
}

}  // namespace scl
