// Device memory pool of libsclens_hip.so. Every sclens() call creates and destroys sessions, worker sessions and worker contexts
// (api.sclens / the Julia shim: one session per call, `streams` - 1 workers), each with tens of GB of scratch at the sizes of
// BASELINE.json; hipMalloc / hipFree of such blocks cost ~2 s per call at 100 000 x 30 000 (the teardown of one worker session
// alone 2.2 s, DESIGN.md section 7 of round 2). Freed blocks therefore go to a per-device free list and are handed out again to
// requests of the same size class (the next call asks for exactly the same sizes); nothing is returned to the driver until
// sclens_hip_trim() / the cap is exceeded / an allocation fails. SCLENS_HIP_POOL=0 switches the pool off (plain hipMalloc / hipFree).
//
// Ordering: a block is only put on the free list by pool_free(p, stream) AFTER `stream` (the stream of the context that used it)
// has been synchronised, so a later owner on another stream never races with pending work on the block.
#include <map>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace scl {

namespace {
struct DevPool {
  std::multimap<size_t, void*> free_blocks;  // size -> block
  size_t cached = 0, live = 0, hits = 0, misses = 0, peak = 0;  // peak: largest `live` since the last reset
};
std::mutex g_mu;
std::map<int, DevPool> g_pool;                              // by device
std::unordered_map<void*, std::pair<int, size_t>> g_live;   // block -> (device, size)
bool pool_enabled() {
  static const bool on = !(getenv("SCLENS_HIP_POOL") && atoi(getenv("SCLENS_HIP_POOL")) == 0);
  return on;
}
// Cached (idle) bytes per device above which freed blocks go straight back to the driver. Default: what was FREE on the device when the
// pool first looked (other processes and other allocators of this process keep what they hold), less an eighth of the device (at least
// 24 GB) for everybody else -- 252 GB on an otherwise empty MI355X (what the library itself holds at that moment counts as available). sclens_hip_pool_set_cap() overrides it (hosts that put several
// ranks on one device give each its share; SCLENS_HIP_POOL_MAX_GB is the same knob for unmodified hosts, read once). Round 4 started
// with HALF of the device: a 100 000 x 30 000 call held ~218 GB then, so every call ended over that cap and the NEXT call began with
// 1.2 s of hipMalloc (profiles/r04_pool_cap.log). The cache is only an optimisation for back-to-back calls: api.sclens() and the
// Julia shim trim it when a call returns unless the host asks to keep it warm, comm_create trims it before RCCL allocates,
// sclens_hip_trim() is the public hook, and a failed pool_malloc trims and retries.
std::map<int, size_t> g_caps;  // under g_mu
// pending: bytes of a block that is being freed right now (already off `live`, still allocated as far as hipMemGetInfo can see)
size_t pool_cap_bytes(int dev, size_t pending = 0) {
  auto it = g_caps.find(dev);
  if (it != g_caps.end()) return it->second;
  size_t cap = (size_t)64 << 30;
  static const char* e = getenv("SCLENS_HIP_POOL_MAX_GB");
  if (e) {
    cap = (size_t)atoll(e) << 30;
  } else {
    size_t fr = 0, tot = 0;
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != dev) hipSetDevice(dev);
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0) {
      const size_t reserve = std::max<size_t>(tot / 8, std::min<size_t>((size_t)24 << 30, tot / 2));
      // what this process could use: free now + what the library itself holds -- idle in the cache, handed out (live), and the block in
      // hand. (Without the last two a first free under tens of GB of live blocks fixed the cap too low for the rest of the process.)
      const size_t avail = std::min(fr + g_pool[dev].cached + g_pool[dev].live + pending, tot);
      cap = avail > reserve ? avail - reserve : 0;
    }
    if (cur != dev) hipSetDevice(cur);
    (void)hipGetLastError();
  }
  g_caps[dev] = cap;
  return cap;
}
size_t size_class(size_t bytes) {
  if (bytes < 512) return 512;
  if (bytes < (1u << 20)) return (bytes + 511) & ~(size_t)511;
  return (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);  // 2 MB granules
}
void trim_locked(DevPool& dp) {
  for (auto& kv : dp.free_blocks) hipFree(kv.second);
  dp.free_blocks.clear();
  dp.cached = 0;
}
}  // namespace

hipError_t pool_malloc(void** p, size_t bytes) {
  *p = nullptr;
  if (!pool_enabled()) return hipMalloc(p, bytes ? bytes : 16);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t sz = size_class(bytes);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    DevPool& dp = g_pool[dev];
    auto it = dp.free_blocks.lower_bound(sz);
    // exact class, or a block at most 1/8 larger (the sizes of successive calls are identical; the slack only absorbs rounding)
    if (it != dp.free_blocks.end() && it->first <= sz + sz / 8) {
      *p = it->second;
      dp.cached -= it->first;
      dp.live += it->first;
      if (dp.live > dp.peak) dp.peak = dp.live;
      g_live[*p] = {dev, it->first};
      dp.free_blocks.erase(it);
      dp.hits += 1;
      return hipSuccess;
    }
    dp.misses += 1;
  }
  e = hipMalloc(p, sz);
  if (e != hipSuccess) {  // give the cache back and try once more
    {
      std::lock_guard<std::mutex> lk(g_mu);
      trim_locked(g_pool[dev]);
    }
    (void)hipGetLastError();
    e = hipMalloc(p, sz);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_pool[dev].live += sz;
  if (g_pool[dev].live > g_pool[dev].peak) g_pool[dev].peak = g_pool[dev].live;
  g_live[*p] = {dev, sz};
  return hipSuccess;
}

void pool_free(void* p, hipStream_t stream) {
  if (!p) return;
  if (stream) hipStreamSynchronize(stream);
  if (!pool_enabled()) {
    hipFree(p);
    return;
  }
  std::unique_lock<std::mutex> lk(g_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) {  // not ours (allocated before the pool was enabled / foreign): plain free
    lk.unlock();
    hipFree(p);
    return;
  }
  const int dev = it->second.first;
  const size_t sz = it->second.second;
  g_live.erase(it);
  DevPool& dp = g_pool[dev];
  dp.live -= sz;
  if (dp.cached + sz > pool_cap_bytes(dev, sz)) {
    lk.unlock();
    hipFree(p);
    return;
  }
  dp.free_blocks.emplace(sz, p);
  dp.cached += sz;
}

void pool_trim(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_pool)
    if (device < 0 || kv.first == device) {
      int cur = 0;
      hipGetDevice(&cur);
      hipSetDevice(kv.first);
      trim_locked(kv.second);
      hipSetDevice(cur);
    }
}

void pool_set_cap(int device, long long bytes) {  // bytes < 0: back to the default rule (evaluated again at the next free)
  std::lock_guard<std::mutex> lk(g_mu);
  if (bytes < 0) {
    g_caps.erase(device);
    return;
  }
  g_caps[device] = (size_t)bytes;
  DevPool& dp = g_pool[device];
  if (dp.cached > (size_t)bytes) {  // a lower cap takes effect at once: idle blocks above it go back to the driver, largest first
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != device) hipSetDevice(device);
    while (dp.cached > (size_t)bytes && !dp.free_blocks.empty()) {
      auto last = std::prev(dp.free_blocks.end());
      hipFree(last->second);
      dp.cached -= last->first;
      dp.free_blocks.erase(last);
    }
    if (cur != device) hipSetDevice(cur);
  }
}

size_t pool_peak(int device, bool reset) {  // largest number of bytes the library held at once on the device since the last reset
  std::lock_guard<std::mutex> lk(g_mu);
  DevPool& dp = g_pool[device];
  const size_t p = dp.peak;
  if (reset) dp.peak = dp.live;
  return p;
}

void pool_stats(int device, size_t* cached, size_t* live, size_t* hits, size_t* misses) {
  std::lock_guard<std::mutex> lk(g_mu);
  const DevPool& dp = g_pool[device];
  if (cached) *cached = dp.cached;
  if (live) *live = dp.live;
  if (hits) *hits = dp.hits;
  if (misses) *misses = dp.misses;
}

}  // namespace scl
