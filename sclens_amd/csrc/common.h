// Internal declarations shared by the HIP/C++ translation units of libsclens_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/sclens_hip.h"

namespace scl {

// ------------------------------------------------------------------ options of a context
// Every tunable of the library is a named integer of the context: sclens_hip_set_option(ctx, "name", value) / sclens_hip_get_option,
// sclens_hip_copy_options for the worker contexts of a call. A new context starts from the defaults below; SCLENS_HIP_OPTIONS
// ("name=value,name=value") is applied ONCE, at sclens_hip_create -- nothing in the library reads the environment per call.
// `precision` is the one a host normally touches (the reference's `device_` kwarg picks the arithmetic, scLENS.jl:649): 0 = every
// product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4: the reference GPU path's arithmetic), 1 = the large products
// from operands split into two fp16 pieces on the fp16 matrix cores (22-bit operands, fp32 accumulation; default). With precision = 0
// the *_split* thresholds below are ignored, q2_variant reads as 3, and gram_bits only keeps the exact co-occurrence product of
// binarised matrices (with 33-bit cell weights: gram_bits_strict).
#define SCL_OPTION_TABLE(X)                                                                                                        \
  X(precision, 1)                                                                                                                  \
  X(two_stage, -1)          /* eigensolver: 1 two-stage (sbr.hip), 0 one-stage (tridiag.hip), -1 by order */                       \
  X(two_stage_min_n, 8192)                                                                                                         \
  X(gram_bits, -1)          /* Gram of binarised matrices + search statistic on the fp16 MFMA: 1 always, 0 never, -1 by order */   \
  X(gram_bits_min_n, 16000)                                                                                                        \
  X(gram_bits_terms, 2)     /* fp16 pieces of the cell weights: 2 (22 bits) or 3 (33 bits) */                                      \
  X(gram_bits_strict, 1)    /* precision = 0: the Gram matrix of a BINARISED matrix still as the exact co-occurrence product, with   \
                               33-bit cell weights (no operand narrower than fp32: the 0/1 pattern is exact in fp16); 0: fp32 product */ \
  X(gram_split_min_n, 16000) /* dense Gram products from split operands from this order (0: never) */                              \
  X(gram_ksplit, 32768)     /* dense fp32 Gram products longer than 1.5 x this in slices of this many cells, added once (0: one chain) */ \
  X(gram_sparse, -1)        /* Gram matrices of cells > genes matrices from their sparse structure (gram_sparse.hip): 1 always,     \
                               0 never, -1 from gram_sparse_min_n AND where its multiply-adds (sum r_i^2 / 2) at the measured 4.3e11 \
                               per second undercut the dense product (131 / 366 TF/s): below ~7 % / ~4.5 % density at 100k x 30k */   \
  X(gram_sparse_min_n, 16000)                                                                                                      \
  X(implicit_min_n, 16000)  /* ensemble: the Gram matrix applied as two passes over the scaled matrix from this order */           \
  X(chefsi_b0, 0)           /* block size of the partial eigensolver (0: min_pc + 40 rounded up to 32) */                          \
  X(chefsi_tail_gap_micro, -1) /* gap-aware target of the tail pairs x 1e6 (-1: what the caller asks for) */                       \
  X(chefsi_lock, 1)                                                                                                                \
  X(chefsi_split, 1)        /* implicit block products (blocks of <= 64 rows) from split images on the 64 x 256 kernel (0: fp32) */  \
  X(chefsi_split_s1, 0) X(chefsi_split_s2, 0) /* split-K slices of its two products (0: chosen to fill whole rounds of CU slots) */   \
  X(sy2sb_split_min, 4096)  /* trailing updates of the band reduction from split operands from this many rows (0: never) */        \
  X(sy2sb_split_scales, 2)  /* 2: separate scales for reflector and Z columns, 1: one scale (round 3) */                           \
  X(sy2sb_zmax, 1)          /* largest |Z| from the kernel that writes Z (0: by a pass over the operands) */                       \
  X(sy2sb_wsplit_min, -1)   /* W = A22 V from pieces split in registers from this many rows (-1: sy2sb_split_min, 0: never) */     \
  X(sy2sb_lookahead, 1)                                                                                                            \
  X(sy2sb_delay, 1)         /* rank-256 updates for pairs of panels ... */                                                         \
  X(sy2sb_delay_min, 12288) /* ... while the trailing matrix has at least this many rows */                                        \
  X(sy2sb_fold_diag, 1)                                                                                                            \
  X(q1_split_min, 1024)     /* first back-transformation from split operands from this many vectors / rows (0: never) */           \
  X(q1_w1_split, 1)         /* its first product: 0 fp32, 1 Z split in registers, 2 Z through a split image */                     \
  X(q1_prep, 1)             /* block reflectors of all groups prepared beside the chase */                                         \
  X(q1_group, 0)            /* panels per block reflector: 4, 8 or 0 = by size */                                                  \
  X(chase_mb, 1)            /* bulge chase by messages + prefetch (0: the round-2 kernel, the bitwise reference) */                \
  X(chase_wgs, 0)           /* cap on its workgroups (0: one per CU) */                                                            \
  X(q2_variant, 16)         /* second back-transformation: 16 / 15 / 14 image-fed (passes of 8 / 4 blocks; one group ahead), 3 fp32 */ \
  X(q2_reference, 0)        /* 1: the unblocked reference kernel (tests) */                                                        \
  X(q2_fp32_blocks, 4)      /* fp32 Q2 kernel: blocks of 32 sweeps per pass (4, 8, 12, 16): longer is faster ALONE, 4 inside a call */  \
  X(q2_tg_early, 1)         /* its group data built on the auxiliary stream beside the inverse iteration */                        \
  X(stein_pf, 16)           /* inverse iteration: steps of loads in flight (4, 16, 32) */                                          \
  X(stein_its, 2)           /* growth-checked iterations before a vector is accepted (dstein: 3) */                                \
  X(stein_shared, 0)        /* its [n][batch] workspaces: 0 one set per context; 1 ONE per device, used in turn (-18 GB, +0.9 % time) */ \
  X(bisect_div, 0)          /* 1: Sturm counts in the ratio form (round 2) */                                                      \
  X(split_pipe, 1)          /* stage loop of the split-fp16 products as a software pipeline (0: the two-buffer loop of round 4) */   \
  X(split_acc_init, 1)      /* split updates start their accumulators from C (0: C added in the epilogue) */                       \
  X(dense_fused, 1)         /* scaled matrix written in one pass per gene (0: fill + scatter kernels) */                           \
  X(host_pattern, 0)        /* 1: sparse pattern built on the host (tests compare the two builders) */                             \
  X(val_csr, 1)             /* CSR-ordered companion copies of the value arrays */                                                 \
  X(chunk_cache_gb, -1)     /* chunked session: device memory for chunk patterns kept between visits (the rest are rebuilt);       \
                               -1: the device's memory less 200 GiB for everything else (88 GiB on an MI355X) */                     \
  X(gemm_force, 0)          /* tests: 1 the large-tile kernels on small shapes, 2 the 128 x 128 kernel on every shape */           \
  X(panel_prof, 0) X(chase_prof, 0) X(q2_prof, 0) /* per-phase shader clocks on stderr (diagnostic builds of the same kernels) */  \
  X(debug, 0)

struct Options {
#define X(name, def) int64_t name = def;
  SCL_OPTION_TABLE(X)
#undef X
  bool set(const std::string& key, int64_t v) {
#define X(name, def) \
  if (key == #name) { name = v; return true; }
    SCL_OPTION_TABLE(X)
#undef X
    return false;
  }
  bool get(const std::string& key, int64_t* v) const {
#define X(name, def) \
  if (key == #name) { *v = name; return true; }
    SCL_OPTION_TABLE(X)
#undef X
    return false;
  }
  // "a=1,b=2": unknown names and malformed items are reported in *bad (comma separated), the rest is applied
  void parse(const char* text, std::string* bad) {
    std::string s(text ? text : "");
    size_t i = 0;
    while (i < s.size()) {
      size_t j = s.find(',', i);
      if (j == std::string::npos) j = s.size();
      const std::string item = s.substr(i, j - i);
      const size_t eq = item.find('=');
      bool ok = false;
      if (eq != std::string::npos && eq > 0 && eq + 1 < item.size()) {
        char* end = nullptr;
        const long long v = strtoll(item.c_str() + eq + 1, &end, 10);
        ok = end && *end == 0 && set(item.substr(0, eq), (int64_t)v);
      }
      if (!ok && !item.empty() && bad) *bad += (bad->empty() ? "" : ",") + item;
      i = j + 1;
    }
  }
  // ---- effective switches: `precision == 0` keeps every product on the fp32 matrix cores
  bool split() const { return precision != 0; }
  static int64_t from(int64_t min_n) { return min_n > 0 ? min_n : ((int64_t)1 << 60); }  // "0 = never" thresholds
  int64_t eff_gram_split_min() const { return split() ? from(gram_split_min_n) : from(0); }
  int64_t eff_sy2sb_split_min() const { return split() ? from(sy2sb_split_min) : from(0); }
  int64_t eff_sy2sb_wsplit_min() const { return split() ? (sy2sb_wsplit_min < 0 ? from(sy2sb_split_min) : from(sy2sb_wsplit_min)) : from(0); }
  int64_t eff_q1_split_min() const { return split() ? q1_split_min : 0; }
  int eff_gram_bits() const { return split() ? (gram_bits < 0 ? -1 : (gram_bits != 0)) : 0; }
  // the co-occurrence form of a binarised matrix's Gram product alone (session.hip, use_gram_bits): also with precision = 0, where
  // its operands are no narrower than fp32 -- the pattern is 0/1, the cell weights carry three fp16 pieces = 33 bits
  int eff_gram_binary() const { return split() ? eff_gram_bits() : ((gram_bits_strict != 0 && gram_bits != 0) ? (gram_bits < 0 ? -1 : 1) : 0); }
  int eff_gram_bits_terms() const { return split() ? (gram_bits_terms == 3 ? 3 : 2) : 3; }
  int eff_q2_variant() const { return split() ? (int)q2_variant : 3; }
  int eff_two_stage() const { return two_stage < 0 ? -1 : (two_stage != 0); }
};

struct Ctx {
  int device = 0;
  Options opt;
  hipStream_t stream = nullptr;
  // second stream + events for work that overlaps inside one call (look-ahead of the band reduction); created on first use
  hipStream_t aux_stream = nullptr;
  hipStream_t swapped_main = nullptr;  // the main stream while `stream` temporarily names the auxiliary one (sbr_q1_prepare)
  hipEvent_t aux_ev[2] = {nullptr, nullptr};
  // T factors of the second back-transformation, built on the auxiliary stream right after the bulge chase (sbr.hip): the event
  // that marks them complete and the order they were built for (-1: none / not valid for the current reflectors)
  hipEvent_t q2_ev = nullptr;
  int64_t q2_tg_n = -1;
  int q2_built_variant = -1;  // opt.q2_variant the group data (T factors or LDS images) was last built for
  // block reflectors of the first back-transformation prepared on the auxiliary stream right after the band reduction (sbr.hip,
  // sbr_q1_prepare): the order / panels per group they were built for (-1: none) and the event that marks them complete
  int64_t q1p_n = -1;
  int q1p_g = 0;
  hipEvent_t q1_ev = nullptr;
  bool q2_prebuild = true;  // cleared by a caller that will ask for eigenvalues only (the null matrix)
  std::string err;
  // grow-only named device workspaces (freed at destroy); avoids hipMalloc inside hot loops
  std::map<std::string, std::pair<void*, size_t>> ws;
  long ws_epoch = 0;  // bumped whenever a workspace is released: holders of derived data (a session's split image of Vr2) re-derive
  // per-stage timing (ms) accumulated when timing is enabled
  bool timing = false;
  std::map<std::string, double> t_ms;
  std::map<std::string, long> t_calls;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // optional per-launch HIP-event timing of the dominant kernel (trd_colB): bench.py's roofline leg
  bool prof_symv = false;
  std::vector<hipEvent_t> prof_ev;   // pairs (start, stop) recorded on `stream`
  size_t prof_used = 0;
  double prof_bytes = 0.0;           // algorithmic bytes of the recorded launches
  // state between the two calls of the QC filter (preprocess.hip)
  struct PpState {
    bool valid = false;
    int64_t N = 0, M = 0, n_cells = 0, n_genes = 0, nnz_out = 0;
  } pp;
  // which eigensolver holds the state eig_vectors continues from (opt.two_stage selects it); calls of gram_binary
  bool last_two_stage = false;
  long gram_bits_used = 0;
  long gram_sparse_used = 0;
  // sessions keep their eigenvector / ensemble buffers in this context's named workspaces ("ses.*", "eig.*"): one live
  // session per context (worker sessions of session_clone bring their own context)
  int live_sessions = 0;
  // kernels whose dynamic LDS limit has been raised above 64 KB on this context's device (the attribute is per device)
  std::map<const void*, int> lds_attr;

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  // returns nullptr on OOM (err set)
  void* workspace(const std::string& name, size_t bytes);
  void release(const std::string& name);
  void release_all();
};

// Everything a context may still have in flight on a block it is about to hand back: BOTH of its streams (the look-ahead of the band
// reduction and the group data of the second back-transformation run on the auxiliary one). hipFree used to synchronise the whole
// device as a side effect; the pool (pool.hip) does not, so every path that returns a block while work may be queued calls this
// first -- in particular the error paths, which leave a function in the middle of a sequence of launches.
static inline void ctx_quiesce(Ctx* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
  if (c->swapped_main) (void)hipStreamSynchronize(c->swapped_main);
}

#define SCL_HIP(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      return (ctx)->fail(e__ == hipErrorOutOfMemory ? SCLENS_ERR_OOM : SCLENS_ERR_HIP,           \
                         std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ +  \
                             ":" + std::to_string(__LINE__) + ")");                              \
    }                                                                                            \
  } while (0)

#define SCL_TRY(expr)                 \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != SCLENS_OK) return rc__; \
  } while (0)

#define SCL_WS(ctx, var, type, name, count)                                                     \
  type* var = static_cast<type*>((ctx)->workspace((name), sizeof(type) * (size_t)(count)));     \
  if (!(var) && (count) > 0) return SCLENS_ERR_OOM

struct StageTimer {  // HIP-event timing of one stage on ctx->stream (only when ctx->timing)
  Ctx* c;
  const char* name;
  StageTimer(Ctx* ctx, const char* n) : c(ctx), name(n) {
    if (c->timing) hipEventRecord(c->ev0, c->stream);
  }
  ~StageTimer() {
    if (c->timing) {
      hipEventRecord(c->ev1, c->stream);
      hipEventSynchronize(c->ev1);
      float ms = 0;
      hipEventElapsedTime(&ms, c->ev0, c->ev1);
      c->t_ms[name] += ms;
      c->t_calls[name] += 1;
    }
  }
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------ device memory pool (pool.hip)
// every device allocation of the library; `stream` = the stream whose pending work may still touch the block (synchronised
// before the block can be handed to another owner; nullptr: the caller has synchronised already)
hipError_t pool_malloc(void** p, size_t bytes);
void pool_free(void* p, hipStream_t stream);
void pool_trim(int device);  // give cached blocks back to the driver (device < 0: all devices)
void pool_set_cap(int device, long long bytes);
size_t pool_peak(int device, bool reset);  // peak of the live bytes since the last reset  // idle bytes kept per device (bytes < 0: the default rule)
void pool_stats(int device, size_t* cached, size_t* live, size_t* hits, size_t* misses);

// raise a kernel's dynamic-LDS limit once per context (= per device; a process-wide `static` would cover only the first device)
static inline int ensure_dyn_lds(Ctx* ctx, const void* fn, int bytes) {
  auto it = ctx->lds_attr.find(fn);
  if (it != ctx->lds_attr.end() && it->second >= bytes) return SCLENS_OK;
  SCL_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->lds_attr[fn] = bytes;
  return SCLENS_OK;
}

// ------------------------------------------------------------------ GEMM (gemm.hip)
// C[M x N] (ldc) = alpha * P[M x K] * op(Q) + beta * C, everything row-major fp32.
//   P is [M][K] with K contiguous (ldp).
//   q_kcontig = 1: Q is [N][K] with K contiguous (ldq)   ("NT":  C = P * Q^T)
//   q_kcontig = 0: Q is [K][N] with N contiguous (ldq)   ("NN":  C = P * Q)
//   lower = 1: M == N, only tiles on/below the diagonal are computed and every value is also
//              written to its mirrored position (exactly symmetric result).
//   colabsmax != nullptr: C is not written; colabsmax[n] = max(colabsmax[n], max_m |alpha*acc|)
//              (caller zero-fills colabsmax; values are float bit patterns of non-negative floats).
struct GemmArgs {
  const float* P;
  const float* Q;
  float* C;
  int64_t M, N, K;
  int64_t ldp, ldq, ldc;
  float alpha, beta;
  int q_kcontig;
  int lower;
  unsigned* colabsmax;
  // split-K inside one launch (grid.y = splits): slice s covers K range [s*k_chunk, min(K, (s+1)*k_chunk)) and
  // writes its own output C + s*c_split_off (partials are summed by the consumer in a fixed order).
  // splits <= 1 means a plain GEMM. k_chunk must be a multiple of 16.
  int splits = 1;
  int64_t k_chunk = 0;
  int64_t c_split_off = 0;
  // ask for the large-tile NT kernels (256 x 256, or 256 x 64 when N <= 64) even for a short K / fewer tiles: callers whose
  // product is bound by the C traffic or runs alone on the GPU (band reduction, back-transformations)
  int prefer_big = 0;
  // C += P Q' with C loaded into the accumulators before the contraction (alpha == beta == 1 required): the large-tile kernels
  // then read C while the first operand stage is in flight and their epilogue only stores. Ignored by the 128 x 128 kernel's
  // arithmetic (it applies alpha / beta as usual: same result up to the order of the fp32 additions).
  int acc_init = 0;
  int stagger_ns = 0;  // set by gemm_f32: period over which the first workgroups of the CUs are staggered (see gemm_nt_big)
};
int gemm_f32(Ctx* ctx, const GemmArgs& a);

// fp32-accurate NT products on the fp16 MFMA from operands split into two fp16 pieces (gram_bits.hip): the split image of
// a row-major [rows][K] matrix, and colabsmax[j] = max(colabsmax[j], max_i |A_i . B_j|) from two images
size_t split_image_bytes(int64_t rows, int64_t K);
int split_image_f16(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst);
int corr_colabsmax_split(Ctx* ctx, const void* Aimg, int64_t M, const void* Bimg, int64_t N, int64_t K, unsigned* colabsmax);
// C (+)= P Q' for short contractions with the products on the fp16 matrix cores (round 3: trailing updates of the band reduction,
// the read-modify-write product of the first back-transformation). split_image_scaled: the split image of src [rows][K] scaled by
// the power of two that brings its largest |entry| to [2^13, 2^14) (scale[0] receives it, device memory; scale[1] is scratch);
// gemm_split_update: C[M][N] += P Q' (lower != 0: M == N, tiles on / below the diagonal, mirrored) from two such images and
// their scales, accumulators started from C -- the arithmetic of `acc_init` in gemm.hip with 22-bit operands.
int split_image_scaled(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* scale_dev);
// the same under a FIXED power-of-two scale (no pass over the data): operands whose entries are at most 1 in magnitude
int split_image_fixed(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* scale_dev, float scale);
// one power-of-two scale PER ROW (largest |entry| of the row -> [2^13, 2^14)); inv_scale_dev[r] = 1 / scale of row r (exact)
int split_image_rows(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, float* inv_scale_dev);
// the split image of the TRANSPOSE of src [rows][K] (K image rows of round_up(rows, 32) entries) under the scale scale_dev[0]
int split_image_transposed(Ctx* ctx, const float* src, int64_t rows, int64_t K, int64_t ld, void* dst, const float* scale_dev);
// C_s[M <= 64][N] = post * rowscale[m] * A B' over K-slice s from split images: 64 x 256 tiles, two workgroups per CU, for products that
// stream a large B against a block of at most 64 rows (slab s at C + s * c_split_off; C is not read)
int gemm_split_skinny(Ctx* ctx, const void* Aimg, const float* sA, const float* rowscale, int64_t M, const void* Bimg, const float* sB, int64_t N,
                      int64_t K, float* C, int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off, float post);
// C_s = P Q' over the K-slice s of `splits` (slab s at C + s * c_split_off, row pitch ldc; C is not read): split-K partials
int gemm_split_nt(Ctx* ctx, const void* Pimg, const float* sP, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K, float* C,
                  int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off);
int gemm_split_nt_f32a(Ctx* ctx, const float* P, int64_t ldp, float p_scale, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K,
                       float* C, int64_t ldc, int splits, int64_t k_chunk, int64_t c_split_off);
// two matrices of the same shape that hold the same magnitudes (the two operands of a symmetric rank-2k update): one scale, from src1
int split_image_pair_scaled(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, void* dst1, void* dst2,
                            float* scale_dev);
// two kinds of columns alternating in blocks of `half`, one scale each (scale_dev: 4 floats; pass scale_dev and scale_dev + 2 to
// gemm_split_update as sP and sQ)
int split_image_pair_zmax(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, int half, void* dst1,
                          void* dst2, float* scale_dev, const unsigned* zmax_dev, int nz);
int split_image_pair_scaled2(Ctx* ctx, const float* src1, const float* src2, int64_t rows, int64_t K, int64_t ld, int half, void* dst1,
                             void* dst2, float* scale_dev);
int gemm_split_update(Ctx* ctx, const void* Pimg, const float* sP, int64_t M, const void* Qimg, const float* sQ, int64_t N, int64_t K,
                      float* C, int64_t ldc, int lower, float post = 1.f);  // C += post * P Q'; long contractions walk the tile list

// ------------------------------------------------------------------ eigensolver (tridiag.hip, trieig.hip)
// Symmetric eigensolver on a device-resident n x n fp32 matrix A (row-major, lda, FULL storage,
// exactly symmetric). A is destroyed (holds the Householder reflectors afterwards).
//   w64[n]      : all eigenvalues ascending (fp64, of the fp32-tridiagonalised matrix)
//   vec_lo/hi   : eigenvectors for ascending indices [vec_lo, vec_hi) are written to
//                 Zt[(idx - vec_lo) * ldz + i], i.e. one eigenvector per ROW (fp32).
int eigh_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, double* w64_dev, int64_t vec_lo,
             int64_t vec_hi, float* Zt, int64_t ldz);

// the same in two phases (values first, vectors of a range chosen afterwards); state lives in ctx workspaces
// n_low >= 0: only the n_low smallest eigenvalues and the largest one are computed (the others are NaN in w64_dev)
int eig_values(Ctx* ctx, float* A, int64_t n, int64_t lda, double* w64_dev, int64_t n_low = -1);
int eig_vectors(Ctx* ctx, const float* A, int64_t n, int64_t lda, const double* w64_dev, int64_t vec_lo,
                int64_t vec_hi, float* Zt, int64_t ldz);
int eig_values_redo_all(Ctx* ctx, int64_t n, double* w64_dev);
int eig_values_two_stage_redo(Ctx* ctx, int64_t n, double* w64_dev);

// pieces (exposed for unit tests through the C ABI)
int sytrd_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev,
              float* tau_dev);
// n_low >= 0: only the indices [0, n_low) and k_top (default n - 1); the other entries of w_dev become NaN
int stebz_f64(Ctx* ctx, const double* d_dev, const double* e_dev, int64_t n, double* w_dev, int64_t n_low = -1, int64_t k_top = -1);
int stein_f64(Ctx* ctx, const double* d_dev, const double* e_dev, int64_t n, const double* w_dev,
              int64_t lo, int64_t hi, float* Zt, int64_t ldz);
void stein_shared_release(int device, hipStream_t stream);
int ormtr_f32(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* tau_dev, float* Zt,
              int64_t m, int64_t ldz);

struct PpParams {  // keyword arguments of preprocess (scLENS.jl:160-162)
  double min_tp_c, min_tp_g, max_tp_c, max_tp_g;
  int64_t min_genes_per_cell, max_genes_per_cell, min_cells_per_gene;
  double mito_percent, ribo_percent;
};
int preprocess_stats(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                     const uint8_t* is_mito, const uint8_t* is_ribo, const PpParams& P, uint8_t* keep_cell,
                     int64_t* gene_order, int64_t* n_cells, int64_t* n_genes, int64_t* nnz_out);
int preprocess_gather(Ctx* ctx, int64_t* out_colptr, int32_t* out_rowval, float* out_nzval);
int sy2sb_f32(Ctx* ctx, float* A, int64_t n, int64_t lda, float* Tall, int* breakdown);
int sbr_apply_q1(Ctx* ctx, const float* A, int64_t n, int64_t lda, const float* Tall, float* Zt, int64_t m, int64_t ldz);
int sbr_apply_q2(Ctx* ctx, int64_t n, float* Zt, int64_t m, int64_t ldz);
int eig_values_two_stage(Ctx* ctx, const float* A, int64_t n, int64_t lda, double* w64_dev, int* used, int64_t n_low = -1);
int eig_vectors_two_stage(Ctx* ctx, int64_t n, int64_t vec_lo, int64_t vec_hi, float* Zt, int64_t ldz);
int sb2st_f32(Ctx* ctx, const float* A, int64_t n, int64_t lda, double* d_dev, double* e_dev);
int symv_probe(Ctx* ctx, int64_t n, int64_t* launches, double* total_ms, double* total_bytes);

// ------------------------------------------------------------------ partial eigensolver (chefsi.hip)
// Top-m eigenpairs of a symmetric PSD matrix A (n x n fp32, row-major, lda; NOT modified) by Chebyshev-filtered
// subspace iteration with Rayleigh-Ritz, started from X0t (b rows of n: approximate leading eigenvectors) and
// theta0[b] (their eigenvalue estimates, descending); the first m_strict pairs get the tight residual target. On success (*converged = 1) w_desc[m] (host) holds the m largest
// eigenvalues (descending) and Zt rows 0..m-1 (device, ldz) the unit eigenvectors in the same order.
// Bop != nullptr: A is given implicitly as Bop Bop' / div with Bop [n x Kop] row-major (ldb) -- the Gram matrix is not needed
// (A may be nullptr); every block product then costs two passes over Bop.
// tail_gap > 0: the pairs m_strict .. m-1 are additionally held to tail_gap x (theta_q - smallest Ritz value of the block).
// tail_free != 0: the pairs m_strict .. m-1 are not part of the convergence test at all (Ritz pairs of the block as they are).
int topk_chefsi(Ctx* ctx, const float* A, int64_t n, int64_t lda, int m, int m_strict, int b, const float* X0t, int64_t ldx,
                const double* theta0, double* w_desc, float* Zt, int64_t ldz, int* converged, int* iters,
                const float* Bop = nullptr, int64_t Kop = 0, int64_t ldb = 0, float div = 1.f, double tail_gap = 0.0, int tail_free = 0);

// ------------------------------------------------------------------ small device helpers (util.hip)
int fill_f32(Ctx* ctx, float* p, int64_t n, float v);
int normalize_rows_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld);
int scale_rows_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld, const float* s);
int row_sqnorms_f32(Ctx* ctx, const float* A, int64_t rows, int64_t cols, int64_t ld, double* out);
int scale_rows_rsqrt_f32(Ctx* ctx, float* A, int64_t rows, int64_t cols, int64_t ld, const double* sq);
// out[c*ldo + r] = in[r*ldi + c]
int transpose_f32(Ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo);
// out row q = in row (rows-1-q)
int reverse_rows_f32(Ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo);
// A (n x n, lda, zero padded) = B B^T / divisor for B [n x K] row-major (ldb), exactly symmetric
// accumulate: A += B B^T / divisor (A must hold an exactly symmetric matrix; chunked sessions sum the contributions of their cell chunks)
int gram_f32(Ctx* ctx, const float* B, int64_t n, int64_t K, int64_t ldb, float divisor, float* A, int64_t lda, bool allow_split = true,
             bool accumulate = false);

}  // namespace scl

// the opaque context handle of the C ABI (capi.hip, comm.hip)
struct sclens_hip_ctx {
  scl::Ctx c;
};
