// Context, pattern builder, the device-resident session behind sclens() and the per-call drop-ins.
// Reference call sites: sclens scLENS.jl:649-832, get_sigev :526-594, get_eigvec :489-524,
// _wishart_matrix :332-361, _get_eigen :375-387, corr_mat :363-373.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

#include "common.h"
#include "pattern.h"
#include "pattern_host.h"
#include "rng.h"

namespace scl {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ ctx
void* Ctx::workspace(const std::string& name, size_t bytes) {
  auto it = ws.find(name);
  if (it != ws.end() && it->second.second >= bytes) return it->second.first;
  if (it != ws.end()) {
    ctx_quiesce(this);  // regrow: kernels of either stream may still use the old block
    pool_free(it->second.first, nullptr);
    ws.erase(it);
  }
  if (bytes == 0) bytes = 16;
  void* p = nullptr;
  hipError_t e = pool_malloc(&p, bytes);
  if (e != hipSuccess) {
    fail(SCLENS_ERR_OOM, "hipMalloc(" + name + ", " + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    return nullptr;
  }
  ws[name] = {p, bytes};
  return p;
}
void Ctx::release(const std::string& name) {
  auto it = ws.find(name);
  if (it != ws.end()) {
    ctx_quiesce(this);
    pool_free(it->second.first, nullptr);
    ws.erase(it);
    ws_epoch += 1;
  }
}
void Ctx::release_all() {
  if (stream) hipStreamSynchronize(stream);
  // work enqueued on the auxiliary stream is normally joined back into `stream` by an event, except the T factors of the second
  // back-transformation that a values-only decomposition builds ahead and nobody consumes: their kernel must not outlive its
  // workspace (the block goes back to the pool and may be handed to another context at once)
  if (aux_stream) hipStreamSynchronize(aux_stream);
  for (auto& kv : ws) pool_free(kv.second.first, nullptr);
  ws.clear();
}

// ------------------------------------------------------------------------------------------------ utils
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, int64_t rows, int64_t cols,
                                                   int64_t ldi, float* __restrict__ out, int64_t ldo) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(r0 + r) * ldi + c0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < cols && r0 + tx < rows) out[(c0 + r) * ldo + r0 + tx] = tile[tx][r];
}
int transpose_f32(Ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0) return SCLENS_OK;
  const int64_t gy = (rows + 31) / 32;
  if (gy > 65535) {
    // slab the row dimension
    for (int64_t r0 = 0; r0 < rows; r0 += 65535LL * 32) {
      const int64_t rr = std::min<int64_t>(rows - r0, 65535LL * 32);
      hipLaunchKernelGGL(k_transpose, dim3((unsigned)((cols + 31) / 32), (unsigned)((rr + 31) / 32)), dim3(256), 0,
                         ctx->stream, in + r0 * ldi, rr, cols, ldi, out + r0, ldo);
    }
  } else {
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((cols + 31) / 32), (unsigned)gy), dim3(256), 0, ctx->stream, in,
                       rows, cols, ldi, out, ldo);
  }
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}
__global__ void k_reverse_rows(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t ldi,
                               float* __restrict__ out, int64_t ldo) {
  const int64_t q = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cols) out[q * ldo + c] = in[(rows - 1 - q) * ldi + c];
}
int reverse_rows_f32(Ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0) return SCLENS_OK;
  hipLaunchKernelGGL(k_reverse_rows, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream,
                     in, rows, cols, ldi, out, ldo);
  SCL_HIP(ctx, hipGetLastError());
  return SCLENS_OK;
}

// y += x, both n x lda fp32 with lda a multiple of 4 (the padding is zero on both sides): one rounding per entry
__global__ void k_add_f32(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(x + i), b = *reinterpret_cast<const f32x4_t*>(y + i);
    *reinterpret_cast<f32x4_t*>(y + i) = a + b;
  } else {
    for (int64_t q = i; q < n; ++q) y[q] += x[q];
  }
}
// allow_split = false: the per-call drop-ins of the reference's functions (`_wishart_matrix`, `get_eigvec`) -- a caller that asks
// for THE fp32 product gets it at every size unless the context option gram_bits = 1 asks for the accelerated form explicitly
int gram_f32(Ctx* ctx, const float* B, int64_t n, int64_t K, int64_t ldb, float divisor, float* A, int64_t lda, bool allow_split,
             bool accumulate) {
  StageTimer tm(ctx, "gram");
  if (!accumulate) SCL_HIP(ctx, hipMemsetAsync(A, 0, sizeof(float) * (size_t)n * lda, ctx->stream));
  // Large products (n >= 16 000, or context option gram_split_min_n = <n>; 0 = never): the scaled matrix is split once into two fp16 pieces per
  // entry (scaled by the power of two that brings its largest entry to 2^13..2^14: 22 significant bits of every entry down to
  // 2^-38 of the largest) and the product runs on the fp16 matrix cores with fp32 accumulation (gemm_split_update, gram_bits.hip):
  // 682 -> ~250 ms at 100 000 x 30 000. Context option gram_bits = 0 (bench.py's strict step) also keeps this product in fp32.
  const int gb = ctx->opt.eff_gram_bits();
  if (n >= ctx->opt.eff_gram_split_min() && gb != 0 && (allow_split || gb == 1)) {
    void* img = ctx->workspace("gram.img", split_image_bytes(n, K));
    float* sc = static_cast<float*>(ctx->workspace("gram.sc", 4 * sizeof(float)));
    if (!img || !sc) return SCLENS_ERR_OOM;
    SCL_TRY(split_image_scaled(ctx, B, n, K, ldb, img, sc));
    return gemm_split_update(ctx, img, sc, n, img, sc, n, K, A, lda, 1, 1.0f / divisor);
  }
  GemmArgs g{};
  g.P = B; g.Q = B; g.C = A;
  g.M = n; g.N = n; g.K = K;
  g.ldp = ldb; g.ldq = ldb; g.ldc = lda;
  g.alpha = 1.0f / divisor; g.beta = accumulate ? 1.f : 0.f;
  g.q_kcontig = 1; g.lower = 1; g.colabsmax = nullptr;
  // A long contraction in slices (context option gram_ksplit, default 32 768): every accumulator of the fp32 product is ONE chain of K
  // additions, and once it has grown the small terms of a dense scaled matrix (its background -s_i mu_j - cent_j) drop out of its
  // mantissa -- at K = 100 000 the diagonal entries of sparsely expressed genes come out up to 8e-4 low (round 6,
  // profiles/r06_gram_sparse_ab_cfg4.log). Slices are formed on their own and added with one rounding each: the chain is K / slices long.
  const int64_t ks = ctx->opt.gram_ksplit;
  if (ks >= 1024 && K > ks + ks / 2 && n >= 1024) {
    float* part = static_cast<float*>(ctx->workspace("gram.kpart", sizeof(float) * (size_t)n * lda));
    if (!part) return SCLENS_ERR_OOM;
    const int64_t step = round_up(ks, 32);
    for (int64_t k0 = 0; k0 < K; k0 += step) {
      const bool direct = k0 == 0 && !accumulate;  // the first slice of a fresh product goes straight to A
      if (!direct) SCL_HIP(ctx, hipMemsetAsync(part, 0, sizeof(float) * (size_t)n * lda, ctx->stream));
      g.P = B + k0; g.Q = B + k0;
      g.K = std::min(step, K - k0);
      g.C = direct ? A : part;
      g.beta = 0.f;
      SCL_TRY(gemm_f32(ctx, g));
      if (!direct) {
        const int64_t cnt = n * lda;
        hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, ctx->stream, part, cnt, A);
        SCL_HIP(ctx, hipGetLastError());
      }
    }
    return SCLENS_OK;
  }
  return gemm_f32(ctx, g);
}

// ------------------------------------------------------------------------------------------------ pattern
template <typename T>
static int upload(Ctx* ctx, PatternOwner* o, const std::vector<T>& h, const T** dev) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(16, h.size() * sizeof(T));
  hipError_t e = pool_malloc(&p, bytes);
  if (e != hipSuccess) return ctx->fail(SCLENS_ERR_OOM, std::string("pattern upload: ") + hipGetErrorString(e));
  o->allocs.push_back(p);
  // on the context's own stream (pattern_build synchronises it before the host vectors die): a pattern may be built
  // on an auxiliary context while other contexts run decompositions, and must not serialise with them
  if (!h.empty()) SCL_HIP(ctx, hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  *dev = static_cast<const T*>(p);
  return SCLENS_OK;
}

int pattern_build(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                  int64_t ncand, const uint32_t* z1, const uint32_t* z2, PatternOwner* out, int64_t row0, int64_t N_global) {
  if (N <= 0 || M <= 0 || !colptr) return ctx->fail(SCLENS_ERR_ARG, "pattern_build: bad arguments");
  if ((colptr[M] > 0 && (!rowval || !nzval)) || (ncand > 0 && (!z1 || !z2)))
    return ctx->fail(SCLENS_ERR_ARG, "pattern_build: bad arguments");
  if (N_global <= 0) N_global = N;
  if (row0 < 0 || row0 + N > N_global) return ctx->fail(SCLENS_ERR_ARG, "pattern_build: bad row range");
  // all cells in this session: the pattern is built on the device (pattern_dev.hip: identical arrays, no host passes);
  // context option host_pattern = 1 keeps the host builder (tests compare the two)
  if (row0 == 0 && N_global == N && !ctx->opt.host_pattern)
    return pattern_build_device(ctx, N, M, colptr, rowval, nzval, ncand, z1, z2, 0, 0, out);
  HostPattern hp;
  std::string herr;
  if (pattern_build_host(N, M, colptr, rowval, nzval, ncand, z1, z2, row0, N_global, 0, &hp, &herr) != SCLENS_OK)
    return ctx->fail(SCLENS_ERR_ARG, herr);
  const int64_t nU = hp.nU;
  const std::vector<int64_t>&ucol = hp.ucol, &cpos = hp.cpos, &rptr = hp.rptr, &c2c = hp.c2c;
  const std::vector<int32_t>&urow = hp.urow, &ccol = hp.ccol;
  const std::vector<float>& uval = hp.uval;
  out->dev.N = N; out->dev.M = M; out->dev.nU = nU; out->dev.ncand = ncand;
  SCL_TRY(upload(ctx, out, ucol, &out->dev.colptr));
  SCL_TRY(upload(ctx, out, urow, &out->dev.row));
  SCL_TRY(upload(ctx, out, rptr, &out->dev.rowptr));
  SCL_TRY(upload(ctx, out, c2c, &out->dev.csr2csc));
  SCL_TRY(upload(ctx, out, ccol, &out->dev.csrcol));
  SCL_TRY(upload(ctx, out, cpos, &out->dev.cand_pos));
  const float* bv = nullptr;
  SCL_TRY(upload(ctx, out, uval, &bv));
  out->base_val = const_cast<float*>(bv);
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return pattern_add_csr_companions(ctx, out);
}
void pattern_free(PatternOwner* p, Ctx* busy) {
  // contract of the normal path: no session uses the pattern any more and the blocking API has left its streams idle (busy == nullptr).
  // Error paths leave a builder between launches: they pass the context whose streams may still hold work on the blocks.
  ctx_quiesce(busy);
  for (void* q : p->allocs) pool_free(q, nullptr);
  p->allocs.clear();
  p->dev = PatternDev();
  p->base_val = nullptr;
  p->z1_dev = p->z2_dev = nullptr;
}

// `L .> 0` (scLENS.jl:495, :515) with a rounding floor. For N <= M the mean-centred matrix has one structurally
// zero eigenvalue; in fp32 it comes out as +-(1e-8..1e-6) * lambda_max, and the reference keeps or drops that
// eigenvector by the sign of its rounding error (SURVEY 8a defect 6). Here it is always dropped.
//   factor = 8 (the drop-in get_eigvec, whose input carries no structure to rely on): positive means
//     lambda > 8 * eps32 * sqrt(n) * lambda_max (measured null |lambda| / lambda_max: 5e-7 at n = 10k).
//   Sessions (round 6): factor = 1 -- ten times the error the eigenvalues are measured to have (0.01 .. 0.1 sqrt(n) eps32 lambda_max
//     against float64 at n = 260 .. 30 000) -- and the structural zero dropped BY COUNT (`structural_zeros` smallest values are never
//     positive), so that the floor no longer has to clear the null value's rounding with a wide margin. The factor-8 floor is 7.7e-6
//     lambda_max at n = 260: a cells > genes matrix that is nearly square has GENUINE eigenvalues below that (6.9e-6 lambda_max in case
//     100 of `fuzz_parity.py 150 41`), the float64 oracle keeps them, the device dropped them from Vr2 only -- one ~0 entry entered
//     d_arr, the second smallest became the float64 run's smallest, and the search ended an evaluation early (4 of 300 random cases,
//     profiles/r06_fuzz_null_floor.md; the "unresolved pair" of round 5's sweep was this).
static int64_t count_positive_tol(const std::vector<double>& w, double factor = 8.0, int64_t structural_zeros = 0) {
  if (w.empty()) return 0;
  const double tol = factor * 5.96e-8 * std::sqrt((double)w.size()) * std::max(0.0, w.back());
  int64_t r = 0;
  for (double v : w) r += (v > tol) ? 1 : 0;
  return std::min<int64_t>(r, (int64_t)w.size() - structural_zeros);
}

// ------------------------------------------------------------------------------------------------ session
// Chunked session (cells > genes; all cells on this device, visited in chunks of rows): what a matrix of a chunked decomposition is
struct MatSpec {
  int set = 0;     // 0: the count matrix, 1: the null matrix X_r
  int cands = 0;   // the chunk patterns carry the zero candidates
  int binary = 0;  // stored counts as ones (scLENS.jl:664)
  int sample = 0;  // 0: no candidate ones, 1: the uploaded index list (idx_dev, m), 2: the keyed permutation (seed, m)
  uint64_t seed = 0;
  int64_t m = 0;
  int f32path = 1;
};
struct ChunkSrc {
  int64_t row0 = 0, N = 0;
  Counts* counts = nullptr;         // the chunk's cells as device CSC (owned)
  int64_t ncand = 0, cand_off = 0;  // its window of the global candidate list
};
struct ChunkPat {
  int set = 0, cands = 0, g = 0;
  PatternOwner pat;
  size_t bytes = 0;
};
struct ChunkStatsDev {  // statistics of one normalisation that span all cells (device, session-owned): what rebuilds a chunk's scaled block
  double *stdv = nullptr, *mu = nullptr, *cent = nullptr, *red = nullptr;  // red[0] = ||mu||^2
  double c = 0.0;                                                         // mean cell norm
  MatSpec spec;
  bool valid = false;
};

struct Session {
  Ctx* ctx = nullptr;
  int64_t N = 0, M = 0, n = 0, K = 0;
  int cells_major = 1;  // N <= M: rows of B are cells (Gram over genes); else rows are genes
  PatternOwner pat;
  float* val = nullptr;       // [nU] working value array
  float* Bmain = nullptr;     // scaled data matrix, [n][ldb]
  float* Btmp = nullptr;      // scaled null / binary / perturbed matrix
  // split fp16 image of Vr2t for the search statistic on the fp16 MFMA (gram_bits.hip); valid while Vr2h_of == Vr2t
  void* Vr2h = nullptr;
  const float* Vr2h_of = nullptr;
  long Vr2h_epoch = -1;  // ctx->ws_epoch when the image was built (a release_scratch in between may have taken the block)
  int64_t ldb = 0;
  float* A = nullptr;         // [n][lda] Gram / reflectors
  int64_t lda = 0;
  double* w64 = nullptr;      // [n] eigenvalues (device)
  std::vector<double> w_host;
  float* Zt = nullptr;        // eigenvector rows [cap][ldz]
  int64_t ldz = 0, zcap = 0;
  float* Vr2t = nullptr;      // [r][ldz]
  int64_t r_vr2 = 0;
  float* Z0t = nullptr;       // leading b0 eigenvectors of the data Gram matrix (descending), seed of the CheFSI block
  std::vector<double> theta0; // their eigenvalues (descending)
  int64_t b0 = 0;
  int use_chefsi = 1;
  int centering = 0;  // 0 mean, 1 median (scLENS.jl:651-654)
  // row-sharded session: N = local cells, K = local contraction length, Kdiv = the reference's divisor
  // size(X', 2) = number of ALL cells; sh.fn sums partial results over the ranks
  ShardReduce sh;
  int64_t Kdiv = 0;
  // first decompositions of a row-sharded session on ONE rank each instead of replicated (session_set_int "shard_rank", "solve_root"):
  // the partial Gram matrix is summed onto solve_root only, that rank runs the eigensolver, and what the others need of its result --
  // the eigenvalues, the few eigenvectors they recover their cells from, Vr2 -- reaches them as a sum in which they contribute zeros
  int shard_rank = -1, solve_root = -1;
  int data_root = -1;  // the rank that holds the data matrix's reduction (refine_eigenvalues / signal_vectors continue from it)
  bool solves(int root) const { return root < 0 || shard_rank == root; }
  int64_t chefsi_used = 0, chefsi_fallback = 0;
  double chefsi_tail_gap = 0.0;  // > 0: gap-aware targets for the tail pairs of the partial eigensolver (chefsi.hip)
  int chefsi_tail_free = 0;      // != 0: the tail pairs k .. min_pc-1 of an ensemble member are not converged at all (chefsi.hip); the caller
                                 // consults match_uncertain after session_robustness and solves the members it names again
  std::vector<int> match_uncertain;  // per member: the matching certificate of session_robustness did NOT hold
  float* nVt = nullptr;       // signal vectors, cell side, descending, [k][ldn]
  int64_t k = 0, ldn = 0;
  std::vector<float*> ens;    // slot t: [ncols][ldn], descending
  std::vector<int64_t> ens_cols;
  uint32_t* idx_dev = nullptr;
  int64_t idx_cap = 0;
  bool have_spectrum = false;
  std::vector<void*> allocs;
  // chunked session: N = all cells, the count matrix (and, during null_spectrum, X_r) as device-resident CSC chunks of rows; Btmp is ONE
  // chunk's scaled block [M][ldb], ldb = largest chunk rounded up; Bmain does not exist (blocks are rebuilt from st_data when needed)
  std::vector<ChunkSrc> chunks, null_chunks;
  std::vector<ChunkPat*> pcache;  // chunk patterns kept between visits (context option chunk_cache_gb); all of one (set, cands) kind
  size_t pcache_bytes = 0;
  bool chunk_committed = false, cands_counted = false;
  uint64_t cand_seed = 0;
  int64_t nnz_global = 0, ncand_total = 0;
  MatSpec cspec;  // the matrix the next decomposition is about
  ChunkStatsDev st_data, st_last;
  int64_t chunk_builds = 0, chunk_visits = 0;
  bool chunked() const { return !chunks.empty(); }

  int dmalloc(void** p, size_t bytes) {
    hipError_t e = pool_malloc(p, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return ctx->fail(SCLENS_ERR_OOM, std::string("session hipMalloc: ") + hipGetErrorString(e));
    allocs.push_back(*p);
    return SCLENS_OK;
  }
  int ensure_zt(int64_t rows) {
    if (rows <= zcap) return SCLENS_OK;
    float* p = static_cast<float*>(ctx->workspace("ses.Zt", sizeof(float) * (size_t)rows * ldz));
    if (!p) return SCLENS_ERR_OOM;
    Zt = p;
    zcap = rows;
    return SCLENS_OK;
  }
  int upload_idx(const uint32_t* h, int64_t m) {
    if (m > idx_cap) {
      uint32_t* p = static_cast<uint32_t*>(ctx->workspace("ses.idx", sizeof(uint32_t) * (size_t)m));
      if (!p) return SCLENS_ERR_OOM;
      idx_dev = p;
      idx_cap = m;
    }
    if (m > 0) SCL_HIP(ctx, hipMemcpyAsync(idx_dev, h, sizeof(uint32_t) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    return SCLENS_OK;
  }
  // partial: only the lowest eigenvalues and the largest one were computed (eig_values with n_low >= 0); the entries in
  // between come back as NaN and are known to lie between their neighbours: they count as positive (+inf here)
  int fetch_w(bool partial = false) {
    w_host.resize(n);
    SCL_HIP(ctx, hipMemcpyAsync(w_host.data(), w64, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n > 0 && w_host.back() != w_host.back()) return ctx->fail(SCLENS_ERR_NAN, "NaN eigenvalue");
    for (int64_t i = 0; i + 1 < n; ++i)
      if (w_host[i] != w_host[i]) {
        if (!partial || i == 0) return ctx->fail(SCLENS_ERR_NAN, "NaN eigenvalue");
        w_host[i] = w_host.back();  // somewhere in (last computed, largest]: the exact value is not consumed
      }
    return SCLENS_OK;
  }
  // one structural zero: the Gram matrix is taken over the cells of a mean-centred matrix (plain session, N <= M); row-sharded and
  // chunked sessions are cells > genes (their N is the local / global number of cells, the Gram matrix is genes x genes)
  int64_t count_positive() const {
    const bool cell_side = !sh.on() && !chunked() && N <= M;
    return count_positive_tol(w_host, 1.0, (cell_side && centering == 0) ? 1 : 0);
  }
};

// counts != nullptr: the count matrix is a device-resident sclens_hip_counts (colptr / rowval / nzval unused)
static int session_create_impl(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                               int64_t ncand, const uint32_t* z1, const uint32_t* z2, const Counts* counts, Session** out) {
  if (ctx->live_sessions > 0)
    return ctx->fail(SCLENS_ERR_STATE, "session_create: this context already has a live session (its workspaces are per context)");
  Session* s = new Session();
  s->ctx = ctx;
  s->N = N; s->M = M;
  s->n = std::min(N, M); s->K = std::max(N, M);
  s->Kdiv = s->K;
  s->cells_major = (N <= M) ? 1 : 0;
  int rc = counts ? pattern_build_device(ctx, N, M, counts->colptr, counts->row, counts->val, ncand, z1, z2, 0, 0, &s->pat, counts->nnz)
                  : pattern_build(ctx, N, M, colptr, rowval, nzval, ncand, z1, z2, &s->pat);
  if (rc != SCLENS_OK) { pattern_free(&s->pat, ctx); delete s; return rc; }
  s->ldb = round_up(s->K, 32);
  s->lda = round_up(s->n, 32);
  s->ldz = round_up(s->n, 32);
  s->ldn = round_up(N, 32);
  auto fail = [&](int code) { pattern_free(&s->pat, s->ctx); for (void* p : s->allocs) pool_free(p, nullptr); delete s; return code; };
  if ((rc = s->dmalloc((void**)&s->val, sizeof(float) * s->pat.dev.val_floats())) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Bmain, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Btmp, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->A, sizeof(float) * (size_t)s->n * s->lda)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->w64, sizeof(double) * s->n)) != SCLENS_OK) return fail(rc);
  s->ctx->live_sessions += 1;
  *out = s;
  return SCLENS_OK;
}
int session_create(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                   int64_t ncand, const uint32_t* z1, const uint32_t* z2, Session** out) {
  return session_create_impl(ctx, N, M, colptr, rowval, nzval, ncand, z1, z2, nullptr, out);
}
// the session of a count matrix that is already in HBM (preprocess -> sclens without the host round trip);
// counts only -- the zero candidates are attached later (session_set_pattern), as api.sclens does
int session_create_from_counts(Ctx* ctx, const Counts* c, Session** out) {
  if (!c || c->device != ctx->device) return ctx->fail(SCLENS_ERR_ARG, "session_create_from_counts: counts of another device");
  return session_create_impl(ctx, c->N, c->M, nullptr, nullptr, nullptr, 0, nullptr, nullptr, c, out);
}
int pattern_create_drawn_from_counts(Ctx* ctx, const Counts* c, uint64_t seed, PatternOwner** out, int64_t* ncand) {
  if (!c || c->device != ctx->device) return ctx->fail(SCLENS_ERR_ARG, "pattern_create_drawn_from_counts: counts of another device");
  PatternOwner* p = new PatternOwner();
  const int rc = pattern_build_device(ctx, c->N, c->M, c->colptr, c->row, c->val, 0, nullptr, nullptr, 1, seed, p, c->nnz);
  if (rc != SCLENS_OK) {
    pattern_free(p, ctx);
    delete p;
    return rc;
  }
  if (ncand) *ncand = p->dev.ncand;
  *out = p;
  return SCLENS_OK;
}
int counts_upload(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, Counts** out) {
  if (N <= 0 || M <= 0 || !colptr || colptr[M] < 0 || (colptr[M] > 0 && (!rowval || !nzval)))
    return ctx->fail(SCLENS_ERR_ARG, "counts_upload: bad arguments");
  Counts* c = new Counts();
  c->device = ctx->device; c->N = N; c->M = M; c->nnz = colptr[M];
  auto bail = [&](int code) { counts_free(c, ctx); return code; };
  if (pool_malloc((void**)&c->colptr, sizeof(int64_t) * (M + 1)) != hipSuccess || pool_malloc((void**)&c->row, sizeof(int32_t) * std::max<int64_t>(c->nnz, 4)) != hipSuccess ||
      pool_malloc((void**)&c->val, sizeof(float) * std::max<int64_t>(c->nnz, 4)) != hipSuccess)
    return bail(ctx->fail(SCLENS_ERR_OOM, "counts_upload: out of device memory"));
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemcpyAsync(c->colptr, colptr, sizeof(int64_t) * (M + 1), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && c->nnz > 0) e = hipMemcpyAsync(c->row, rowval, sizeof(int32_t) * c->nnz, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && c->nnz > 0) e = hipMemcpyAsync(c->val, nzval, sizeof(float) * c->nnz, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return bail(ctx->fail(SCLENS_ERR_HIP, std::string("counts_upload: ") + hipGetErrorString(e)));
  *out = c;
  return SCLENS_OK;
}
int counts_download(Ctx* ctx, const Counts* c, int64_t* colptr, int32_t* rowval, float* nzval) {
  hipStream_t st = ctx->stream;
  if (colptr) SCL_HIP(ctx, hipMemcpyAsync(colptr, c->colptr, sizeof(int64_t) * (c->M + 1), hipMemcpyDeviceToHost, st));
  if (rowval && c->nnz > 0) SCL_HIP(ctx, hipMemcpyAsync(rowval, c->row, sizeof(int32_t) * c->nnz, hipMemcpyDeviceToHost, st));
  if (nzval && c->nnz > 0) SCL_HIP(ctx, hipMemcpyAsync(nzval, c->val, sizeof(float) * c->nnz, hipMemcpyDeviceToHost, st));
  SCL_HIP(ctx, hipStreamSynchronize(st));
  return SCLENS_OK;
}
void counts_free(Counts* c, Ctx* busy) {
  if (!c) return;
  ctx_quiesce(busy);              // error paths: an upload / gather may still be queued on that context
  pool_free(c->colptr, nullptr);  // contract otherwise: no session / pattern build reads it any more (blocking API)
  pool_free(c->row, nullptr);
  pool_free(c->val, nullptr);
  delete c;
}

int pattern_create(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                   int64_t ncand, const uint32_t* z1, const uint32_t* z2, PatternOwner** out) {
  PatternOwner* p = new PatternOwner();
  const int rc = pattern_build(ctx, N, M, colptr, rowval, nzval, ncand, z1, z2, p);
  if (rc != SCLENS_OK) {
    pattern_free(p, ctx);
    delete p;
    return rc;
  }
  *out = p;
  return SCLENS_OK;
}
// counts' CSC in, candidates drawn on the device (R1) and merged into the union pattern there
int pattern_create_drawn(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, uint64_t seed,
                         PatternOwner** out, int64_t* ncand) {
  PatternOwner* p = new PatternOwner();
  const int rc = pattern_build_device(ctx, N, M, colptr, rowval, nzval, 0, nullptr, nullptr, 1, seed, p);
  if (rc != SCLENS_OK) {
    pattern_free(p, ctx);
    delete p;
    return rc;
  }
  if (ncand) *ncand = p->dev.ncand;
  *out = p;
  return SCLENS_OK;
}
int pattern_candidates(Ctx* ctx, PatternOwner* p, uint32_t* z1, uint32_t* z2) {
  if (!p || (p->dev.ncand > 0 && (!p->z1_dev || !p->z2_dev)))
    return ctx->fail(SCLENS_ERR_STATE, "pattern_candidates: this pattern does not hold its candidate list on the device");
  if (p->dev.ncand > 0) {
    SCL_HIP(ctx, hipMemcpyAsync(z1, p->z1_dev, sizeof(uint32_t) * p->dev.ncand, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipMemcpyAsync(z2, p->z2_dev, sizeof(uint32_t) * p->dev.ncand, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SCLENS_OK;
}
// the device arrays of a pattern, for tests (host copies): which = 0 colptr[M+1] i64, 1 row[nU] i32, 2 rowptr[N+1] i64,
// 3 csr2csc[nU] i64, 4 csrcol[nU] i32, 5 cand_pos[ncand] i64, 6 base_val[nU] f32
int pattern_download(Ctx* ctx, PatternOwner* p, int which, void* dst) {
  const PatternDev& d = p->dev;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (which) {
    case 0: src = d.colptr; bytes = sizeof(int64_t) * (d.M + 1); break;
    case 1: src = d.row; bytes = sizeof(int32_t) * d.nU; break;
    case 2: src = d.rowptr; bytes = sizeof(int64_t) * (d.N + 1); break;
    case 3: src = d.csr2csc; bytes = sizeof(int64_t) * d.nU; break;
    case 4: src = d.csrcol; bytes = sizeof(int32_t) * d.nU; break;
    case 5: src = d.cand_pos; bytes = sizeof(int64_t) * d.ncand; break;
    case 6: src = p->base_val; bytes = sizeof(float) * d.nU; break;
    default: return ctx->fail(SCLENS_ERR_ARG, "pattern_download: which must be 0..6");
  }
  if (bytes) {
    SCL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SCLENS_OK;
}
void pattern_destroy(PatternOwner* p) {
  if (!p) return;
  pattern_free(p);
  delete p;
}

// Row-sharded session: this process holds the cells [row0, row0 + N_local) of an N_global x M matrix,
// N_global > M (the Gram matrix is genes x genes, a sum over cells). colptr/rowval/nzval describe the LOCAL cells (row
// indices 0 .. N_local-1); z1 holds GLOBAL cell indices. Every session call must then be made by all ranks in the same
// order; results that live on the gene side (spectra, search statistics, scores, gene basis) come out identical on every
// rank, cell-side outputs (signal vectors, ensemble slots, TGC / norm_tgc) cover the local cells.
int session_create_sharded(Ctx* ctx, int64_t N_global, int64_t row0, int64_t N_local, int64_t M, const int64_t* colptr,
                           const int32_t* rowval, const float* nzval, int64_t ncand, const uint32_t* z1, const uint32_t* z2,
                           sclens_hip_allreduce_fn fn, void* user, Session** out) {
  if (!fn) return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded: an all-reduce function is required");
  if (N_global <= M) return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded: only the cells > genes layout shards by cells");
  if (N_local <= 0 || row0 < 0 || row0 + N_local > N_global) return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded: bad cell range");
  if (ctx->live_sessions > 0)
    return ctx->fail(SCLENS_ERR_STATE, "session_create_sharded: this context already has a live session");
  Session* s = new Session();
  s->ctx = ctx;
  s->N = N_local; s->M = M;
  s->n = M; s->K = N_local; s->Kdiv = N_global;
  s->cells_major = 0;
  s->sh.N_global = N_global; s->sh.row0 = row0; s->sh.fn = fn; s->sh.user = user;
  int rc = pattern_build(ctx, N_local, M, colptr, rowval, nzval, ncand, z1, z2, &s->pat, row0, N_global);
  if (rc != SCLENS_OK) { delete s; return rc; }
  s->ldb = round_up(s->K, 32);
  s->lda = round_up(s->n, 32);
  s->ldz = round_up(s->n, 32);
  s->ldn = round_up(N_local, 32);
  auto fail = [&](int code) { pattern_free(&s->pat, s->ctx); for (void* p : s->allocs) pool_free(p, nullptr); delete s; return code; };
  if ((rc = s->dmalloc((void**)&s->val, sizeof(float) * s->pat.dev.val_floats())) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Bmain, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Btmp, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->A, sizeof(float) * (size_t)s->n * s->lda)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->w64, sizeof(double) * s->n)) != SCLENS_OK) return fail(rc);
  s->ctx->live_sessions += 1;
  *out = s;
  return SCLENS_OK;
}
int session_set_reducer(Session* s, sclens_hip_allreduce_fn fn, void* user) {
  if (!s->sh.on() || !fn) return s->ctx->fail(SCLENS_ERR_STATE, "set_reducer: not a row-sharded session");
  s->sh.fn = fn;
  s->sh.user = user;
  s->sh.inherited = false;  // a worker clone now has a channel of its own
  return SCLENS_OK;
}
int session_set_reduce_to(Session* s, sclens_hip_reduce_fn fn, void* user) {
  if (!s->sh.on()) return s->ctx->fail(SCLENS_ERR_STATE, "set_reduce_to: not a row-sharded session");
  s->sh.rfn = fn;
  s->sh.ruser = user;
  return SCLENS_OK;
}
// row-sharded session with LOCAL candidates drawn on the device (this rank's part of the global draw sequence)
int session_create_sharded_drawn(Ctx* ctx, int64_t N_global, int64_t row0, int64_t N_local, int64_t M, const int64_t* colptr,
                                 const int32_t* rowval, const float* nzval, int64_t nnz_global, uint64_t seed,
                                 sclens_hip_allreduce_fn fn, void* user, Session** out, int64_t* ncand_local) {
  if (!fn) return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded_drawn: an all-reduce function is required");
  if (N_global <= M) return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded_drawn: only the cells > genes layout shards by cells");
  if (N_local <= 0 || row0 < 0 || row0 + N_local > N_global || !colptr || nnz_global < colptr[M])
    return ctx->fail(SCLENS_ERR_ARG, "session_create_sharded_drawn: bad cell range / entry count");
  if (ctx->live_sessions > 0)
    return ctx->fail(SCLENS_ERR_STATE, "session_create_sharded_drawn: this context already has a live session");
  Session* s = new Session();
  s->ctx = ctx;
  s->N = N_local; s->M = M;
  s->n = M; s->K = N_local; s->Kdiv = N_global;
  s->cells_major = 0;
  s->sh.N_global = N_global; s->sh.row0 = row0; s->sh.fn = fn; s->sh.user = user;
  const BlockDraw blk{N_global, row0, nnz_global};
  int rc = pattern_build_device(ctx, N_local, M, colptr, rowval, nzval, 0, nullptr, nullptr, 1, seed, &s->pat, -1, &blk);
  if (rc != SCLENS_OK) { pattern_free(&s->pat, ctx); delete s; return rc; }
  // until the host has gathered the ranks' counts the window is unknown: samples are refused (population 0 < m)
  s->pat.dev.cand_off = 0;
  s->pat.dev.ncand_global = s->pat.dev.ncand;
  s->ldb = round_up(s->K, 32);
  s->lda = round_up(s->n, 32);
  s->ldz = round_up(s->n, 32);
  s->ldn = round_up(N_local, 32);
  auto fail = [&](int code) { pattern_free(&s->pat, s->ctx); for (void* p : s->allocs) pool_free(p, nullptr); delete s; return code; };
  if ((rc = s->dmalloc((void**)&s->val, sizeof(float) * s->pat.dev.val_floats())) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Bmain, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Btmp, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->A, sizeof(float) * (size_t)s->n * s->lda)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->w64, sizeof(double) * s->n)) != SCLENS_OK) return fail(rc);
  s->ctx->live_sessions += 1;
  if (ncand_local) *ncand_local = s->pat.dev.ncand;
  *out = s;
  return SCLENS_OK;
}
int session_set_candidate_range(Session* s, int64_t cand_off, int64_t ncand_global) {
  if (!s->sh.on()) return s->ctx->fail(SCLENS_ERR_STATE, "set_candidate_range: not a row-sharded session");
  if (cand_off < 0 || cand_off + s->pat.dev.ncand > ncand_global || ncand_global >= 0xFFFFFFF0ll)
    return s->ctx->fail(SCLENS_ERR_ARG, "set_candidate_range: the window does not fit the global list");
  s->pat.dev.cand_off = cand_off;
  s->pat.dev.ncand_global = ncand_global;
  return SCLENS_OK;
}
__global__ void k_add_u32(const uint32_t* __restrict__ in, int64_t n, uint32_t add, uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + add;
}
int session_local_candidates(Session* s, uint32_t* z1, uint32_t* z2) {
  Ctx* ctx = s->ctx;
  const int64_t nc = s->pat.dev.ncand;
  if (nc > 0 && (!s->pat.z1_dev || !s->pat.z2_dev))
    return ctx->fail(SCLENS_ERR_STATE, "local_candidates: this session does not hold its candidate list on the device");
  if (nc == 0) return SCLENS_OK;
  SCL_WS(ctx, tmp, uint32_t, "ses.z1g", nc);
  hipLaunchKernelGGL(k_add_u32, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, s->pat.z1_dev, nc, (uint32_t)s->sh.row0, tmp);
  SCL_HIP(ctx, hipMemcpyAsync(z1, tmp, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipMemcpyAsync(z2, s->pat.z2_dev, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

// A second session on another context (= another stream of the same GPU) that shares the read-only device data of
// `src` (sparse pattern, Vr2, CheFSI seed block) and owns its scratch: independent search iterations / ensemble members
// can then run concurrently, the latency-bound column kernels of one decomposition overlapping the bandwidth-bound
// kernel of the other. The clone must be destroyed before `src`.
int session_clone(Ctx* ctx2, Session* src, Session** out) {
  if (ctx2->live_sessions > 0 || ctx2 == src->ctx)
    return ctx2->fail(SCLENS_ERR_STATE, "session_clone: the worker needs a context of its own without a live session");
  if (src->chunked()) return ctx2->fail(SCLENS_ERR_STATE, "session_clone: a chunked session has no worker sessions");
  Session* s = new Session();
  s->ctx = ctx2;
  ctx2->opt = src->ctx->opt;  // the worker decomposes the way its parent does
  s->N = src->N; s->M = src->M; s->n = src->n; s->K = src->K;
  s->Kdiv = src->Kdiv;
  s->sh = src->sh;  // same cells; the worker gets its own reducer channel through session_set_reducer, and until then has none
  if (s->sh.on()) {
    s->sh.inherited = true;
    s->sh.rfn = nullptr;
    s->sh.ruser = nullptr;
  }
  s->cells_major = src->cells_major;
  s->centering = src->centering;
  s->pat.dev = src->pat.dev;            // shared, not owned (allocs stays empty)
  s->pat.base_val = src->pat.base_val;
  s->ldb = src->ldb; s->lda = src->lda; s->ldz = src->ldz; s->ldn = src->ldn;
  s->Vr2t = src->Vr2t; s->r_vr2 = src->r_vr2;
  s->Z0t = src->Z0t; s->theta0 = src->theta0; s->b0 = src->b0; s->use_chefsi = src->use_chefsi;
  s->k = src->k;
  int rc;
  auto fail = [&](int code) { ctx_quiesce(s->ctx); for (void* p : s->allocs) pool_free(p, nullptr); delete s; return code; };
  if ((rc = s->dmalloc((void**)&s->val, sizeof(float) * s->pat.dev.val_floats())) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->Btmp, sizeof(float) * (size_t)s->n * s->ldb)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->A, sizeof(float) * (size_t)s->n * s->lda)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->w64, sizeof(double) * s->n)) != SCLENS_OK) return fail(rc);
  s->ctx->live_sessions += 1;
  *out = s;
  return SCLENS_OK;
}

static void chunk_cache_flush(Session* s);
static std::vector<ChunkSrc>& chunk_set(Session* s, int set);
static bool gram_sparse_pays(Ctx* ctx, const PatternDev& p, int64_t n, int64_t K);
// Chunked session: the shell (gene-side buffers), then the chunks one at a time (the host never holds more than one), then commit.
int session_create_chunked(Ctx* ctx, int64_t N_global, int64_t M, int n_chunks, int64_t nnz_global, uint64_t seed, Session** out) {
  if (N_global <= M || M <= 0 || n_chunks <= 0 || n_chunks > 4096 || nnz_global < 0 || nnz_global >= 0xFFFFFFF0ll)
    return ctx->fail(SCLENS_ERR_ARG, "session_create_chunked: needs cells > genes, 1..4096 chunks, fewer than 2^32 stored entries");
  if (ctx->live_sessions > 0) return ctx->fail(SCLENS_ERR_STATE, "session_create_chunked: this context already has a live session");
  Session* s = new Session();
  s->ctx = ctx;
  s->N = N_global; s->M = M;
  s->n = M; s->K = N_global; s->Kdiv = N_global;
  s->cells_major = 0;
  s->chunks.resize((size_t)n_chunks);
  s->nnz_global = nnz_global;
  s->cand_seed = seed;
  s->lda = round_up(M, 32);
  s->ldz = round_up(M, 32);
  s->ldn = round_up(N_global, 32);
  int rc;
  auto fail = [&](int code) { for (void* p : s->allocs) pool_free(p, nullptr); delete s; return code; };
  if ((rc = s->dmalloc((void**)&s->A, sizeof(float) * (size_t)s->n * s->lda)) != SCLENS_OK) return fail(rc);
  if ((rc = s->dmalloc((void**)&s->w64, sizeof(double) * s->n)) != SCLENS_OK) return fail(rc);
  for (ChunkStatsDev* st : {&s->st_data, &s->st_last}) {
    double* blk = nullptr;
    if ((rc = s->dmalloc((void**)&blk, sizeof(double) * (size_t)(3 * M + 8))) != SCLENS_OK) return fail(rc);
    st->stdv = blk; st->mu = blk + M; st->cent = blk + 2 * M; st->red = blk + 3 * M;
  }
  s->ctx->live_sessions += 1;
  *out = s;
  return SCLENS_OK;
}
int session_chunk_add(Session* s, int which, int g, int64_t row0, int64_t N_local, const int64_t* colptr, const int32_t* rowval,
                      const float* nzval) {
  Ctx* ctx = s->ctx;
  if (!s->chunked()) return ctx->fail(SCLENS_ERR_STATE, "chunk_add: not a chunked session");
  if ((which != 0 && which != 1) || g < 0 || g >= (int)s->chunks.size() || N_local <= 0 || row0 < 0 || row0 + N_local > s->N || !colptr)
    return ctx->fail(SCLENS_ERR_ARG, "chunk_add: bad chunk index / cell range");
  if (which == 0 && s->chunk_committed) return ctx->fail(SCLENS_ERR_STATE, "chunk_add: the count matrix is already committed");
  if (which == 1) {
    if (!s->chunk_committed) return ctx->fail(SCLENS_ERR_STATE, "chunk_add: commit the count matrix before adding X_r");
    if (s->null_chunks.empty()) s->null_chunks.resize(s->chunks.size());
    if (row0 != s->chunks[g].row0 || N_local != s->chunks[g].N) return ctx->fail(SCLENS_ERR_ARG, "chunk_add: X_r must be cut like the count matrix");
  }
  ChunkSrc& c = chunk_set(s, which)[g];
  if (c.counts) return ctx->fail(SCLENS_ERR_STATE, "chunk_add: this chunk has been added already");
  c.row0 = row0; c.N = N_local;
  return counts_upload(ctx, N_local, s->M, colptr, rowval, nzval, &c.counts);
}
int session_chunk_commit(Session* s) {
  Ctx* ctx = s->ctx;
  if (!s->chunked() || s->chunk_committed) return ctx->fail(SCLENS_ERR_STATE, "chunk_commit: not a chunked session, or committed already");
  int64_t next = 0, nmax = 0, nnz = 0;
  for (const ChunkSrc& c : s->chunks) {  // consecutive blocks of cells covering [0, N)
    if (!c.counts || c.row0 != next) return ctx->fail(SCLENS_ERR_ARG, "chunk_commit: the chunks must be consecutive blocks of cells, all added");
    next += c.N;
    nmax = std::max(nmax, c.N);
    nnz += c.counts->nnz;
  }
  if (next != s->N) return ctx->fail(SCLENS_ERR_ARG, "chunk_commit: the chunks do not cover all cells");
  if (nnz != s->nnz_global) return ctx->fail(SCLENS_ERR_ARG, "chunk_commit: stored entries of the chunks != nnz_global");
  s->ldb = round_up(nmax, 32);
  SCL_TRY(s->dmalloc((void**)&s->Btmp, sizeof(float) * (size_t)s->n * s->ldb));
  s->chunk_committed = true;
  return SCLENS_OK;
}
void session_destroy(Session* s) {
  if (!s) return;
  hipStreamSynchronize(s->ctx->stream);
  s->ctx->live_sessions -= 1;
  chunk_cache_flush(s);
  for (ChunkSrc& c : s->chunks) counts_free(c.counts);
  for (ChunkSrc& c : s->null_chunks) counts_free(c.counts);
  pattern_free(&s->pat);
  for (void* p : s->allocs) pool_free(p, nullptr);  // the stream has just been synchronised
  delete s;
}

// ------------------------------------------------------------------------------------------------ chunked session
// BASELINE configs[4] ("chunked Gram accumulation"): a cells > genes matrix whose scaled form does not fit the device -- 1 000 000 x
// 30 000 is 120 GB dense, 175 GB as a union pattern -- is held as device-resident CSC chunks of rows (8 bytes per stored entry) and
// every decomposition visits the chunks: pattern of the chunk (built on the device, with the chunk's part of the global candidate
// draw when the matrix carries candidate ones; kept between visits as far as chunk_cache_gb allows), value array, the chunk's terms
// of the statistics (scale.hip, chunked variant: three passes), its scaled block, and the block's contribution to the M x M Gram
// matrix (scLENS.jl:332-361 as a sum over cell blocks). Everything on the gene side -- eigensolver, search statistic, partial
// eigensolver, robustness -- is the code of the plain session; the cell-side vectors of ALL cells stay resident (k x N floats).
// One GPU does then what SURVEY 8e-iii spreads over ranks; the row-sharded session remains the multi-GPU form.
static void chunk_cache_flush(Session* s) {
  if (!s->pcache.empty()) ctx_quiesce(s->ctx);
  for (ChunkPat* e : s->pcache) {
    pattern_free(&e->pat);
    delete e;
  }
  s->pcache.clear();
  s->pcache_bytes = 0;
}
static std::vector<ChunkSrc>& chunk_set(Session* s, int set) { return set == 0 ? s->chunks : s->null_chunks; }
struct ChunkPatRef {
  ChunkPat* e = nullptr;
  bool transient = false;
};
static int chunk_count_candidates(Session* s);
static int chunk_pattern(Session* s, int set, int cands, int g, ChunkPatRef* ref) {
  Ctx* ctx = s->ctx;
  if (cands && !s->cands_counted) SCL_TRY(chunk_count_candidates(s));
  if (!s->pcache.empty() && (s->pcache[0]->set != set || s->pcache[0]->cands != cands)) chunk_cache_flush(s);  // a phase is over
  s->chunk_visits += 1;
  for (ChunkPat* e : s->pcache)
    if (e->g == g) {
      ref->e = e;
      ref->transient = false;
      return SCLENS_OK;
    }
  const ChunkSrc& c = chunk_set(s, set)[g];
  if (!c.counts) return ctx->fail(SCLENS_ERR_STATE, "chunked session: chunk " + std::to_string(g) + " has not been added");
  size_t live0 = 0, live1 = 0;
  pool_stats(ctx->device, nullptr, &live0, nullptr, nullptr);
  ChunkPat* e = new ChunkPat();
  e->set = set; e->cands = cands; e->g = g;
  const BlockDraw blk{s->N, c.row0, s->nnz_global};
  const int rc = pattern_build_device(ctx, c.N, s->M, c.counts->colptr, c.counts->row, c.counts->val, 0, nullptr, nullptr, cands ? 1 : 0,
                                      s->cand_seed, &e->pat, c.counts->nnz, cands ? &blk : nullptr);
  if (rc != SCLENS_OK) {
    pattern_free(&e->pat, ctx);
    delete e;
    return rc;
  }
  s->chunk_builds += 1;
  if (cands) {
    e->pat.dev.cand_off = c.cand_off;
    e->pat.dev.ncand_global = s->ncand_total;
  }
  pool_stats(ctx->device, nullptr, &live1, nullptr, nullptr);
  e->bytes = live1 > live0 ? live1 - live0 : 0;
  size_t budget = (size_t)std::max<int64_t>(0, ctx->opt.chunk_cache_gb) << 30;
  if (ctx->opt.chunk_cache_gb < 0) {  // auto: what the device has beyond the working set of a visit + the gene side (measured at 1M x 30k:
    size_t fr = 0, tot = 0;           // 273 GB live at the peak with 126 GB of patterns, 275 / 247 GB with 105 GB at precision 1 / 0,
                                      // 196 GB with 63 GB): the device (288 GiB) less 200 GiB = four union patterns of 21 GB
    budget = (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > ((size_t)200 << 30)) ? tot - ((size_t)200 << 30) : 0;
  }
  if (s->pcache_bytes + e->bytes <= budget) {
    s->pcache.push_back(e);
    s->pcache_bytes += e->bytes;
    ref->transient = false;
  } else {
    ref->transient = true;
  }
  ref->e = e;
  return SCLENS_OK;
}
static void chunk_pattern_done(Session* s, ChunkPatRef* ref) {
  if (ref->e && ref->transient) {
    ctx_quiesce(s->ctx);
    pattern_free(&ref->e->pat);
    delete ref->e;
  }
  ref->e = nullptr;
}
// the windows of the chunks in the global candidate list (R1, scLENS.jl:668-673): the list is the concatenation of the chunks' parts of
// the one global draw sequence, in chunk order -- the convention of the row-sharded session with local candidates
static int chunk_count_candidates(Session* s) {
  Ctx* ctx = s->ctx;
  if (!s->chunk_committed) return ctx->fail(SCLENS_ERR_STATE, "chunked session: call chunk_commit first");
  s->cands_counted = true;  // (chunk_pattern below must not recurse)
  s->ncand_total = 0;
  chunk_cache_flush(s);
  int64_t off = 0;
  for (size_t g = 0; g < s->chunks.size(); ++g) {
    ChunkPatRef ref;
    const int rc = chunk_pattern(s, 0, 1, (int)g, &ref);
    if (rc != SCLENS_OK) { s->cands_counted = false; return rc; }
    s->chunks[g].ncand = ref.e->pat.dev.ncand;
    s->chunks[g].cand_off = off;
    off += s->chunks[g].ncand;
    chunk_pattern_done(s, &ref);
  }
  if (off >= 0xFFFFFFF0ll) { s->cands_counted = false; return ctx->fail(SCLENS_ERR_ARG, "chunked session: more than 2^32 zero candidates"); }
  s->ncand_total = off;
  for (ChunkPat* e : s->pcache) {  // the patterns that stayed in the cache were built before the windows were known
    e->pat.dev.cand_off = s->chunks[e->g].cand_off;
    e->pat.dev.ncand_global = off;
  }
  return SCLENS_OK;
}
static int chunk_values(Session* s, const MatSpec& ms, const PatternOwner& p, float** out) {
  Ctx* ctx = s->ctx;
  float* val = static_cast<float*>(ctx->workspace("ses.cval", sizeof(float) * p.dev.val_floats()));
  if (!val) return SCLENS_ERR_OOM;
  if (ms.sample != 0 && !ms.cands) return ctx->fail(SCLENS_ERR_STATE, "chunked session: a sample needs patterns with candidates");
  if (ms.sample == 2) SCL_TRY(make_values_seeded(ctx, p.dev, p.base_val, ms.binary, ms.seed, ms.m, val));
  else SCL_TRY(make_values(ctx, p.dev, p.base_val, ms.binary, ms.sample == 1 ? s->idx_dev : nullptr, ms.sample == 1 ? ms.m : 0, val));
  *out = val;
  return SCLENS_OK;
}
// Gram matrix of the scaled matrix `ms` (all cells) / divisor -> A, its statistics -> st (and rec_vals -> keep)
static int chunked_gram(Session* s, const MatSpec& ms, float divisor, ScaleVecs* keep, float* A, ChunkStatsDev* st) {
  Ctx* ctx = s->ctx;
  const int64_t M = s->M;
  const double ng = (double)s->N;
  std::vector<ChunkSrc>& set = chunk_set(s, ms.set);
  if (!s->chunk_committed || set.empty()) return ctx->fail(SCLENS_ERR_STATE, "chunked session: no chunks (chunk_add / chunk_commit first)");
  SCL_WS(ctx, acc, double, "ck.acc", 2 * M);
  SCL_WS(ctx, acc2, double, "ck.acc2", M);
  SCL_WS(ctx, accT, double, "ck.accT", M + 2);
  SCL_WS(ctx, mean, double, "ck.mean", M);
  SCL_WS(ctx, zero, double, "ck.zero", M);
  hipStream_t stq = ctx->stream;
  SCL_HIP(ctx, hipMemsetAsync(acc, 0, sizeof(double) * 2 * M, stq));
  SCL_HIP(ctx, hipMemsetAsync(acc2, 0, sizeof(double) * M, stq));
  SCL_HIP(ctx, hipMemsetAsync(accT, 0, sizeof(double) * (M + 2), stq));
  SCL_HIP(ctx, hipMemsetAsync(zero, 0, sizeof(double) * M, stq));
  st->valid = false;
  for (int pass = 0; pass < 3; ++pass) {
    for (size_t g = 0; g < set.size(); ++g) {
      ChunkPatRef ref;
      SCL_TRY(chunk_pattern(s, ms.set, ms.cands, (int)g, &ref));
      float* val = nullptr;
      int rc = chunk_values(s, ms, ref.e->pat, &val);
      const PatternDev& p = ref.e->pat.dev;
      if (rc == SCLENS_OK && pass == 0) rc = chunk_pass_sum(ctx, p, val, ms.f32path, acc);
      if (rc == SCLENS_OK && pass == 1) rc = chunk_pass_var(ctx, p, val, ms.f32path, acc, ng, acc2);
      if (rc == SCLENS_OK && pass == 2) {
        double *tgc = nullptr, *l2 = nullptr, *lg = nullptr, *srow = nullptr;
        // a binarised matrix: the chunk's U'U as the exact co-occurrence product (gram_bits.hip), under the switch of the plain session
        const int gbc = ctx->opt.eff_gram_binary();
        const bool bits = ms.binary && (gbc == 1 || (gbc < 0 && s->n >= ctx->opt.gram_bits_min_n)) &&
                          gram_binary_scratch_bytes(p.N, M) <= sizeof(float) * (size_t)s->n * (size_t)s->ldb;
        const bool sparse = !bits && ctx->opt.gram_sparse != 0 && p.nU < 0x7FFFFFFFll &&  // no dense block at all
                            (ctx->opt.gram_sparse == 1 || (s->n >= ctx->opt.gram_sparse_min_n && gram_sparse_pays(ctx, p, s->n, p.N)));
        {
          StageTimer tm(ctx, "scale");
          rc = chunk_dense(ctx, p, val, ms.f32path, st->stdv, st->mu, st->red, 1.0, zero, accT, (sparse || bits) ? nullptr : s->Btmp, s->ldb, &tgc, &l2,
                           &lg, &srow);
        }
        if (rc == SCLENS_OK && keep) {
          hipError_t e1 = hipMemcpyAsync(keep->tgc + set[g].row0, tgc, sizeof(double) * set[g].N, hipMemcpyDeviceToHost, stq);
          hipError_t e2 = hipMemcpyAsync(keep->norm_tgc + set[g].row0, l2, sizeof(double) * set[g].N, hipMemcpyDeviceToHost, stq);
          if (e1 != hipSuccess || e2 != hipSuccess) rc = ctx->fail(SCLENS_ERR_HIP, "chunked session: rec_vals copy failed");
        }
        if (rc == SCLENS_OK && bits) {
          const ScaleStats cs{tgc, lg, mean, st->stdv, st->mu, l2, srow, zero, st->red};  // srow = 1 / l, cent = 0: the chunk's U'U
          rc = gram_binary_stats(ctx, p, val, ms.f32path, &cs, s->Btmp, divisor, A, s->lda, nullptr, g > 0);
          if (rc == SCLENS_OK) ctx->gram_bits_used += 1;
        } else if (rc == SCLENS_OK && sparse) {  // the chunk's U'U / divisor from its sparse structure (c = 1, no cent term: applied at the end)
          rc = gram_sparse(ctx, p, val, ms.f32path, tgc, lg, st->stdv, st->mu, l2, nullptr, nullptr, 1.0, 1.0 / (double)divisor, 0.0, A, s->lda, g > 0);
          if (rc == SCLENS_OK) ctx->gram_sparse_used += 1;
        } else if (rc == SCLENS_OK) {
          // every chunk's product is formed on its own and added to the sum in ONE fp32 addition per entry: started from the running
          // sum (accumulators loaded from A), the small terms of a chunk would be added to an accumulator that already holds the
          // contributions of all earlier chunks and drop out of its fp32 mantissa (the bias of a long fp32 chain)
          float* target = A;
          if (g > 0) {
            target = static_cast<float*>(ctx->workspace("ck.Atmp", sizeof(float) * (size_t)s->n * s->lda));
            if (!target) rc = SCLENS_ERR_OOM;
          }
          if (rc == SCLENS_OK) rc = gram_f32(ctx, s->Btmp, M, set[g].N, s->ldb, divisor, target, s->lda, true, false);
          if (rc == SCLENS_OK && g > 0) {
            const int64_t cnt = s->n * s->lda;
            hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, ctx->stream, target, cnt, A);
            if (hipGetLastError() != hipSuccess) rc = ctx->fail(SCLENS_ERR_HIP, "chunked session: k_add_f32 launch failed");
          }
        }
      }
      chunk_pattern_done(s, &ref);
      SCL_TRY(rc);
    }
    if (pass == 1) SCL_TRY(chunk_stats_finish(ctx, M, acc, acc2, ng, ms.f32path, mean, st->stdv, st->mu, st->red));
  }
  SCL_TRY(chunk_gram_finish(ctx, A, M, s->lda, accT, st->stdv, st->mu, ng, (double)divisor, st->cent));
  double lsum = 0.0;
  SCL_HIP(ctx, hipMemcpyAsync(&lsum, accT + M + 1, sizeof(double), hipMemcpyDeviceToHost, stq));
  if (keep) {
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_mean, mean, sizeof(double) * M, hipMemcpyDeviceToHost, stq));
    SCL_HIP(ctx, hipMemcpyAsync(keep->mat2_std, st->stdv, sizeof(double) * M, hipMemcpyDeviceToHost, stq));
    SCL_HIP(ctx, hipMemcpyAsync(keep->cent, st->cent, sizeof(double) * M, hipMemcpyDeviceToHost, stq));
  }
  SCL_HIP(ctx, hipStreamSynchronize(stq));
  st->c = lsum / ng;
  st->spec = ms;
  st->valid = true;
  return SCLENS_OK;
}
// the scaled block of chunk g of the matrix st describes -> s->Btmp ([M][ldb], columns = the chunk's cells)
static int chunk_block(Session* s, const ChunkStatsDev& st, int g) {
  Ctx* ctx = s->ctx;
  if (!st.valid) return ctx->fail(SCLENS_ERR_STATE, "chunked session: no statistics to rebuild the scaled block from");
  ChunkPatRef ref;
  SCL_TRY(chunk_pattern(s, st.spec.set, st.spec.cands, g, &ref));
  float* val = nullptr;
  int rc = chunk_values(s, st.spec, ref.e->pat, &val);
  if (rc == SCLENS_OK) {
    StageTimer tm(ctx, "scale");
    rc = chunk_dense(ctx, ref.e->pat.dev, val, st.spec.f32path, st.stdv, st.mu, st.red, st.c, st.cent, nullptr, s->Btmp, s->ldb, nullptr, nullptr);
  }
  chunk_pattern_done(s, &ref);
  return rc;
}

// scaled dense matrix of `val` -> B, Gram -> A, eigenvalues -> w64/w_host
// binary: every value of `val` is 0 or 1 (sparsity search): large problems in the genes-major layout then skip the scaled
// matrix and form the Gram matrix on the fp16 MFMA (gram_bits.hip); B is only scratch in that case
// root's buffer to every rank of a row-sharded session: the others contribute zeros to a sum (no broadcast primitive in the reducer
// interface; the same device as perturb_round's share phase). A root whose eigensolver failed still enters the sum, with NaNs, so that
// every rank sees the failure instead of waiting for ever.
static int share_from_root(Session* s, int root, void* dev, int64_t count, int dtype, int root_rc) {
  Ctx* ctx = s->ctx;
  if (root < 0 || !s->sh.on()) return root_rc;
  const size_t bytes = (size_t)count * (dtype == 0 ? 8 : 4);
  if (!s->solves(root)) SCL_HIP(ctx, hipMemsetAsync(dev, 0, bytes, ctx->stream));
  else if (root_rc != SCLENS_OK) (void)hipMemsetAsync(dev, 0xFF, bytes, ctx->stream);  // all-ones words are NaNs in both widths
  const std::string keep = ctx->err;
  SCL_TRY(s->sh.sum(ctx, dev, count, dtype));
  if (root_rc != SCLENS_OK) ctx->err = keep;
  return root_rc;
}
static int eig_values_on_root(Session* s, int root, int64_t n_low) {
  Ctx* ctx = s->ctx;
  int rc = SCLENS_OK;
  if (s->solves(root)) rc = eig_values(ctx, s->A, s->n, s->lda, s->w64, n_low);
  SCL_TRY(share_from_root(s, root, s->w64, s->n, 0, rc));
  return s->fetch_w(n_low >= 0);  // a NaN from a failed root ends the call on every rank
}
static int eig_vectors_on_root(Session* s, int root, int64_t lo, int64_t hi) {  // -> s->Zt rows 0 .. hi - lo - 1 on every rank
  Ctx* ctx = s->ctx;
  int rc = SCLENS_OK;
  if (s->solves(root)) rc = eig_vectors(ctx, s->A, s->n, s->lda, s->w64, lo, hi, s->Zt, s->ldz);
  return share_from_root(s, root, s->Zt, (hi - lo) * s->ldz, 1, rc);
}

static bool use_gram_bits(const Session* s) {
  const int gb = s->ctx->opt.eff_gram_binary();
  if (s->centering || s->cells_major || gb == 0 || s->chunked()) return false;  // row-sharded sessions: each rank's additive part
  return gb == 1 || s->n >= s->ctx->opt.gram_bits_min_n;
}
// the search statistic from split fp16 images (22-bit operands, fp32 accumulation) under the same switch
static bool use_f16_corr(const Session* s) {
  const int gb = s->ctx->opt.eff_gram_bits();
  return gb == 1 || (gb < 0 && s->n >= s->ctx->opt.gram_bits_min_n);
}
// measured at 100 000 x 30 000 (profiles/r06_gram_sparse_ab_cfg4.log): 5.1e11 multiply-adds in 1.19 s against 0.68 s (fp32 MFMA) / 0.24 s
// (split fp16) for the 9.0e13 flop of the dense lower half
static bool gram_sparse_pays(Ctx* ctx, const PatternDev& p, int64_t n, int64_t K) {
  double macs = 0.0;
  if (gram_sparse_macs(ctx, p, &macs) != SCLENS_OK) return false;
  const double t_sparse = macs / 4.3e11;
  const double t_dense = (double)n * (double)(n + 1) * (double)K / (ctx->opt.split() && n >= ctx->opt.eff_gram_split_min() ? 3.66e14 : 1.31e14);
  return t_sparse < 0.9 * t_dense;
}
static bool use_gram_sparse(const Session* s, const PatternDev& p) {
  const int64_t g = s->ctx->opt.gram_sparse;
  if (g == 0 || s->centering || s->cells_major || s->sh.on() || s->chunked() || p.nU >= 0x7FFFFFFFll || !p.rowptr) return false;
  if (g == 1) return true;
  return s->n >= s->ctx->opt.gram_sparse_min_n && gram_sparse_pays(s->ctx, p, s->n, p.N);
}
// sum_root >= 0 (row-sharded session): the Gram matrix is summed onto that rank only and formed in `Aout` (default s->A);
// solve = false: stop after the Gram matrix
static int decompose(Session* s, const PatternDev& p, const float* val, int f32path, float* B, float divisor,
                     ScaleVecs* keep, int64_t n_low = -1, bool binary = false, bool solve = true, int sum_root = -1,
                     float* Aout = nullptr) {
  float* Ag = Aout ? Aout : s->A;
  if (s->chunked()) {  // the matrix is s->cspec (p / val / B unused): statistics and Gram matrix summed over the chunks of cells
    SCL_TRY(chunked_gram(s, s->cspec, divisor, keep, Ag, keep ? &s->st_data : &s->st_last));
  } else if (binary && !keep && use_gram_bits(s)) {
    SCL_TRY(gram_binary(s->ctx, p, val, f32path, B, divisor, Ag, s->lda, s->sh.on() ? &s->sh : nullptr));
    s->ctx->gram_bits_used += 1;
    if (s->sh.on()) {
      if (sum_root >= 0) SCL_TRY(s->sh.sum_to(s->ctx, Ag, s->n * s->lda, 1, sum_root));
      else SCL_TRY(s->sh.sum(s->ctx, Ag, s->n * s->lda, 1));
    }
  } else if (s->sh.on()) {  // partial statistics and a partial Gram matrix over this rank's cells, summed over the ranks
    SCL_TRY(scale_to_dense_sharded(s->ctx, p, val, f32path, B, s->ldb, keep, s->sh));
    SCL_TRY(gram_f32(s->ctx, B, s->n, s->K, s->ldb, divisor, Ag, s->lda));
    if (sum_root >= 0) SCL_TRY(s->sh.sum_to(s->ctx, Ag, s->n * s->lda, 1, sum_root));
    else SCL_TRY(s->sh.sum(s->ctx, Ag, s->n * s->lda, 1));
  } else if (use_gram_sparse(s, p)) {
    // the Gram matrix from the sparse structure of the scaled matrix (sparse + rank two); the dense matrix is written only
    // where a later step reads it (the data matrix: recovery, guard band, gene basis)
    ScaleStats ss;
    SCL_TRY(scale_to_dense_stats(s->ctx, p, val, f32path, 0, 0, B == s->Bmain ? B : nullptr, s->ldb, keep, &ss));
    SCL_TRY(gram_sparse(s->ctx, p, val, f32path, ss.tgc, ss.lg, ss.stdv, ss.mu, ss.l2, ss.cent, ss.red + 1, (double)p.N, 1.0 / (double)divisor,
                        (double)p.N / (double)divisor, Ag, s->lda, false));
    s->ctx->gram_sparse_used += 1;
  } else {
    SCL_TRY(scale_to_dense(s->ctx, p, val, s->centering ? 1 : f32path, s->centering, s->cells_major, B, s->ldb,
                           s->centering ? nullptr : keep));
    SCL_TRY(gram_f32(s->ctx, B, s->n, s->K, s->ldb, divisor, Ag, s->lda));
  }
  if (!solve) return SCLENS_OK;
  return eig_values_on_root(s, s->sh.on() ? sum_root : -1, n_low);
}

// null matrix X_r (scLENS.jl:701, :704) of a chunked session: its chunks have been added with which = 1 and are released afterwards
int session_null_spectrum_chunked(Session* s, double* Lr) {
  Ctx* ctx = s->ctx;
  if (!s->chunked() || s->null_chunks.empty()) return ctx->fail(SCLENS_ERR_STATE, "null_spectrum_chunked: add the chunks of X_r first");
  for (const ChunkSrc& c : s->null_chunks)
    if (!c.counts) return ctx->fail(SCLENS_ERR_STATE, "null_spectrum_chunked: a chunk of X_r is missing");
  s->cspec = MatSpec{};
  s->cspec.set = 1;
  s->ctx->q2_prebuild = false;  // eigenvalues only
  PatternDev none;
  const int rc = decompose(s, none, nullptr, 1, s->Btmp, (float)s->M, nullptr);
  s->ctx->q2_prebuild = true;
  chunk_cache_flush(s);
  for (ChunkSrc& c : s->null_chunks) counts_free(c.counts, ctx);
  s->null_chunks.clear();
  SCL_TRY(rc);
  if (Lr) std::copy(s->w_host.begin(), s->w_host.end(), Lr);
  return SCLENS_OK;
}

// null matrix X_r (scLENS.jl:701, :704): closure path, eigenvalues only (:532, :572)
int session_null_spectrum(Session* s, const int64_t* rc_, const int32_t* rr_, const float* rv_, double* Lr) {
  Ctx* ctx = s->ctx;
  if (s->chunked()) return ctx->fail(SCLENS_ERR_STATE, "null_spectrum: a chunked session takes X_r as chunks (chunk_add which = 1, null_spectrum_chunked)");
  PatternOwner pr;
  SCL_TRY(pattern_build(ctx, s->N, s->M, rc_, rr_, rv_, 0, nullptr, nullptr, &pr));  // sharded: this rank's cells of X_r
  const float* valr = pr.base_val;  // (no copy: see session_data_spectrum)
  int rc = SCLENS_OK;
  if (pr.dev.base_val_csr) {
    float* vr = static_cast<float*>(ctx->workspace("ses.valr", sizeof(float) * pr.dev.val_floats()));
    rc = vr ? SCLENS_OK : SCLENS_ERR_OOM;
    if (rc == SCLENS_OK) rc = make_values(ctx, pr.dev, pr.base_val, 0, nullptr, 0, vr);
    valr = vr;
  }
  s->ctx->q2_prebuild = false;  // eigenvalues only
  if (rc == SCLENS_OK) rc = decompose(s, pr.dev, valr, 1, s->Btmp, (float)s->M, nullptr, -1, false, true, s->sh.on() ? s->solve_root : -1);
  s->ctx->q2_prebuild = true;
  hipStreamSynchronize(ctx->stream);
  pattern_free(&pr);
  SCL_TRY(rc);
  if (Lr) std::copy(s->w_host.begin(), s->w_host.end(), Lr);
  return SCLENS_OK;
}

// the same from a pattern that was built (and uploaded) beforehand, e.g. on a host thread while the session was created
int session_null_spectrum_pattern(Session* s, PatternOwner* pr, double* Lr) {
  Ctx* ctx = s->ctx;
  if (s->chunked()) return ctx->fail(SCLENS_ERR_STATE, "null_spectrum: a chunked session takes X_r as chunks (chunk_add which = 1, null_spectrum_chunked)");
  if (!pr || pr->allocs.empty() || pr->dev.N != s->N || pr->dev.M != s->M)
    return ctx->fail(SCLENS_ERR_ARG, "null_spectrum: the pattern is empty or has different dimensions");
  const float* valr = pr->base_val;  // (no copy: see session_data_spectrum)
  if (pr->dev.base_val_csr) {
    float* vr = static_cast<float*>(ctx->workspace("ses.valr", sizeof(float) * pr->dev.val_floats()));
    if (!vr) return SCLENS_ERR_OOM;
    SCL_TRY(make_values(ctx, pr->dev, pr->base_val, 0, nullptr, 0, vr));
    valr = vr;
  }
  s->ctx->q2_prebuild = false;  // eigenvalues only
  const int rc_null = decompose(s, pr->dev, valr, 1, s->Btmp, (float)s->M, nullptr, -1, false, true, s->sh.on() ? s->solve_root : -1);
  s->ctx->q2_prebuild = true;
  SCL_TRY(rc_null);
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (Lr) std::copy(s->w_host.begin(), s->w_host.end(), Lr);
  return SCLENS_OK;
}

// data matrix: inline Float64 path with rec_vals (scLENS.jl:676-696); divisor size(X,2) = M
int session_data_spectrum(Session* s, double* L, ScaleVecs* keep) {
  Ctx* ctx = s->ctx;
  if (!s->Bmain && !s->chunked()) return ctx->fail(SCLENS_ERR_STATE, "data_spectrum: not available on a worker session");
  if (s->centering && keep)
    return ctx->fail(SCLENS_ERR_ARG, "data_spectrum: centering=median has no rec_vals (scLENS.jl:697-698), pass NULL");
  if (s->chunked()) {
    // the statistics must land in st_data whether or not the caller wants rec_vals: a host-side sink when it does not
    static thread_local std::vector<double> sink_n, sink_m;
    ScaleVecs local{};
    if (!keep) {
      sink_n.resize((size_t)2 * s->N);
      sink_m.resize((size_t)3 * s->M);
      local = ScaleVecs{sink_n.data(), sink_m.data(), sink_m.data() + s->M, sink_n.data() + s->N, sink_m.data() + 2 * s->M};
      keep = &local;
    }
    s->cspec = MatSpec{};
    s->cspec.f32path = 0;
  }
  // the stored counts themselves are the value array of the data matrix: a pattern without CSR companion copies (counts-only patterns)
  // needs no copy of them (k_val_init moved 2.5 GB for nothing at 100 000 x 30 000)
  const float* vdata = s->val;
  if (!s->chunked()) {
    if (s->pat.dev.base_val_csr) SCL_TRY(make_values(ctx, s->pat.dev, s->pat.base_val, 0, nullptr, 0, s->val));
    else vdata = s->pat.base_val;
  }
  s->data_root = s->sh.on() ? s->solve_root : -1;
  SCL_TRY(decompose(s, s->pat.dev, vdata, 0, s->Bmain, (float)s->M, keep, -1, false, true, s->data_root));
  if (L) std::copy(s->w_host.begin(), s->w_host.end(), L);
  s->have_spectrum = true;
  return SCLENS_OK;
}

int session_spectrum(Session* s, const int64_t* rc_, const int32_t* rr_, const float* rv_, double* L, double* Lr,
                     ScaleVecs* keep) {
  if (rc_) SCL_TRY(session_null_spectrum(s, rc_, rr_, rv_, Lr));
  return session_data_spectrum(s, L, keep);
}

// copy the shared read-only results of `src` (Vr2 and/or the seed block of the partial eigensolver) into `dst`
// (re)allocate the working value array for the session's current pattern
static int session_realloc_val(Session* s) {
  if (s->val) {
    auto it = std::find(s->allocs.begin(), s->allocs.end(), (void*)s->val);
    if (it != s->allocs.end()) s->allocs.erase(it);
    pool_free(s->val, s->ctx->stream);
    s->val = nullptr;
  }
  return s->dmalloc((void**)&s->val, sizeof(float) * s->pat.dev.val_floats());
}

// Hand a pattern built by pattern_create (counts + zero candidates) to an idle owner session that was created without
// candidates. The session takes ownership; *p is left empty. Worker clones refresh their view with adopt(what = 4).
int session_set_pattern(Session* s, PatternOwner* p) {
  Ctx* ctx = s->ctx;
  if (!p || p->allocs.empty()) return ctx->fail(SCLENS_ERR_ARG, "set_pattern: empty pattern");
  if (p->dev.N != s->N || p->dev.M != s->M) return ctx->fail(SCLENS_ERR_ARG, "set_pattern: pattern has different dimensions");
  if (s->pat.allocs.empty()) return ctx->fail(SCLENS_ERR_STATE, "set_pattern: not available on a worker session (adopt it instead)");
  if (s->sh.on()) return ctx->fail(SCLENS_ERR_STATE, "set_pattern: a row-sharded session takes its candidates at creation");
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pattern_free(&s->pat);
  s->pat.dev = p->dev;
  s->pat.base_val = p->base_val;
  s->pat.allocs.swap(p->allocs);
  p->dev = PatternDev();
  p->base_val = nullptr;
  return session_realloc_val(s);
}

int session_adopt(Session* dst, Session* src, int what) {
  if (what & 4) {  // the (new) sparse pattern of src; dst must be a clone (it never owns a pattern)
    if (!dst->pat.allocs.empty()) return dst->ctx->fail(SCLENS_ERR_STATE, "adopt: only a worker session can adopt a pattern");
    SCL_HIP(dst->ctx, hipStreamSynchronize(dst->ctx->stream));
    dst->pat.dev = src->pat.dev;
    dst->pat.base_val = src->pat.base_val;
    SCL_TRY(session_realloc_val(dst));
  }
  if (what & 1) { dst->Vr2t = src->Vr2t; dst->r_vr2 = src->r_vr2; dst->Vr2h_of = nullptr; }
  if (what & 2) { dst->Z0t = src->Z0t; dst->theta0 = src->theta0; dst->b0 = src->b0; dst->k = src->k; }
  return SCLENS_OK;
}

// rows of Zt (ascending eigen-index, gene- or cell-side, `cnt` rows) -> cell-side unit vectors,
// descending, in dst[cnt][ldn]. For N > M: normalize(X * v) (scLENS.jl:503-508, :556-558; the
// Lambda^-1/2 factor is positive and drops out of the normalisation).
static int copy_rows_f32(Ctx* ctx, const float* src, int64_t rows, int64_t cols, int64_t lds, float* dst, int64_t ldd) {
  SCL_HIP(ctx, hipMemcpy2DAsync(dst, sizeof(float) * ldd, src, sizeof(float) * lds, sizeof(float) * cols, rows,
                                hipMemcpyDeviceToDevice, ctx->stream));
  return SCLENS_OK;
}
// chunked session: B is not resident -- the blocks of the matrix `st` describes are rebuilt one at a time, each yields its cells' part
static int to_cell_side_chunked(Session* s, const ChunkStatsDev& st, int64_t cnt, float* dst, bool desc_input, const float* src) {
  Ctx* ctx = s->ctx;
  StageTimer tm(ctx, "recover");
  float* tmp = static_cast<float*>(ctx->workspace("ses.rec", sizeof(float) * (size_t)cnt * s->ldn));
  float* tc = static_cast<float*>(ctx->workspace("ses.recc", sizeof(float) * (size_t)cnt * s->ldb));
  if (!tmp || !tc) return SCLENS_ERR_OOM;
  for (size_t g = 0; g < s->chunks.size(); ++g) {
    const ChunkSrc& c = s->chunks[g];
    SCL_TRY(chunk_block(s, st, (int)g));
    GemmArgs a{};
    a.P = src; a.Q = s->Btmp; a.C = tc;
    a.M = cnt; a.N = c.N; a.K = s->M;
    a.ldp = s->ldz; a.ldq = s->ldb; a.ldc = s->ldb;
    a.alpha = 1.f; a.beta = 0.f; a.q_kcontig = 0; a.lower = 0; a.colabsmax = nullptr;
    SCL_TRY(gemm_f32(ctx, a));
    SCL_HIP(ctx, hipMemcpy2DAsync(tmp + c.row0, sizeof(float) * s->ldn, tc, sizeof(float) * s->ldb, sizeof(float) * c.N, cnt,
                                  hipMemcpyDeviceToDevice, ctx->stream));
  }
  SCL_TRY(normalize_rows_f32(ctx, tmp, cnt, s->N, s->ldn));
  return desc_input ? copy_rows_f32(ctx, tmp, cnt, s->N, s->ldn, dst, s->ldn) : reverse_rows_f32(ctx, tmp, cnt, s->N, s->ldn, dst, s->ldn);
}
static int to_cell_side(Session* s, const float* B, int64_t cnt, float* dst, bool desc_input = false,
                        const float* src = nullptr) {
  Ctx* ctx = s->ctx;
  if (!src) src = s->Zt;
  if (s->chunked()) return to_cell_side_chunked(s, B == nullptr ? s->st_data : s->st_last, cnt, dst, desc_input, src);
  if (s->cells_major)
    return desc_input ? copy_rows_f32(ctx, src, cnt, s->N, s->ldz, dst, s->ldn)
                      : reverse_rows_f32(ctx, src, cnt, s->N, s->ldz, dst, s->ldn);
  StageTimer tm(ctx, "recover");
  float* tmp = static_cast<float*>(ctx->workspace("ses.rec", sizeof(float) * (size_t)cnt * s->ldn));
  if (!tmp) return SCLENS_ERR_OOM;
  GemmArgs g{};
  g.P = src; g.Q = B; g.C = tmp;
  g.M = cnt; g.N = s->N; g.K = s->M;
  g.ldp = s->ldz; g.ldq = s->ldb; g.ldc = s->ldn;
  g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
  SCL_TRY(gemm_f32(ctx, g));
  if (s->sh.on()) {  // the recovery GEMM is row-local; only the vector norms span all cells
    double* nrm = static_cast<double*>(ctx->workspace("ses.recn", sizeof(double) * (size_t)cnt));
    if (!nrm) return SCLENS_ERR_OOM;
    SCL_TRY(row_sqnorms_f32(ctx, tmp, cnt, s->N, s->ldn, nrm));
    SCL_TRY(s->sh.sum(ctx, nrm, cnt, 0));
    SCL_TRY(scale_rows_rsqrt_f32(ctx, tmp, cnt, s->N, s->ldn, nrm));
  } else {
    SCL_TRY(normalize_rows_f32(ctx, tmp, cnt, s->N, s->ldn));
  }
  return desc_input ? copy_rows_f32(ctx, tmp, cnt, s->N, s->ldn, dst, s->ldn)
                    : reverse_rows_f32(ctx, tmp, cnt, s->N, s->ldn, dst, s->ldn);
}

// Guard band of the signal threshold (hard part 2): `sum(L .> lambda_c)` (scLENS.jl:539, :580) is a hard
// cut, and an fp32 eigenvalue carries an error of about sqrt(n) eps32 lambda_max. For the eigenvalues idx_lo .. idx_hi-1
// (ascending index) of the data matrix this returns the fp64 Rayleigh quotients rho = ||B' z||^2 / (divisor z'z) of their fp32
// eigenvectors z against the resident scaled matrix B (products and sums in fp64): second-order accurate in the error of z,
// i.e. the eigenvalue of the fp32 data to ~1e-9 relative. Valid between data_spectrum and signal_vectors.
constexpr int RQ_V = 8;  // vectors per pass over B
__global__ __launch_bounds__(256) void k_rayleigh_part(const float* __restrict__ B, int64_t n, int64_t K, int64_t ldb,
                                                       const float* __restrict__ Zt, int64_t ldz, int cnt, double* __restrict__ part) {
  __shared__ float zs[RQ_V][256];
  __shared__ double red[4][RQ_V];
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double y[RQ_V];
#pragma unroll
  for (int q = 0; q < RQ_V; ++q) y[q] = 0.0;
  for (int64_t i0 = 0; i0 < n; i0 += 256) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RQ_V; ++q) zs[q][threadIdx.x] = (q < cnt && i0 + threadIdx.x < n) ? Zt[(int64_t)q * ldz + i0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int lim = (int)((n - i0 < 256) ? n - i0 : 256);
    if (k < K) {
      for (int i = 0; i < lim; ++i) {
        const double b = (double)B[(i0 + i) * ldb + k];
#pragma unroll
        for (int q = 0; q < RQ_V; ++q) y[q] += b * (double)zs[q][i];
      }
    }
  }
  // sum of y^2 over the block's columns, fixed order
#pragma unroll
  for (int q = 0; q < RQ_V; ++q) {
    double v = y[q] * y[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < RQ_V) part[(int64_t)blockIdx.x * RQ_V + threadIdx.x] =
      (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int session_refine_eigenvalues(Session* s, int64_t idx_lo, int64_t idx_hi, double* rho) {
  Ctx* ctx = s->ctx;
  if (!s->have_spectrum || (!s->Bmain && !s->chunked())) return ctx->fail(SCLENS_ERR_STATE, "refine_eigenvalues: call data_spectrum first");
  if (idx_lo < 0 || idx_hi > s->n || idx_lo > idx_hi || !rho) return ctx->fail(SCLENS_ERR_ARG, "refine_eigenvalues: bad index range");
  const int64_t cnt = idx_hi - idx_lo;
  if (cnt == 0) return SCLENS_OK;
  StageTimer tm(ctx, "refine");
  SCL_TRY(s->ensure_zt(cnt));
  SCL_TRY(eig_vectors_on_root(s, s->data_root, idx_lo, idx_hi));
  if (s->chunked()) {  // ||B' z||^2 is a sum over the cells: block by block (each rebuilt once, all vector groups against it)
    const int64_t nbc = (s->ldb + 255) / 256;
    SCL_WS(ctx, partc, double, "ses.rq", nbc * RQ_V);
    std::vector<double> hpc((size_t)nbc * RQ_V);
    std::fill(rho, rho + cnt, 0.0);
    for (size_t g = 0; g < s->chunks.size(); ++g) {
      SCL_TRY(chunk_block(s, s->st_data, (int)g));
      const int64_t Kc = s->chunks[g].N, nbg = (Kc + 255) / 256;
      for (int64_t q0 = 0; q0 < cnt; q0 += RQ_V) {
        const int c = (int)std::min<int64_t>(RQ_V, cnt - q0);
        hipLaunchKernelGGL(k_rayleigh_part, dim3((unsigned)nbg), dim3(256), 0, ctx->stream, s->Btmp, s->n, Kc, s->ldb, s->Zt + q0 * s->ldz, s->ldz,
                           c, partc);
        SCL_HIP(ctx, hipGetLastError());
        SCL_HIP(ctx, hipMemcpyAsync(hpc.data(), partc, sizeof(double) * nbg * RQ_V, hipMemcpyDeviceToHost, ctx->stream));
        SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < c; ++q)
          for (int64_t b = 0; b < nbg; ++b) rho[q0 + q] += hpc[(size_t)b * RQ_V + q];
      }
    }
  }
  const int64_t nb = s->chunked() ? 0 : (s->K + 255) / 256;
  SCL_WS(ctx, part, double, "ses.rq2", nb * RQ_V + 1);
  std::vector<double> hp((size_t)nb * RQ_V);
  for (int64_t q0 = 0; !s->chunked() && q0 < cnt; q0 += RQ_V) {
    const int c = (int)std::min<int64_t>(RQ_V, cnt - q0);
    hipLaunchKernelGGL(k_rayleigh_part, dim3((unsigned)nb), dim3(256), 0, ctx->stream, s->Bmain, s->n, s->K, s->ldb,
                       s->Zt + q0 * s->ldz, s->ldz, c, part);
    SCL_HIP(ctx, hipGetLastError());
    SCL_HIP(ctx, hipMemcpyAsync(hp.data(), part, sizeof(double) * nb * RQ_V, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < c; ++q) {
      double acc = 0.0;
      for (int64_t b = 0; b < nb; ++b) acc += hp[(size_t)b * RQ_V + q];
      rho[q0 + q] = acc;
    }
  }
  if (s->sh.on()) {  // row-sharded: this rank holds a block of the contraction (cells); sum the partial quotients
    SCL_WS(ctx, dr, double, "ses.rqd", cnt);
    SCL_HIP(ctx, hipMemcpyAsync(dr, rho, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
    SCL_TRY(s->sh.sum(ctx, dr, cnt, 0));
    SCL_HIP(ctx, hipMemcpyAsync(rho, dr, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  // z comes out of fp32 back-transformations: its norm is 1 only to ~1e-6 at n ~ 3 * 10^4, which is the size of the error the
  // band is there to resolve, so the quotient is taken against z'z (fp64 sum of squares of the eigen-side vector, replicated
  // on every rank of a row-sharded session)
  SCL_WS(ctx, zsq, double, "ses.rqz", cnt);
  SCL_TRY(row_sqnorms_f32(ctx, s->Zt, cnt, s->n, s->ldz, zsq));
  std::vector<double> hz((size_t)cnt);
  SCL_HIP(ctx, hipMemcpyAsync(hz.data(), zsq, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t q = 0; q < cnt; ++q) {
    if (!(hz[(size_t)q] > 0.0)) return ctx->fail(SCLENS_ERR_NAN, "refine_eigenvalues: zero or NaN eigenvector");
    rho[q] /= (double)s->M * hz[(size_t)q];  // the divisor of data_spectrum: size(X, 2) = M (Appendix A8)
  }
  return SCLENS_OK;
}

int session_signal_vectors(Session* s, int64_t k, float* nV) {
  Ctx* ctx = s->ctx;
  if (!s->have_spectrum) return ctx->fail(SCLENS_ERR_STATE, "signal_vectors: call spectrum first");
  if (k < 0 || k > s->n) return ctx->fail(SCLENS_ERR_ARG, "signal_vectors: bad k");
  s->k = k;
  if (k == 0) return SCLENS_OK;
  // besides the k signal vectors keep the leading b0 eigenvectors of the data matrix: they seed the subspace
  // iteration of the ensemble members (min_pc = ceil(1.5 k) wanted + a guard band)
  const int64_t min_pc = (3 * k + 1) / 2;
  int64_t b0 = round_up(min_pc + 40, 32);  // guard band: the block product streams the matrix once whatever b <= 128 is
  // (a 128-row block was measured at 50 000 x 30 000: 5 instead of 6.25 sweeps per member, but 248 instead of 194 ms -- the
  // b x b Rayleigh-Ritz problem on the host grows with b^3; context option chefsi_b0 overrides for experiments)
  if (ctx->opt.chefsi_b0 > 0) b0 = round_up(std::max<int64_t>(min_pc + 8, ctx->opt.chefsi_b0), 32);
  if (b0 > 128 || b0 > s->n / 2) b0 = 0;  // too wide for the small-block solver: ensemble uses the full solver
  const int64_t nv = std::max(k, b0);
  SCL_TRY(s->ensure_zt(nv));
  SCL_TRY(eig_vectors_on_root(s, s->data_root, s->n - nv, s->n));
  s->b0 = b0;
  if (b0 > 0) {
    float* z0 = static_cast<float*>(ctx->workspace("ses.Z0t", sizeof(float) * (size_t)b0 * s->ldz));
    if (!z0) return SCLENS_ERR_OOM;
    s->Z0t = z0;
    SCL_TRY(reverse_rows_f32(ctx, s->Zt + (nv - b0) * s->ldz, b0, s->n, s->ldz, z0, s->ldz));
    s->theta0.resize(b0);
    for (int64_t q = 0; q < b0; ++q) s->theta0[q] = s->w_host[s->n - 1 - q];
  }
  {
    float* p = static_cast<float*>(ctx->workspace("ses.nVt", sizeof(float) * (size_t)k * s->ldn));
    if (!p) return SCLENS_ERR_OOM;
    s->nVt = p;
  }
  SCL_TRY(to_cell_side(s, s->Bmain, k, s->nVt, false, s->Zt + (nv - k) * s->ldz));  // top-k rows (ascending)
  if (nV) {
    SCL_HIP(ctx, hipMemcpy2DAsync(nV, sizeof(float) * s->N, s->nVt, sizeof(float) * s->ldn, sizeof(float) * s->N, k,
                                  hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  s->have_spectrum = false;  // reflectors are overwritten by the next decomposition
  return SCLENS_OK;
}

int session_binary_basis(Session* s, double* L_bin, int64_t* r_out) {
  Ctx* ctx = s->ctx;
  if (s->chunked()) {
    s->cspec = MatSpec{};
    s->cspec.binary = 1;
  } else {
    SCL_TRY(make_values(ctx, s->pat.dev, s->pat.base_val, 1, nullptr, 0, s->val));
  }
  // get_eigvec(scaled', ...) for N > M / get_eigvec(scaled) otherwise: n x n Gram, divisor = K (Appendix A8)
  const int broot = s->sh.on() ? s->solve_root : -1;
  SCL_TRY(decompose(s, s->pat.dev, s->val, 1, s->Btmp, (float)s->Kdiv, nullptr, -1, /*binary=*/true, true, broot));
  if (L_bin) std::copy(s->w_host.begin(), s->w_host.end(), L_bin);
  const int64_t r = s->count_positive();
  s->r_vr2 = r;
  if (r_out) *r_out = r;
  if (r == 0) return SCLENS_OK;
  SCL_TRY(s->ensure_zt(r));
  SCL_TRY(eig_vectors_on_root(s, broot, s->n - r, s->n));
  float* v = static_cast<float*>(ctx->workspace("ses.Vr2t", sizeof(float) * (size_t)r * s->ldz));
  if (!v) return SCLENS_ERR_OOM;
  s->Vr2t = v;
  s->Vr2h_of = nullptr;
  SCL_HIP(ctx, hipMemcpyAsync(v, s->Zt, sizeof(float) * (size_t)r * s->ldz, hipMemcpyDeviceToDevice, ctx->stream));
  // worker sessions adopt this pointer and read it from their own streams: the copy must have landed on return
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

static int search_core(Session* s, int64_t n_2, double* d5, int64_t* r_it, bool gram_done = false);
int session_search_step(Session* s, const uint32_t* sample, int64_t m, int64_t n_2, double* d5, int64_t* r_it) {
  Ctx* ctx = s->ctx;
  if (!s->Vr2t) return ctx->fail(SCLENS_ERR_STATE, "search_step: call binary_basis first");
  if (s->chunked()) {
    if (!s->cands_counted) SCL_TRY(chunk_count_candidates(s));
    if (m < 0 || m > s->ncand_total) return ctx->fail(SCLENS_ERR_ARG, "search_step: bad sample size");
    SCL_TRY(s->upload_idx(sample, m));
    s->cspec = MatSpec{};
    s->cspec.cands = 1; s->cspec.binary = 1; s->cspec.sample = 1; s->cspec.m = m;
    return search_core(s, n_2, d5, r_it);
  }
  if (m < 0 || m > s->pat.dev.population()) return ctx->fail(SCLENS_ERR_ARG, "search_step: bad sample size");
  SCL_TRY(s->upload_idx(sample, m));
  SCL_TRY(make_values(ctx, s->pat.dev, s->pat.base_val, 1, s->idx_dev, m, s->val));
  return search_core(s, n_2, d5, r_it);
}
int session_search_step_seeded(Session* s, uint64_t seed, int64_t m, int64_t n_2, double* d5, int64_t* r_it) {
  Ctx* ctx = s->ctx;
  if (!s->Vr2t) return ctx->fail(SCLENS_ERR_STATE, "search_step: call binary_basis first");
  if (s->chunked()) {
    if (!s->cands_counted) SCL_TRY(chunk_count_candidates(s));
    if (m < 0 || m > s->ncand_total) return ctx->fail(SCLENS_ERR_ARG, "search_step: bad sample size");
    s->cspec = MatSpec{};
    s->cspec.cands = 1; s->cspec.binary = 1; s->cspec.sample = 2; s->cspec.seed = seed; s->cspec.m = m;
    return search_core(s, n_2, d5, r_it);
  }
  SCL_TRY(make_values_seeded(ctx, s->pat.dev, s->pat.base_val, 1, seed, m, s->val));
  return search_core(s, n_2, d5, r_it);
}
// all eigenvalues of the reduction that the last partial eig_values left on the context (fallback of search_core)
static int stebz_redo_all(Session* s) {
  SCL_TRY(eig_values_redo_all(s->ctx, s->n, s->w64));
  return s->fetch_w();
}
// gram_done: s->A already holds the (summed) Gram matrix of the evaluation (search rounds of a row-sharded session)
static int search_core(Session* s, int64_t n_2, double* d5, int64_t* r_it, bool gram_done) {
  Ctx* ctx = s->ctx;
  // only the lower part of the spectrum is consumed (and the largest eigenvalue for the positivity floor): eigenvalues
  // [0, n_2 + 1 + slack) cover the n_2 + 1 smallest positive ones unless more than `slack` are non-positive -- then all
  const int64_t slack = 64;
  const int64_t n_low = (n_2 + 1 + slack < s->n - 1) ? n_2 + 1 + slack : -1;
  if (gram_done) {
    SCL_TRY(eig_values(ctx, s->A, s->n, s->lda, s->w64, n_low));
    SCL_TRY(s->fetch_w(n_low >= 0));
  } else {
    SCL_TRY(decompose(s, s->pat.dev, s->val, 1, s->Btmp, (float)s->Kdiv, nullptr, n_low, /*binary=*/true));
  }
  int64_t r = s->count_positive();
  if (n_low >= 0 && (s->n - r) > slack) {  // (never seen: a Gram matrix of the path has at most a few null eigenvalues)
    SCL_TRY(stebz_redo_all(s));
    r = s->count_positive();
  }
  if (r_it) *r_it = r;
  // nV_2[:, end-n_2:end] (scLENS.jl:742): the n_2+1 smallest positive eigenvalues
  int64_t cnt = std::min<int64_t>(n_2 + 1, r);
  if (cnt < 5) return ctx->fail(SCLENS_ERR_ARG, "search_step: fewer than 5 columns to compare");
  const int64_t lo = s->n - r;
  SCL_TRY(s->ensure_zt(cnt));
  SCL_TRY(eig_vectors(ctx, s->A, s->n, s->lda, s->w64, lo, lo + cnt, s->Zt, s->ldz));
  unsigned* cmax = static_cast<unsigned*>(ctx->workspace("ses.cmax", sizeof(unsigned) * (size_t)cnt));
  if (!cmax) return SCLENS_ERR_OOM;
  {
    StageTimer tm(ctx, "corr");
    SCL_HIP(ctx, hipMemsetAsync(cmax, 0, sizeof(unsigned) * (size_t)cnt, ctx->stream));
    if (use_f16_corr(s)) {  // |Vr2' * nV_2| column maxima (scLENS.jl:742) on the fp16 MFMA from split operands
      if (s->Vr2h_of != s->Vr2t || s->Vr2h_epoch != ctx->ws_epoch) {  // once per binary basis (and again after a scratch release)
        s->Vr2h = ctx->workspace("ses.Vr2h", split_image_bytes(s->r_vr2, s->n));
        if (!s->Vr2h) return SCLENS_ERR_OOM;
        SCL_TRY(split_image_f16(ctx, s->Vr2t, s->r_vr2, s->n, s->ldz, s->Vr2h));
        s->Vr2h_of = s->Vr2t;
        s->Vr2h_epoch = ctx->ws_epoch;
      }
      void* zimg = ctx->workspace("ses.Zh", split_image_bytes(cnt, s->n));
      if (!zimg) return SCLENS_ERR_OOM;
      SCL_TRY(split_image_f16(ctx, s->Zt, cnt, s->n, s->ldz, zimg));
      SCL_TRY(corr_colabsmax_split(ctx, s->Vr2h, s->r_vr2, zimg, cnt, s->n, cmax));
    } else {
      GemmArgs g{};  // the same, never materialised, on the fp32 MFMA
      g.P = s->Vr2t; g.Q = s->Zt; g.C = nullptr;
      g.M = s->r_vr2; g.N = cnt; g.K = s->n;
      g.ldp = s->ldz; g.ldq = s->ldz; g.ldc = 0;
      g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = cmax;
      SCL_TRY(gemm_f32(ctx, g));
    }
  }
  std::vector<float> d(cnt);
  SCL_HIP(ctx, hipMemcpyAsync(d.data(), cmax, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::partial_sort(d.begin(), d.begin() + 5, d.end());
  for (int i = 0; i < 5; ++i) d5[i] = (double)d[i];
  return SCLENS_OK;
}

// One round of the sparsity search of a row-sharded session: every rank contributes its partial Gram matrix to each of the
// `count` evaluations, the sum of evaluation e lands on rank roots[e] (reduce; all-reduce if the host gave no reduce function),
// then every rank decomposes the one evaluation it is the root of. The partial Gram products of a round cost count / world of
// a full product per rank; the eigensolves -- 85 % of an evaluation -- run once each, in parallel, instead of on every rank.
// A round of a row-sharded session is a sequence of collectives; a rank that leaves it early (bad argument, out of memory for its
// scratch matrix) would leave its peers waiting in the next reduce for ever. Every round therefore starts with an agreement: each
// rank's local status (everything that can fail before the first collective has been tried by then) is summed over the ranks, and
// either all ranks enter the round or all return -- the failing ones their own code, the others SCLENS_ERR_STATE naming the cause.
static int round_entry_agreement(Session* s, int local_rc, const char* what) {
  Ctx* ctx = s->ctx;
  double* flag = static_cast<double*>(ctx->workspace("ses.roundflag", sizeof(double)));
  if (!flag) return local_rc != SCLENS_OK ? local_rc : SCLENS_ERR_OOM;  // (8 bytes: cannot fail in practice)
  const double mine = local_rc == SCLENS_OK ? 0.0 : 1.0;
  const std::string my_err = ctx->err;
  hipError_t e = hipMemcpyAsync(flag, &mine, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return ctx->fail(SCLENS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  SCL_TRY(s->sh.sum(ctx, flag, 1, 0));
  double total = 0.0;
  SCL_HIP(ctx, hipMemcpy(&total, flag, sizeof(double), hipMemcpyDeviceToHost));
  if (local_rc != SCLENS_OK) {
    ctx->err = my_err;
    return local_rc;
  }
  if (total != 0.0)
    return ctx->fail(SCLENS_ERR_STATE, std::string(what) + ": " + std::to_string((long long)(total + 0.5)) + " other rank(s) could not enter the round");
  return SCLENS_OK;
}

int session_search_round_seeded(Session* s, const uint64_t* seeds, const int64_t* m, const int32_t* roots, int count, int my_slot,
                                int64_t n_2, double* d5, int64_t* r_it) {
  Ctx* ctx = s->ctx;
  if (!s->sh.on()) return ctx->fail(SCLENS_ERR_STATE, "search_round: row-sharded sessions only (use search_step)");
  if (!s->Vr2t) return ctx->fail(SCLENS_ERR_STATE, "search_round: call binary_basis first");
  // everything that can fail on ONE rank before the first collective, then the agreement
  int pre = SCLENS_OK;
  if (count <= 0 || !seeds || !m || !roots || my_slot >= count) pre = ctx->fail(SCLENS_ERR_ARG, "search_round: bad arguments");
  for (int e = 0; pre == SCLENS_OK && e < count; ++e)
    if (m[e] < 0 || m[e] > s->pat.dev.population()) pre = ctx->fail(SCLENS_ERR_ARG, "search_round: bad sample size");
  float* scratch = nullptr;
  if (pre == SCLENS_OK && (count > 1 || my_slot < 0)) {
    scratch = static_cast<float*>(ctx->workspace("ses.Ascr", sizeof(float) * (size_t)s->n * s->lda));
    if (!scratch) pre = SCLENS_ERR_OOM;
  }
  SCL_TRY(round_entry_agreement(s, pre, "search_round"));
  for (int e = 0; e < count; ++e) {
    SCL_TRY(make_values_seeded(ctx, s->pat.dev, s->pat.base_val, 1, seeds[e], m[e], s->val));
    float* target = s->A;
    if (e != my_slot) target = scratch;
    SCL_TRY(decompose(s, s->pat.dev, s->val, 1, s->Btmp, (float)s->Kdiv, nullptr, -1, /*binary=*/true, /*solve=*/false, roots[e], target));
  }
  if (my_slot < 0) return SCLENS_OK;
  return search_core(s, n_2, d5, r_it, /*gram_done=*/true);
}

static int perturb_core(Session* s, int64_t t, int64_t min_pc, double* nL_top, int64_t* ncols);
int session_perturb(Session* s, int64_t t, const uint32_t* sample, int64_t m, int64_t min_pc, double* nL_top,
                    int64_t* ncols) {
  Ctx* ctx = s->ctx;
  if (t < 0 || min_pc <= 0) return ctx->fail(SCLENS_ERR_ARG, "perturb: bad slot / min_pc");
  if (s->chunked()) {
    if (!s->cands_counted) SCL_TRY(chunk_count_candidates(s));
    if (m < 0 || m > s->ncand_total) return ctx->fail(SCLENS_ERR_ARG, "perturb: bad sample size");
    SCL_TRY(s->upload_idx(sample, m));
    s->cspec = MatSpec{};
    s->cspec.cands = 1; s->cspec.sample = 1; s->cspec.m = m;
    return perturb_core(s, t, min_pc, nL_top, ncols);
  }
  if (m < 0 || m > s->pat.dev.population()) return ctx->fail(SCLENS_ERR_ARG, "perturb: bad sample size");
  SCL_TRY(s->upload_idx(sample, m));
  SCL_TRY(make_values(ctx, s->pat.dev, s->pat.base_val, 0, s->idx_dev, m, s->val));
  return perturb_core(s, t, min_pc, nL_top, ncols);
}
int session_perturb_seeded(Session* s, int64_t t, uint64_t seed, int64_t m, int64_t min_pc, double* nL_top,
                           int64_t* ncols) {
  Ctx* ctx = s->ctx;
  if (t < 0 || min_pc <= 0) return ctx->fail(SCLENS_ERR_ARG, "perturb: bad slot / min_pc");
  if (s->chunked()) {
    if (!s->cands_counted) SCL_TRY(chunk_count_candidates(s));
    if (m < 0 || m > s->ncand_total) return ctx->fail(SCLENS_ERR_ARG, "perturb: bad sample size");
    s->cspec = MatSpec{};
    s->cspec.cands = 1; s->cspec.sample = 2; s->cspec.seed = seed; s->cspec.m = m;
    return perturb_core(s, t, min_pc, nL_top, ncols);
  }
  SCL_TRY(make_values_seeded(ctx, s->pat.dev, s->pat.base_val, 0, seed, m, s->val));
  return perturb_core(s, t, min_pc, nL_top, ncols);
}
static int perturb_core(Session* s, int64_t t, int64_t min_pc, double* nL_top, int64_t* ncols) {
  Ctx* ctx = s->ctx;
  if ((int64_t)s->ens.size() <= t) { s->ens.resize(t + 1, nullptr); s->ens_cols.resize(t + 1, 0); }
  float* slot = static_cast<float*>(ctx->workspace("ses.ens" + std::to_string(t), sizeof(float) * (size_t)min_pc * s->ldn));
  if (!slot) return SCLENS_ERR_OOM;
  s->ens[t] = slot;
  // get_eigvec(logn_scale(pre_scale(tmp_X))) (scLENS.jl:775): closure path, divisor size(X,2) = M
  // only the first min_pc eigenpairs are consumed (:776): subspace iteration seeded with the data matrix's vectors. From
  // n = 16 000 the iteration applies the Gram matrix implicitly (two passes over the scaled matrix per block product) and the
  // n x n x K Gram product is skipped altogether (0.69 s per member at 100 000 x 30 000); it is formed only if the iteration
  // does not converge and the full solver has to run.
  const bool can_chefsi = s->use_chefsi && s->b0 >= min_pc + 8 && s->Z0t;
  const int64_t implicit_min_n = ctx->opt.implicit_min_n;
  const bool implicit_op = can_chefsi && !s->sh.on() && !s->chunked() && s->n >= implicit_min_n;
  if (s->chunked()) {  // the member's Gram matrix summed over the chunks (the scaled matrix is never whole: no implicit operator)
    SCL_TRY(chunked_gram(s, s->cspec, (float)s->M, nullptr, s->A, &s->st_last));
  } else if (s->sh.on()) {
    SCL_TRY(scale_to_dense_sharded(ctx, s->pat.dev, s->val, 1, s->Btmp, s->ldb, nullptr, s->sh));
    SCL_TRY(gram_f32(ctx, s->Btmp, s->n, s->K, s->ldb, (float)s->M, s->A, s->lda));
    SCL_TRY(s->sh.sum(ctx, s->A, s->n * s->lda, 1));
  } else {
    SCL_TRY(scale_to_dense(ctx, s->pat.dev, s->val, 1, s->centering, s->cells_major, s->Btmp, s->ldb, nullptr));
    if (!implicit_op) SCL_TRY(gram_f32(ctx, s->Btmp, s->n, s->K, s->ldb, (float)s->M, s->A, s->lda));
  }
  if (can_chefsi) {
    SCL_TRY(s->ensure_zt(min_pc));
    std::vector<double> wd(min_pc);
    int conv = 0, its = 0;
    SCL_TRY(topk_chefsi(ctx, implicit_op ? nullptr : s->A, s->n, s->lda, (int)min_pc, (int)std::min<int64_t>(s->k, min_pc), (int)s->b0,
                        s->Z0t, s->ldz, s->theta0.data(), wd.data(), s->Zt, s->ldz, &conv, &its, implicit_op ? s->Btmp : nullptr,
                        s->K, s->ldb, (float)s->M, s->chefsi_tail_gap, s->chefsi_tail_free));
    const double tol = 8.0 * 5.96e-8 * std::sqrt((double)s->n) * std::max(0.0, wd.empty() ? 0.0 : wd[0]);
    if (conv && wd[min_pc - 1] > tol) {  // all min_pc eigenvalues positive: c = min(min_pc, r) = min_pc
      s->chefsi_used += 1;
      s->ens_cols[t] = min_pc;
      if (ncols) *ncols = min_pc;
      for (int64_t q = 0; q < min_pc; ++q) nL_top[q] = wd[q];
      return to_cell_side(s, s->Btmp, min_pc, slot, /*desc_input=*/true);
    }
    s->chefsi_fallback += 1;
    if (implicit_op) SCL_TRY(gram_f32(ctx, s->Btmp, s->n, s->K, s->ldb, (float)s->M, s->A, s->lda));
  }
  SCL_TRY(eig_values(ctx, s->A, s->n, s->lda, s->w64));
  SCL_TRY(s->fetch_w());
  const int64_t r = s->count_positive();
  const int64_t c = std::min<int64_t>(min_pc, r);
  if (ncols) *ncols = c;
  for (int64_t q = 0; q < min_pc; ++q) nL_top[q] = (q < c) ? s->w_host[s->n - 1 - q] : 0.0;
  s->ens_cols[t] = c;
  if (c == 0) return SCLENS_OK;
  SCL_TRY(s->ensure_zt(c));
  SCL_TRY(eig_vectors(ctx, s->A, s->n, s->lda, s->w64, s->n - c, s->n, s->Zt, s->ldz));
  return to_cell_side(s, s->Btmp, c, slot);
}

// One round of the perturbation ensemble of a row-sharded session (scLENS.jl:771-778, members t[e]): partial Gram matrices summed
// onto the members' roots, the roots decompose in parallel, then -- member by member -- the root's leading gene-side vectors are
// shared (a sum in which the other ranks contribute zeros: min_pc x n floats) and every rank recovers its own cells of the
// member's cell-side vectors from its block of the member's scaled matrix, which it forms a second time (the normalisation is
// cheap next to keeping `count` scaled matrices).
int session_perturb_round_seeded(Session* s, const int64_t* t, const uint64_t* seeds, const int64_t* m, const int32_t* roots, int count,
                                 int my_slot, int64_t min_pc, double* nL_top, int64_t* ncols) {
  Ctx* ctx = s->ctx;
  if (!s->sh.on()) return ctx->fail(SCLENS_ERR_STATE, "perturb_round: row-sharded sessions only (use perturb)");
  const char* phase = "start";
  auto tag = [&](int rc) {  // which part of the round failed
    if (rc != SCLENS_OK) ctx->err = std::string("perturb_round [") + phase + "]: " + ctx->err;
    return rc;
  };
#define PR_TRY(expr) SCL_TRY(tag(expr))
  // everything that can fail on ONE rank before the first collective (arguments, the workspaces of the whole round), then the
  // agreement of round_entry_agreement: all ranks enter the round or none does
  int pre = SCLENS_OK;
  if (count <= 0 || !t || !seeds || !m || !roots || my_slot >= count || min_pc <= 0 || !nL_top || !ncols)
    pre = ctx->fail(SCLENS_ERR_ARG, "perturb_round: bad arguments");
  for (int e = 0; pre == SCLENS_OK && e < count; ++e)
    if (t[e] < 0 || m[e] < 0 || m[e] > s->pat.dev.population()) pre = ctx->fail(SCLENS_ERR_ARG, "perturb_round: bad slot / sample size");
  float *scratch = nullptr, *zs = nullptr, *mine = nullptr;
  double* ls = nullptr;
  if (pre == SCLENS_OK) {
    if (count > 1 || my_slot < 0) scratch = static_cast<float*>(ctx->workspace("ses.Ascr", sizeof(float) * (size_t)s->n * s->lda));
    zs = static_cast<float*>(ctx->workspace("ses.zshare", sizeof(float) * (size_t)min_pc * s->ldz));
    ls = static_cast<double*>(ctx->workspace("ses.lshare", sizeof(double) * (size_t)(min_pc + 1)));
    mine = static_cast<float*>(ctx->workspace("ses.zmine", sizeof(float) * (size_t)min_pc * s->ldz));
    if (((count > 1 || my_slot < 0) && !scratch) || !zs || !ls || !mine) pre = SCLENS_ERR_OOM;
    for (int e = 0; pre == SCLENS_OK && e < count; ++e) {
      if ((int64_t)s->ens.size() <= t[e]) { s->ens.resize(t[e] + 1, nullptr); s->ens_cols.resize(t[e] + 1, 0); }
      if (!ctx->workspace("ses.ens" + std::to_string(t[e]), sizeof(float) * (size_t)min_pc * s->ldn)) pre = SCLENS_ERR_OOM;
    }
  }
  SCL_TRY(round_entry_agreement(s, pre, "perturb_round"));
  for (int e = 0; e < count; ++e) {
    phase = "partial Gram";
    PR_TRY(make_values_seeded(ctx, s->pat.dev, s->pat.base_val, 0, seeds[e], m[e], s->val));
    float* target = (e != my_slot) ? scratch : s->A;
    PR_TRY(decompose(s, s->pat.dev, s->val, 1, s->Btmp, (float)s->M, nullptr, -1, false, /*solve=*/false, roots[e], target));
  }
  phase = "root eigensolve";
  // ---- the root's eigen-solve: leading min_pc pairs, rows of `mine` descending. A failure here is local to this rank while the
  // share phase below is a collective of all: the root then shares the marker -1 in place of its column count (and zeros), every
  // rank sees it in the summed buffer and all leave the round with an error after the exchange -- nobody waits for a peer.
  std::vector<double> my_l((size_t)min_pc + 1, 0.0);
  int root_rc = SCLENS_OK;
  std::string root_err;
  if (my_slot >= 0) {
    auto solve = [&]() -> int {
      int64_t c = 0;
      bool done = false;
      const bool can_chefsi = s->use_chefsi && s->b0 >= min_pc + 8 && s->Z0t;
      if (can_chefsi) {
        PR_TRY(s->ensure_zt(min_pc));
        std::vector<double> wd(min_pc);
        int conv = 0, its = 0;
        PR_TRY(topk_chefsi(ctx, s->A, s->n, s->lda, (int)min_pc, (int)std::min<int64_t>(s->k, min_pc), (int)s->b0, s->Z0t, s->ldz,
                            s->theta0.data(), wd.data(), s->Zt, s->ldz, &conv, &its, nullptr, 0, 0, 1.f, 0.05));  // rounds: always strict tails
        const double tol = 8.0 * 5.96e-8 * std::sqrt((double)s->n) * std::max(0.0, wd[0]);
        if (conv && wd[min_pc - 1] > tol) {
          s->chefsi_used += 1;
          c = min_pc;
          for (int64_t q = 0; q < min_pc; ++q) my_l[1 + q] = wd[q];
          PR_TRY(copy_rows_f32(ctx, s->Zt, c, s->n, s->ldz, mine, s->ldz));
          done = true;
        } else {
          s->chefsi_fallback += 1;
        }
      }
      if (!done) {
        PR_TRY(eig_values(ctx, s->A, s->n, s->lda, s->w64));
        PR_TRY(s->fetch_w());
        c = std::min<int64_t>(min_pc, s->count_positive());
        for (int64_t q = 0; q < c; ++q) my_l[1 + q] = s->w_host[s->n - 1 - q];
        if (c > 0) {
          PR_TRY(s->ensure_zt(c));
          PR_TRY(eig_vectors(ctx, s->A, s->n, s->lda, s->w64, s->n - c, s->n, s->Zt, s->ldz));
          PR_TRY(reverse_rows_f32(ctx, s->Zt, c, s->n, s->ldz, mine, s->ldz));
        }
      }
      my_l[0] = (double)c;
      return SCLENS_OK;
    };
    root_rc = solve();
    if (root_rc != SCLENS_OK) {
      root_err = ctx->err;
      std::fill(my_l.begin(), my_l.end(), 0.0);
      my_l[0] = -1.0;  // the marker every rank will see
    }
  }
  // ---- share member by member, recover the local cells
  for (int e = 0; e < count; ++e) {
    phase = "share + recover";
    float* slot = static_cast<float*>(ctx->workspace("ses.ens" + std::to_string(t[e]), sizeof(float) * (size_t)min_pc * s->ldn));  // exists (entry)
    if (!slot) return SCLENS_ERR_OOM;
    s->ens[t[e]] = slot;
    if (e == my_slot) {
      SCL_HIP(ctx, hipMemcpyAsync(ls, my_l.data(), sizeof(double) * (size_t)(min_pc + 1), hipMemcpyHostToDevice, ctx->stream));
      const int64_t c = std::max<int64_t>(0, (int64_t)my_l[0]);
      if (c > 0) SCL_HIP(ctx, hipMemcpyAsync(zs, mine, sizeof(float) * (size_t)c * s->ldz, hipMemcpyDeviceToDevice, ctx->stream));
      if (c < min_pc)
        SCL_HIP(ctx, hipMemsetAsync(zs + c * s->ldz, 0, sizeof(float) * (size_t)(min_pc - c) * s->ldz, ctx->stream));
    } else {
      SCL_HIP(ctx, hipMemsetAsync(ls, 0, sizeof(double) * (size_t)(min_pc + 1), ctx->stream));
      SCL_HIP(ctx, hipMemsetAsync(zs, 0, sizeof(float) * (size_t)min_pc * s->ldz, ctx->stream));
    }
    PR_TRY(s->sh.sum(ctx, ls, min_pc + 1, 0));
    PR_TRY(s->sh.sum(ctx, zs, min_pc * s->ldz, 1));
    std::vector<double> hl((size_t)min_pc + 1);
    SCL_HIP(ctx, hipMemcpyAsync(hl.data(), ls, sizeof(double) * (size_t)(min_pc + 1), hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hl[0] < -0.5) {  // the root of this member could not decompose it: every rank has seen the marker, all stop here
      if (e == my_slot && root_rc != SCLENS_OK) {
        ctx->err = root_err;
        return root_rc;
      }
      return ctx->fail(SCLENS_ERR_NOCONV, "perturb_round: the root of member " + std::to_string((long long)t[e]) + " (rank " +
                                              std::to_string(roots[e]) + ") failed to decompose it");
    }
    const int64_t c = std::min<int64_t>(min_pc, std::max<int64_t>(0, (int64_t)(hl[0] + 0.5)));  // (a sane count whatever the reducer does)
    ncols[e] = c;
    for (int64_t q = 0; q < min_pc; ++q) nL_top[(int64_t)e * min_pc + q] = (q < c) ? hl[1 + q] : 0.0;
    s->ens_cols[t[e]] = c;
    if (c == 0) continue;
    PR_TRY(make_values_seeded(ctx, s->pat.dev, s->pat.base_val, 0, seeds[e], m[e], s->val));
    PR_TRY(scale_to_dense_sharded(ctx, s->pat.dev, s->val, 1, s->Btmp, s->ldb, nullptr, s->sh));
    PR_TRY(to_cell_side(s, s->Btmp, c, slot, /*desc_input=*/true, zs));
  }
  return SCLENS_OK;
}
#undef PR_TRY

// Device buffers of the two read-only results that other ranks need when the first three decompositions are spread
// over the ranks (api.sclens with world > 1): Vr2 (what = 1; rows = positive eigenvalues of the binarised matrix) and the
// seed block of the partial eigensolver (what = 2; rows = b0, with its eigenvalues theta0 and the signal count k).
// rows > 0: receiver side, (re)allocate for `rows` rows and record the metadata; rows == 0: owner side, query.
int session_shared_buffer(Session* s, int what, int64_t rows, int64_t k, double* theta0, void** ptr, int64_t* rows_out,
                          int64_t* k_out, int64_t* ld) {
  Ctx* ctx = s->ctx;
  if (what != 1 && what != 2) return ctx->fail(SCLENS_ERR_ARG, "shared_buffer: what must be 1 (Vr2) or 2 (seed block)");
  if (rows < 0 || rows > s->n) return ctx->fail(SCLENS_ERR_ARG, "shared_buffer: bad row count");
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (what == 1) {
    if (rows > 0) {
      float* v = static_cast<float*>(ctx->workspace("ses.Vr2t", sizeof(float) * (size_t)rows * s->ldz));
      if (!v) return SCLENS_ERR_OOM;
      s->Vr2t = v;
      s->r_vr2 = rows;
    }
    if (!s->Vr2t) return ctx->fail(SCLENS_ERR_STATE, "shared_buffer: no Vr2 on this session");
    if (ptr) *ptr = s->Vr2t;
    if (rows_out) *rows_out = s->r_vr2;
  } else {
    if (rows > 0) {
      if (!theta0 || k < 0) return ctx->fail(SCLENS_ERR_ARG, "shared_buffer: the seed block needs k and theta0");
      float* z0 = static_cast<float*>(ctx->workspace("ses.Z0t", sizeof(float) * (size_t)rows * s->ldz));
      if (!z0) return SCLENS_ERR_OOM;
      s->Z0t = z0;
      s->b0 = rows;
      s->k = k;
      s->theta0.assign(theta0, theta0 + rows);
    } else if (s->Z0t && theta0) {
      std::copy(s->theta0.begin(), s->theta0.end(), theta0);
    }
    if (ptr) *ptr = s->Z0t;  // may be null: the data matrix gave no usable seed block (b0 = 0)
    if (rows_out) *rows_out = s->Z0t ? s->b0 : 0;
    if (k_out) *k_out = s->k;
  }
  if (ld) *ld = s->ldz;
  return SCLENS_OK;
}

int session_set_int(Session* s, const char* name, int64_t value) {
  const std::string k(name ? name : "");
  if (k == "chefsi") { s->use_chefsi = value != 0; return SCLENS_OK; }
  if (k == "chefsi_tail_gap_milli") { s->chefsi_tail_gap = (double)value * 1e-3; return SCLENS_OK; }
  if (k == "chefsi_tail_free") { s->chefsi_tail_free = value != 0; return SCLENS_OK; }
  if (k == "shard_rank" || k == "solve_root") {
    if (!s->sh.on()) return s->ctx->fail(SCLENS_ERR_STATE, "session_set_int: " + k + " is an option of row-sharded sessions");
    if (value < -1) return s->ctx->fail(SCLENS_ERR_ARG, "session_set_int: bad rank");
    if (k == "solve_root" && value >= 0 && s->shard_rank < 0) return s->ctx->fail(SCLENS_ERR_STATE, "session_set_int: set shard_rank before solve_root");
    (k == "shard_rank" ? s->shard_rank : s->solve_root) = (int)value;
    return SCLENS_OK;
  }
  if (k == "centering") {  // 0 = "mean", 1 = "median" (scLENS.jl:651-654); set before the first decomposition
    if (value != 0 && value != 1) return s->ctx->fail(SCLENS_ERR_ARG, "session_set_int: centering must be 0 or 1");
    if (value == 1 && s->sh.on()) return s->ctx->fail(SCLENS_ERR_ARG, "session_set_int: a row-sharded session supports mean centring only");
    if (s->have_spectrum) return s->ctx->fail(SCLENS_ERR_STATE, "session_set_int: centering is fixed once the data spectrum exists");
    s->centering = (int)value;
    return SCLENS_OK;
  }
  return s->ctx->fail(SCLENS_ERR_ARG, "session_set_int: unknown option " + k);
}
int session_get_int(Session* s, const char* name, int64_t* value) {
  const std::string k(name ? name : "");
  if (k == "chefsi_used") { *value = s->chefsi_used; return SCLENS_OK; }
  if (k == "chefsi_fallback") { *value = s->chefsi_fallback; return SCLENS_OK; }
  if (k == "chefsi") { *value = s->use_chefsi; return SCLENS_OK; }
  if (k == "chefsi_tail_gap_milli") { *value = (int64_t)(s->chefsi_tail_gap * 1e3 + 0.5); return SCLENS_OK; }
  if (k == "chefsi_tail_free") { *value = s->chefsi_tail_free; return SCLENS_OK; }
  if (k == "match_uncertain_count") {
    *value = 0;
    for (int u : s->match_uncertain) *value += u != 0;
    return SCLENS_OK;
  }
  if (k.compare(0, 16, "match_uncertain:") == 0) {  // member t of the last session_robustness
    const int64_t t = atoll(k.c_str() + 16);
    *value = (t >= 0 && t < (int64_t)s->match_uncertain.size()) ? s->match_uncertain[t] : 0;
    return SCLENS_OK;
  }
  if (k == "centering") { *value = s->centering; return SCLENS_OK; }
  if (k == "n_cand") {  // length of the candidate list the samples index (chunked session: counted on first use, one pattern build per chunk)
    if (s->chunked() && !s->cands_counted) SCL_TRY(chunk_count_candidates(s));
    *value = s->chunked() ? s->ncand_total : s->pat.dev.population();
    return SCLENS_OK;
  }
  if (k == "chunks") { *value = (int64_t)s->chunks.size(); return SCLENS_OK; }
  if (k == "chunk_builds") { *value = s->chunk_builds; return SCLENS_OK; }
  if (k == "chunk_visits") { *value = s->chunk_visits; return SCLENS_OK; }
  if (k == "chunk_cached") { *value = (int64_t)s->pcache.size(); return SCLENS_OK; }
  if (k == "gram_bits_used") { *value = s->ctx->gram_bits_used; return SCLENS_OK; }
  if (k == "gram_sparse_used") { *value = s->ctx->gram_sparse_used; return SCLENS_OK; }
  return s->ctx->fail(SCLENS_ERR_ARG, "session_get_int: unknown option " + k);
}

int session_get_perturbed(Session* s, int64_t t, float* out) {
  Ctx* ctx = s->ctx;
  if (t < 0 || t >= (int64_t)s->ens.size() || !s->ens[t]) return ctx->fail(SCLENS_ERR_ARG, "get_perturbed: empty slot");
  SCL_HIP(ctx, hipMemcpy2DAsync(out, sizeof(float) * s->N, s->ens[t], sizeof(float) * s->ldn, sizeof(float) * s->N,
                                s->ens_cols[t], hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

int64_t session_slot_ld(Session* s) { return s->ldn; }
int session_export_slot(Session* s, int64_t t, int64_t min_pc, void* dst) {
  Ctx* ctx = s->ctx;
  if (t < 0 || t >= (int64_t)s->ens.size() || !s->ens[t] || !dst) return ctx->fail(SCLENS_ERR_ARG, "export_slot: empty slot");
  SCL_HIP(ctx, hipMemcpyAsync(dst, s->ens[t], sizeof(float) * (size_t)min_pc * s->ldn, hipMemcpyDeviceToDevice, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}
int session_import_slot(Session* s, int64_t t, int64_t min_pc, int64_t ncols, const void* src) {
  Ctx* ctx = s->ctx;
  if (t < 0 || min_pc <= 0 || ncols < 0 || ncols > min_pc || !src) return ctx->fail(SCLENS_ERR_ARG, "import_slot: bad arguments");
  if ((int64_t)s->ens.size() <= t) { s->ens.resize(t + 1, nullptr); s->ens_cols.resize(t + 1, 0); }
  float* slot = static_cast<float*>(ctx->workspace("ses.ens" + std::to_string(t), sizeof(float) * (size_t)min_pc * s->ldn));
  if (!slot) return SCLENS_ERR_OOM;
  s->ens[t] = slot;
  s->ens_cols[t] = ncols;
  SCL_HIP(ctx, hipMemcpyAsync(slot, src, sizeof(float) * (size_t)min_pc * s->ldn, hipMemcpyDeviceToDevice, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

__global__ void k_gather_rows(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ pick, int64_t cols,
                              float* __restrict__ dst, int64_t ldd) {
  const int64_t q = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cols) dst[q * ldd + c] = src[(int64_t)pick[q] * lds + c];
}

int session_robustness(Session* s, int64_t P, int32_t* a_b, double* b) {
  Ctx* ctx = s->ctx;
  const int64_t k = s->k;
  if (k <= 0 || !s->nVt) return ctx->fail(SCLENS_ERR_STATE, "robustness: no signal vectors");
  if (P < 1 || (int64_t)s->ens.size() < P) return ctx->fail(SCLENS_ERR_STATE, "robustness: ensemble incomplete");
  int64_t cmax = 0;
  for (int64_t t = 0; t < P; ++t) {
    if (!s->ens[t] || s->ens_cols[t] <= 0) return ctx->fail(SCLENS_ERR_STATE, "robustness: empty ensemble slot");
    cmax = std::max(cmax, s->ens_cols[t]);
  }
  // a_b[:, t] = argmax_c |nV' * nV_set[t]| per signal (scLENS.jl:788)
  float* C1 = static_cast<float*>(ctx->workspace("ses.C1", sizeof(float) * (size_t)k * cmax));
  float* sub = static_cast<float*>(ctx->workspace("ses.sub", sizeof(float) * (size_t)P * k * s->ldn));
  int32_t* pick = static_cast<int32_t*>(ctx->workspace("ses.pick", sizeof(int32_t) * (size_t)k));
  if (!C1 || !sub || !pick) return SCLENS_ERR_OOM;
  std::vector<float> hC((size_t)k * cmax);
  std::vector<int32_t> hp(k);
  // Matching certificate. Signal vector i has unit length and the eigenvectors of a member are orthonormal, so every unit vector
  // orthogonal to the member's first k eigenvectors correlates with it by at most sqrt(1 - S_i), S_i = sum_{j<k} c_ij^2. If the best
  // of the first k beats that bound, the argmax of :788 lies among them WHATEVER the columns k .. min_pc-1 hold: those columns are
  // then not consumed, up to the angle error of the strict pairs (margin below; the partial eigensolver may leave the tail unconverged,
  // `chefsi_tail_free`). match_uncertain[t] = 1
  // where that cannot be shown for some signal (or a column >= k was picked); the caller solves such a member again.
  s->match_uncertain.assign((size_t)P, 0);
  for (int64_t t = 0; t < P; ++t) {
    const int64_t c = s->ens_cols[t];
    GemmArgs g{};
    g.P = s->nVt; g.Q = s->ens[t]; g.C = C1;
    g.M = k; g.N = c; g.K = s->N;
    g.ldp = s->ldn; g.ldq = s->ldn; g.ldc = cmax;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
    SCL_TRY(gemm_f32(ctx, g));
    SCL_TRY(s->sh.sum(ctx, C1, k * cmax, 1));  // contraction over cells: partial on a row-sharded session
    SCL_HIP(ctx, hipMemcpyAsync(hC.data(), C1, sizeof(float) * (size_t)k * cmax, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < k; ++i) {
      int32_t best = 0;
      float bv = -1.f;
      for (int64_t j = 0; j < c; ++j) {
        const float v = std::fabs(hC[i * cmax + j]);
        if (v > bv) { bv = v; best = (int32_t)j; }  // first maximum (Appendix A25)
      }
      hp[i] = best;
      a_b[i + t * k] = best;
      double S = 0.0, bk = 0.0;
      for (int64_t j = 0; j < std::min<int64_t>(k, c); ++j) {
        const double v = (double)hC[i * cmax + j];
        S += v * v;
        bk = std::max(bk, std::fabs(v));
      }
      // margin 2e-2: the bound holds for vectors orthogonal to the COMPUTED first k Ritz vectors, the reference's converged tail
      // vectors are orthogonal to the true ones -- the strict pairs are accepted at a residual of 2e-3 (theta_q - block edge), which
      // allows a tilt of up to res / (lambda_k - lambda_{k+1}) ~ 1e-2 when a weak signal sits next to the bulk edge; the margin covers
      // that tilt (and the ~1e-6 of the fp32 products). A member inside it is simply solved again with the tail converged.
      if (best >= k || bk * bk <= (1.0 - S) + 2e-2) s->match_uncertain[t] = 1;
    }
    SCL_HIP(ctx, hipMemcpyAsync(pick, hp.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((s->N + 255) / 256), (unsigned)k), dim3(256), 0, ctx->stream,
                       s->ens[t], s->ldn, pick, s->N, sub + (size_t)t * k * s->ldn, s->ldn);
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  // all pairwise |sub_i' sub_j| at once: G = Sub Sub^T ((P k) x (P k)), then row maxima per block (:792-795)
  const int64_t PK = P * k;
  float* G = static_cast<float*>(ctx->workspace("ses.G2", sizeof(float) * (size_t)PK * PK));
  if (!G) return SCLENS_ERR_OOM;
  {
    GemmArgs g{};
    g.P = sub; g.Q = sub; g.C = G;
    g.M = PK; g.N = PK; g.K = s->N;
    g.ldp = s->ldn; g.ldq = s->ldn; g.ldc = PK;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 1; g.colabsmax = nullptr;
    SCL_TRY(gemm_f32(ctx, g));
    SCL_TRY(s->sh.sum(ctx, G, PK * PK, 1));
  }
  std::vector<float> hG((size_t)PK * PK);
  SCL_HIP(ctx, hipMemcpyAsync(hG.data(), G, sizeof(float) * (size_t)PK * PK, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t npairs = P * (P - 1) / 2;
  int64_t col = 0;
  for (int64_t i = 0; i < P; ++i)
    for (int64_t j = i + 1; j < P; ++j, ++col)
      for (int64_t a = 0; a < k; ++a) {
        float mx = 0.f;
        for (int64_t c = 0; c < k; ++c) mx = std::max(mx, std::fabs(hG[(size_t)(i * k + a) * PK + (j * k + c)]));
        b[a * npairs + col] = (double)mx;
      }
  return SCLENS_OK;
}

__global__ void k_sum_slabs_f32(const float* __restrict__ part, int S, int64_t slab, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slab) return;
  float v = 0.f;
  for (int q = 0; q < S; ++q) v += part[(int64_t)q * slab + i];  // fixed order
  out[i] = v;
}

int session_gene_basis(Session* s, const double* nL, float* out) {
  Ctx* ctx = s->ctx;
  const int64_t k = s->k;
  if (k <= 0 || !s->nVt) return ctx->fail(SCLENS_ERR_STATE, "gene_basis: no signal vectors");
  float* G = static_cast<float*>(ctx->workspace("ses.gb", sizeof(float) * (size_t)k * round_up(s->M, 32)));
  float* sc = static_cast<float*>(ctx->workspace("ses.gbs", sizeof(float) * (size_t)k));
  if (!G || !sc) return SCLENS_ERR_OOM;
  const int64_t ldg = round_up(s->M, 32);
  if (s->chunked()) {  // G = sum over the chunks of nV[cells of the chunk]' * block (the contraction runs over the cells)
    float* nc = static_cast<float*>(ctx->workspace("ses.gbc", sizeof(float) * (size_t)k * s->ldb));
    if (!nc) return SCLENS_ERR_OOM;
    for (size_t g = 0; g < s->chunks.size(); ++g) {
      const ChunkSrc& c = s->chunks[g];
      SCL_TRY(chunk_block(s, s->st_data, (int)g));
      SCL_HIP(ctx, hipMemcpy2DAsync(nc, sizeof(float) * s->ldb, s->nVt + c.row0, sizeof(float) * s->ldn, sizeof(float) * c.N, k,
                                    hipMemcpyDeviceToDevice, ctx->stream));
      GemmArgs a{};
      a.P = nc; a.Q = s->Btmp; a.C = G;
      a.M = k; a.N = s->M; a.K = c.N;
      a.ldp = s->ldb; a.ldq = s->ldb; a.ldc = ldg;
      a.alpha = 1.f; a.beta = g > 0 ? 1.f : 0.f; a.q_kcontig = 1; a.lower = 0; a.colabsmax = nullptr;
      SCL_TRY(gemm_f32(ctx, a));
    }
    std::vector<float> hsc(k);
    for (int64_t q = 0; q < k; ++q) hsc[q] = (float)(1.0 / std::sqrt(nL[q]) / std::sqrt((double)s->M));
    SCL_HIP(ctx, hipMemcpyAsync(sc, hsc.data(), sizeof(float) * k, hipMemcpyHostToDevice, ctx->stream));
    SCL_TRY(scale_rows_f32(ctx, G, k, s->M, ldg, sc));
    SCL_HIP(ctx, hipMemcpy2DAsync(out, sizeof(float) * s->M, G, sizeof(float) * ldg, sizeof(float) * s->M, k, hipMemcpyDeviceToHost, ctx->stream));
    SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCLENS_OK;
  }
  // k rows only: the product streams the whole scaled matrix once, so the contraction over the cells is split over the
  // grid (one tile row of M / 128 blocks would read 4 N bytes per gene each from a single CU: 3.7 s at 100 000 x 30 000)
  const int64_t tiles = (s->M + 127) / 128;
  int S = (int)std::min<int64_t>(32, std::max<int64_t>(1, (1024 + tiles - 1) / tiles));
  const int64_t kch = round_up((s->N + S - 1) / S, 32);
  S = (int)((s->N + kch - 1) / kch);
  float* Gp = static_cast<float*>(ctx->workspace("ses.gbp", sizeof(float) * (size_t)S * k * ldg));
  if (!Gp) return SCLENS_ERR_OOM;
  GemmArgs g{};
  g.P = s->nVt; g.Q = s->Bmain; g.C = Gp;
  g.M = k; g.N = s->M; g.K = s->N;
  g.ldp = s->ldn; g.ldq = s->ldb; g.ldc = ldg;
  g.alpha = 1.f; g.beta = 0.f; g.lower = 0; g.colabsmax = nullptr;
  g.q_kcontig = s->cells_major ? 0 : 1;  // Bmain is [N][M] (NN) or [M][N] (NT)
  g.splits = S; g.k_chunk = kch; g.c_split_off = k * ldg;
  SCL_TRY(gemm_f32(ctx, g));
  hipLaunchKernelGGL(k_sum_slabs_f32, dim3((unsigned)((k * ldg + 255) / 256)), dim3(256), 0, ctx->stream, Gp, S, k * ldg, G);
  SCL_HIP(ctx, hipGetLastError());
  SCL_TRY(s->sh.sum(ctx, G, k * ldg, 1));
  std::vector<float> hs(k);
  for (int64_t q = 0; q < k; ++q) hs[q] = (float)(1.0 / std::sqrt(nL[q]) / std::sqrt((double)s->M));
  SCL_HIP(ctx, hipMemcpyAsync(sc, hs.data(), sizeof(float) * k, hipMemcpyHostToDevice, ctx->stream));
  SCL_TRY(scale_rows_f32(ctx, G, k, s->M, ldg, sc));
  SCL_HIP(ctx, hipMemcpy2DAsync(out, sizeof(float) * s->M, G, sizeof(float) * ldg, sizeof(float) * s->M, k,
                                hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

// ------------------------------------------------------------------------------------------------ drop-ins (A)
static int upload_padded(Ctx* ctx, const float* h, int64_t rows, int64_t cols, float* d, int64_t ld) {
  SCL_HIP(ctx, hipMemsetAsync(d, 0, sizeof(float) * (size_t)rows * ld, ctx->stream));
  SCL_HIP(ctx, hipMemcpy2DAsync(d, sizeof(float) * ld, h, sizeof(float) * cols, sizeof(float) * cols, rows,
                                hipMemcpyHostToDevice, ctx->stream));
  return SCLENS_OK;
}
static int download_packed(Ctx* ctx, const float* d, int64_t rows, int64_t cols, int64_t ld, float* h) {
  SCL_HIP(ctx, hipMemcpy2DAsync(h, sizeof(float) * cols, d, sizeof(float) * ld, sizeof(float) * cols, rows,
                                hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

int wishart_host(Ctx* ctx, const float* X, int64_t N, int64_t M, int dims, float* Y) {
  if (!X || !Y || N <= 0 || M <= 0 || (dims != 1 && dims != 2)) return ctx->fail(SCLENS_ERR_ARG, "wishart: bad arguments");
  // column-major N x M == row-major [M][N]
  const int64_t ldx = round_up(N, 32);
  SCL_WS(ctx, dX, float, "w.X", M * ldx);
  SCL_TRY(upload_padded(ctx, X, M, N, dX, ldx));
  if (dims == 2) {  // X'X / M : rows = genes, K = N
    const int64_t lda = round_up(M, 32);
    SCL_WS(ctx, dA, float, "w.A", M * lda);
    SCL_TRY(gram_f32(ctx, dX, M, N, ldx, (float)M, dA, lda, /*allow_split=*/false));
    return download_packed(ctx, dA, M, M, lda, Y);
  }
  const int64_t ldt = round_up(M, 32), lda = round_up(N, 32);
  SCL_WS(ctx, dT, float, "w.T", N * ldt);
  SCL_WS(ctx, dA, float, "w.A", N * lda);
  SCL_HIP(ctx, hipMemsetAsync(dT, 0, sizeof(float) * (size_t)N * ldt, ctx->stream));
  SCL_TRY(transpose_f32(ctx, dX, M, N, ldx, dT, ldt));
  SCL_TRY(gram_f32(ctx, dT, N, M, ldt, (float)M, dA, lda, /*allow_split=*/false));  // XX' / size(X,2)
  return download_packed(ctx, dA, N, N, lda, Y);
}

int get_eigen_host(Ctx* ctx, const float* Y, int64_t n, float* L, float* V) {
  if (!Y || !L || n <= 0) return ctx->fail(SCLENS_ERR_ARG, "get_eigen: bad arguments");
  const int64_t lda = round_up(n, 32);
  SCL_WS(ctx, dA, float, "e.A", n * lda);
  SCL_WS(ctx, dw, double, "e.w", n);
  SCL_TRY(upload_padded(ctx, Y, n, n, dA, lda));
  float* dZ = nullptr;
  if (V) {
    dZ = static_cast<float*>(ctx->workspace("e.Z", sizeof(float) * (size_t)n * lda));
    if (!dZ) return SCLENS_ERR_OOM;
  }
  SCL_TRY(eigh_f32(ctx, dA, n, lda, dw, 0, V ? n : 0, dZ, lda));
  std::vector<double> w(n);
  SCL_HIP(ctx, hipMemcpyAsync(w.data(), dw, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < n; ++i) {
    if (w[i] != w[i]) return ctx->fail(SCLENS_ERR_NAN, "get_eigen: NaN eigenvalue");
    L[i] = (float)w[i];
  }
  if (V) return download_packed(ctx, dZ, n, n, lda, V);  // row k = eigenvector k == column k in column-major
  return SCLENS_OK;
}

int corr_mat_host(Ctx* ctx, const float* X, int64_t n, int64_t p, const float* Yv, int64_t q, float* out) {
  if (!X || !Yv || !out || n <= 0 || p <= 0 || q <= 0) return ctx->fail(SCLENS_ERR_ARG, "corr_mat: bad arguments");
  const int64_t ld = round_up(n, 32), ldc = round_up(p, 32);
  SCL_WS(ctx, dX, float, "c.X", p * ld);
  SCL_WS(ctx, dY, float, "c.Y", q * ld);
  SCL_WS(ctx, dC, float, "c.C", q * ldc);
  SCL_TRY(upload_padded(ctx, X, p, n, dX, ld));
  SCL_TRY(upload_padded(ctx, Yv, q, n, dY, ld));
  GemmArgs g{};  // column-major p x q == row-major [q][p] = Y' X
  g.P = dY; g.Q = dX; g.C = dC;
  g.M = q; g.N = p; g.K = n;
  g.ldp = ld; g.ldq = ld; g.ldc = ldc;
  g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = nullptr;
  SCL_TRY(gemm_f32(ctx, g));
  return download_packed(ctx, dC, q, p, ldc, out);
}

// out[j] = max_i |X_i' Y_j| (the search statistic, scLENS.jl:742), fp32 product or split fp16 images (unit-test piece)
int corr_colmax_host(Ctx* ctx, const float* X, int64_t n, int64_t p, const float* Yv, int64_t q, int use_split, float* out) {
  if (!X || !Yv || !out || n <= 0 || p <= 0 || q <= 0) return ctx->fail(SCLENS_ERR_ARG, "corr_colmax: bad arguments");
  const int64_t ld = round_up(n, 32);
  SCL_WS(ctx, dX, float, "c.X", p * ld);
  SCL_WS(ctx, dY, float, "c.Y", q * ld);
  SCL_WS(ctx, cmax, unsigned, "c.max", q);
  SCL_TRY(upload_padded(ctx, X, p, n, dX, ld));
  SCL_TRY(upload_padded(ctx, Yv, q, n, dY, ld));
  SCL_HIP(ctx, hipMemsetAsync(cmax, 0, sizeof(unsigned) * (size_t)q, ctx->stream));
  if (use_split) {
    void* xi = ctx->workspace("c.Xh", split_image_bytes(p, n));
    void* yi = ctx->workspace("c.Yh", split_image_bytes(q, n));
    if (!xi || !yi) return SCLENS_ERR_OOM;
    SCL_TRY(split_image_f16(ctx, dX, p, n, ld, xi));
    SCL_TRY(split_image_f16(ctx, dY, q, n, ld, yi));
    StageTimer tm(ctx, "corr");  // the product alone (scripts/perf_split_products.py)
    SCL_TRY(corr_colabsmax_split(ctx, xi, p, yi, q, n, cmax));
  } else {
    GemmArgs g{};
    g.P = dX; g.Q = dY; g.C = nullptr;
    g.M = p; g.N = q; g.K = n;
    g.ldp = ld; g.ldq = ld; g.ldc = 0;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 1; g.lower = 0; g.colabsmax = cmax;
    SCL_TRY(gemm_f32(ctx, g));
  }
  SCL_HIP(ctx, hipMemcpyAsync(out, cmax, sizeof(float) * (size_t)q, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SCLENS_OK;
}

// logn_scale(pre_scale(x)) (scLENS.jl:650-654) / the inline twin of the data matrix (:676-696) as a per-call drop-in:
// host CSC in, dense column-major N x M fp32 out.
int scale_csc_host(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                   int centering, int f32path, float* out, ScaleVecs* keep) {
  if (!colptr || !rowval || !nzval || !out || N <= 0 || M <= 0) return ctx->fail(SCLENS_ERR_ARG, "scale_csc: bad arguments");
  if (centering != 0 && centering != 1) return ctx->fail(SCLENS_ERR_ARG, "scale_csc: centering must be 0 (mean) or 1 (median)");
  if (centering == 1 && keep) return ctx->fail(SCLENS_ERR_ARG, "scale_csc: centering=median has no rec_vals");
  PatternOwner pr;
  SCL_TRY(pattern_build(ctx, N, M, colptr, rowval, nzval, 0, nullptr, nullptr, &pr));
  const int64_t ldb = round_up(N, 32);
  float* val = static_cast<float*>(ctx->workspace("w.scval", sizeof(float) * pr.dev.val_floats()));
  float* B = static_cast<float*>(ctx->workspace("w.scB", sizeof(float) * (size_t)M * ldb));
  int rc = (val && B) ? SCLENS_OK : SCLENS_ERR_OOM;
  if (rc == SCLENS_OK) rc = make_values(ctx, pr.dev, pr.base_val, 0, nullptr, 0, val);
  if (rc == SCLENS_OK) rc = scale_to_dense(ctx, pr.dev, val, centering ? 1 : f32path, centering, /*cells_major=*/0, B, ldb, keep);
  if (rc == SCLENS_OK) rc = download_packed(ctx, B, M, N, ldb, out);
  hipStreamSynchronize(ctx->stream);
  pattern_free(&pr);
  return rc;
}

// Gram matrix of the scaled binarised counts, both ways (unit-test piece, see sclens_hip.h)
int gram_binary_host(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int use_bits,
                     float divisor, float* out) {
  if (!colptr || !rowval || !nzval || !out || N <= 0 || M <= 0 || N < M) return ctx->fail(SCLENS_ERR_ARG, "gram_binary: bad arguments");
  PatternOwner pr;
  SCL_TRY(pattern_build(ctx, N, M, colptr, rowval, nzval, 0, nullptr, nullptr, &pr));
  const int64_t ldb = round_up(N, 64), lda = round_up(M, 32);
  float* val = static_cast<float*>(ctx->workspace("w.scval", sizeof(float) * pr.dev.val_floats()));
  float* B = static_cast<float*>(ctx->workspace("w.scB", sizeof(float) * (size_t)M * ldb));
  float* A = static_cast<float*>(ctx->workspace("w.A", sizeof(float) * (size_t)M * lda));
  int rc = (val && B && A) ? SCLENS_OK : SCLENS_ERR_OOM;
  if (rc == SCLENS_OK) rc = make_values(ctx, pr.dev, pr.base_val, /*binary=*/1, nullptr, 0, val);
  if (rc == SCLENS_OK) {
    if (use_bits) {
      rc = gram_binary(ctx, pr.dev, val, 1, B, divisor, A, lda);
    } else {
      rc = scale_to_dense(ctx, pr.dev, val, 1, 0, /*cells_major=*/0, B, ldb, nullptr);
      if (rc == SCLENS_OK) rc = gram_f32(ctx, B, M, N, ldb, divisor, A, lda);
    }
  }
  if (rc == SCLENS_OK) rc = download_packed(ctx, A, M, M, lda, out);
  hipStreamSynchronize(ctx->stream);
  pattern_free(&pr);
  return rc;
}

// Gram matrix of logn_scale(pre_scale(X)) / divisor (f32path = 1) or of the inline Float64 twin (f32path = 0) for a COUNT-VALUED matrix,
// both ways (unit-test piece and the A/B of SURVEY 8f-1): mode 0 = scaled dense matrix + the dense product the context's precision
// selects, mode 1 = from the sparse structure (gram_sparse.hip). binary: every stored count as 1.
int gram_counts_host(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int mode, int f32path,
                     int binary, float divisor, float* out) {
  if (!colptr || !rowval || !nzval || !out || N <= 0 || M <= 0 || N < M || (mode != 0 && mode != 1))
    return ctx->fail(SCLENS_ERR_ARG, "gram_counts: bad arguments");
  PatternOwner pr;
  SCL_TRY(pattern_build(ctx, N, M, colptr, rowval, nzval, 0, nullptr, nullptr, &pr));
  const int64_t ldb = round_up(N, 32), lda = round_up(M, 32);
  float* val = static_cast<float*>(ctx->workspace("w.scval", sizeof(float) * pr.dev.val_floats()));
  float* A = static_cast<float*>(ctx->workspace("w.A", sizeof(float) * (size_t)M * lda));
  int rc = (val && A) ? SCLENS_OK : SCLENS_ERR_OOM;
  if (rc == SCLENS_OK) rc = make_values(ctx, pr.dev, pr.base_val, binary, nullptr, 0, val);
  if (rc == SCLENS_OK) {
    if (mode == 1) {
      ScaleStats ss;
      rc = scale_to_dense_stats(ctx, pr.dev, val, f32path, 0, 0, nullptr, ldb, nullptr, &ss);
      if (rc == SCLENS_OK)
        rc = gram_sparse(ctx, pr.dev, val, f32path, ss.tgc, ss.lg, ss.stdv, ss.mu, ss.l2, ss.cent, ss.red + 1, (double)N, 1.0 / (double)divisor,
                         (double)N / (double)divisor, A, lda, false);
    } else {
      float* B = static_cast<float*>(ctx->workspace("w.scB", sizeof(float) * (size_t)M * ldb));
      rc = B ? SCLENS_OK : SCLENS_ERR_OOM;
      if (rc == SCLENS_OK) rc = scale_to_dense(ctx, pr.dev, val, f32path, 0, /*cells_major=*/0, B, ldb, nullptr);
      if (rc == SCLENS_OK) rc = gram_f32(ctx, B, M, N, ldb, divisor, A, lda);
    }
  }
  if (rc == SCLENS_OK) rc = download_packed(ctx, A, M, M, lda, out);
  hipStreamSynchronize(ctx->stream);
  pattern_free(&pr);
  return rc;
}

int get_eigvec_host(Ctx* ctx, const float* X, int64_t N, int64_t M, int64_t keep_top, float* nL, float* nV, int64_t* r) {
  if (!X || !nL || !r || N <= 0 || M <= 0) return ctx->fail(SCLENS_ERR_ARG, "get_eigvec: bad arguments");
  const int64_t n = std::min(N, M), K = std::max(N, M);
  const int64_t cap = *r;
  const int64_t ldx = round_up(N, 32);
  SCL_WS(ctx, dX, float, "w.X", M * ldx);  // [M][N]
  SCL_TRY(upload_padded(ctx, X, M, N, dX, ldx));
  const float* B = dX;
  int64_t ldb = ldx;
  if (N <= M) {  // need [N][M]
    const int64_t ldt = round_up(M, 32);
    SCL_WS(ctx, dT, float, "w.T", N * ldt);
    SCL_HIP(ctx, hipMemsetAsync(dT, 0, sizeof(float) * (size_t)N * ldt, ctx->stream));
    SCL_TRY(transpose_f32(ctx, dX, M, N, ldx, dT, ldt));
    B = dT;
    ldb = ldt;
  }
  const int64_t lda = round_up(n, 32);
  SCL_WS(ctx, dA, float, "w.A", n * lda);
  SCL_WS(ctx, dw, double, "e.w", n);
  SCL_TRY(gram_f32(ctx, B, n, K, ldb, (float)M, dA, lda, /*allow_split=*/false));
  SCL_TRY(eig_values(ctx, dA, n, lda, dw));
  std::vector<double> w(n);
  SCL_HIP(ctx, hipMemcpyAsync(w.data(), dw, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  SCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (double v : w)
    if (v != v) return ctx->fail(SCLENS_ERR_NAN, "get_eigvec: NaN eigenvalue");
  const int64_t rp = count_positive_tol(w);
  *r = rp;
  if (rp > cap) return ctx->fail(SCLENS_ERR_ARG, "get_eigvec: output capacity too small");
  for (int64_t c = 0; c < rp; ++c) nL[c] = (float)w[n - 1 - c];
  int64_t nv = (keep_top > 0) ? std::min(keep_top, rp) : rp;
  if (!nV || nv == 0) return SCLENS_OK;
  SCL_WS(ctx, dZ, float, "e.Z", nv * lda);
  SCL_TRY(eig_vectors(ctx, dA, n, lda, dw, n - nv, n, dZ, lda));
  const int64_t ldn = round_up(N, 32);
  SCL_WS(ctx, dO, float, "w.O", nv * ldn);
  if (N > M) {
    SCL_WS(ctx, dR, float, "w.R", nv * ldn);
    GemmArgs g{};
    g.P = dZ; g.Q = dX; g.C = dR;
    g.M = nv; g.N = N; g.K = M;
    g.ldp = lda; g.ldq = ldx; g.ldc = ldn;
    g.alpha = 1.f; g.beta = 0.f; g.q_kcontig = 0; g.lower = 0; g.colabsmax = nullptr;
    SCL_TRY(gemm_f32(ctx, g));
    SCL_TRY(normalize_rows_f32(ctx, dR, nv, N, ldn));
    SCL_TRY(reverse_rows_f32(ctx, dR, nv, N, ldn, dO, ldn));
  } else {
    SCL_TRY(reverse_rows_f32(ctx, dZ, nv, N, lda, dO, ldn));
  }
  return download_packed(ctx, dO, nv, N, ldn, nV);
}

}  // namespace scl
