// extern "C" surface of libsclens_hip.so (see include/sclens_hip.h). Thin: argument checks + forwarding.
#include <algorithm>

#include "common.h"
#include "pattern.h"

namespace scl {
struct Session;
int session_create(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int64_t, const uint32_t*,
                   const uint32_t*, Session**);
int session_create_sharded(Ctx*, int64_t, int64_t, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int64_t,
                           const uint32_t*, const uint32_t*, sclens_hip_allreduce_fn, void*, Session**);
int session_set_reducer(Session*, sclens_hip_allreduce_fn, void*);
int session_set_reduce_to(Session*, sclens_hip_reduce_fn, void*);
int session_create_sharded_drawn(Ctx*, int64_t, int64_t, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int64_t, uint64_t,
                                 sclens_hip_allreduce_fn, void*, Session**, int64_t*);
int session_set_candidate_range(Session*, int64_t, int64_t);
int session_create_chunked(Ctx*, int64_t, int64_t, int, int64_t, uint64_t, Session**);
int session_chunk_add(Session*, int, int, int64_t, int64_t, const int64_t*, const int32_t*, const float*);
int session_chunk_commit(Session*);
int session_null_spectrum_chunked(Session*, double*);
int session_local_candidates(Session*, uint32_t*, uint32_t*);
int session_search_round_seeded(Session*, const uint64_t*, const int64_t*, const int32_t*, int, int, int64_t, double*, int64_t*);
int session_perturb_round_seeded(Session*, const int64_t*, const uint64_t*, const int64_t*, const int32_t*, int, int, int64_t, double*, int64_t*);
int session_shared_buffer(Session*, int, int64_t, int64_t, double*, void**, int64_t*, int64_t*, int64_t*);
void session_destroy(Session*);
int session_clone(Ctx*, Session*, Session**);
int session_spectrum(Session*, const int64_t*, const int32_t*, const float*, double*, double*, ScaleVecs*);
int session_null_spectrum(Session*, const int64_t*, const int32_t*, const float*, double*);
int session_data_spectrum(Session*, double*, ScaleVecs*);
int session_adopt(Session*, Session*, int);
int pattern_create_drawn(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, uint64_t, PatternOwner**, int64_t*);
int pattern_candidates(Ctx*, PatternOwner*, uint32_t*, uint32_t*);
int pattern_download(Ctx*, PatternOwner*, int, void*);
int pattern_create(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int64_t, const uint32_t*,
                   const uint32_t*, PatternOwner**);
void pattern_destroy(PatternOwner*);
int session_set_pattern(Session*, PatternOwner*);
int session_create_from_counts(Ctx*, const Counts*, Session**);
int pattern_create_drawn_from_counts(Ctx*, const Counts*, uint64_t, PatternOwner**, int64_t*);
int counts_upload(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, Counts**);
int counts_download(Ctx*, const Counts*, int64_t*, int32_t*, float*);
int preprocess_keep(Ctx*, Counts**);
int session_null_spectrum_pattern(Session*, PatternOwner*, double*);
int session_signal_vectors(Session*, int64_t, float*);
int session_refine_eigenvalues(Session*, int64_t, int64_t, double*);
int session_binary_basis(Session*, double*, int64_t*);
int session_search_step(Session*, const uint32_t*, int64_t, int64_t, double*, int64_t*);
int session_perturb(Session*, int64_t, const uint32_t*, int64_t, int64_t, double*, int64_t*);
int session_get_perturbed(Session*, int64_t, float*);
int session_search_step_seeded(Session*, uint64_t, int64_t, int64_t, double*, int64_t*);
int session_perturb_seeded(Session*, int64_t, uint64_t, int64_t, int64_t, double*, int64_t*);
int session_robustness(Session*, int64_t, int32_t*, double*);
int session_gene_basis(Session*, const double*, float*);
int session_set_int(Session*, const char*, int64_t);
int session_get_int(Session*, const char*, int64_t*);
int64_t session_slot_ld(Session*);
int session_export_slot(Session*, int64_t, int64_t, void*);
int session_import_slot(Session*, int64_t, int64_t, int64_t, const void*);
int wishart_host(Ctx*, const float*, int64_t, int64_t, int, float*);
int get_eigen_host(Ctx*, const float*, int64_t, float*, float*);
int corr_mat_host(Ctx*, const float*, int64_t, int64_t, const float*, int64_t, float*);
int get_eigvec_host(Ctx*, const float*, int64_t, int64_t, int64_t, float*, float*, int64_t*);
int scale_csc_host(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int, int, float*, ScaleVecs*);
int corr_colmax_host(Ctx*, const float*, int64_t, int64_t, const float*, int64_t, int, float*);
int gram_binary_host(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int, float, float*);
int gram_counts_host(Ctx*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int, int, int, float, float*);
int denoise_host(Ctx*, const float*, int64_t, int64_t, const float*, int64_t, const double*, const double*, const double*,
                 const double*, const double*, float*);
}  // namespace scl

using scl::Ctx;

struct sclens_hip_session { scl::Session* s; sclens_hip_ctx* ctx; std::vector<double> scratch; };

extern "C" {

const char* sclens_hip_version(void) { return "sclens_hip 0.1 (gfx950)"; }

int sclens_hip_create(sclens_hip_ctx** out, int device_id) {
  if (!out) return SCLENS_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SCLENS_ERR_NO_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return SCLENS_ERR_NO_DEVICE;
  sclens_hip_ctx* h = new sclens_hip_ctx();
  h->c.device = device_id;
  if (hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->c.ev0) != hipSuccess || hipEventCreate(&h->c.ev1) != hipSuccess) {
    delete h;
    return SCLENS_ERR_NO_DEVICE;
  }
  // the one place the library's tunables meet the environment: once per context, never per call (common.h, SCL_OPTION_TABLE)
  if (const char* eo = getenv("SCLENS_HIP_OPTIONS")) {
    std::string bad;
    h->c.opt.parse(eo, &bad);
    if (!bad.empty()) {
      h->c.err = "SCLENS_HIP_OPTIONS: not understood: " + bad;
      std::string msg = h->c.err;
      sclens_hip_destroy(h);
      fprintf(stderr, "[sclens_hip] %s\n", msg.c_str());
      return SCLENS_ERR_ARG;
    }
  }
  *out = h;
  return SCLENS_OK;
}
int sclens_hip_set_option(sclens_hip_ctx* h, const char* name, int64_t value) {
  if (!h || !name) return SCLENS_ERR_ARG;
  if (h->c.opt.set(name, value)) return SCLENS_OK;
  return h->c.fail(SCLENS_ERR_ARG, std::string("set_option: unknown option ") + name);
}
int sclens_hip_get_option(sclens_hip_ctx* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return SCLENS_ERR_ARG;
  if (h->c.opt.get(name, value)) return SCLENS_OK;
  return h->c.fail(SCLENS_ERR_ARG, std::string("get_option: unknown option ") + name);
}
int sclens_hip_copy_options(sclens_hip_ctx* dst, const sclens_hip_ctx* src) {
  if (!dst || !src) return SCLENS_ERR_ARG;
  dst->c.opt = src->c.opt;
  return SCLENS_OK;
}

void sclens_hip_destroy(sclens_hip_ctx* h) {
  if (!h) return;
  hipSetDevice(h->c.device);
  h->c.release_all();
  if (h->c.ev0) hipEventDestroy(h->c.ev0);
  if (h->c.ev1) hipEventDestroy(h->c.ev1);
  for (hipEvent_t e : h->c.prof_ev) hipEventDestroy(e);
  if (h->c.aux_stream) hipStreamDestroy(h->c.aux_stream);
  for (hipEvent_t e : h->c.aux_ev)
    if (e) hipEventDestroy(e);
  if (h->c.q2_ev) hipEventDestroy(h->c.q2_ev);
  if (h->c.q1_ev) hipEventDestroy(h->c.q1_ev);
  if (h->c.stream) hipStreamDestroy(h->c.stream);
  delete h;
}

const char* sclens_hip_last_error(const sclens_hip_ctx* h) { return h ? h->c.err.c_str() : "null context"; }
void* sclens_hip_stream(sclens_hip_ctx* h) { return h ? (void*)h->c.stream : nullptr; }


int sclens_hip_set_timing(sclens_hip_ctx* h, int enabled) {
  if (!h) return SCLENS_ERR_ARG;
  h->c.timing = enabled != 0;
  return SCLENS_OK;
}
int sclens_hip_reset_timing(sclens_hip_ctx* h) {
  if (!h) return SCLENS_ERR_ARG;
  h->c.t_ms.clear();
  h->c.t_calls.clear();
  return SCLENS_OK;
}
int sclens_hip_get_timing(sclens_hip_ctx* h, const char* stage, double* total_ms, int64_t* calls) {
  if (!h || !stage) return SCLENS_ERR_ARG;
  auto it = h->c.t_ms.find(stage);
  if (total_ms) *total_ms = (it == h->c.t_ms.end()) ? 0.0 : it->second;
  if (calls) *calls = (it == h->c.t_ms.end()) ? 0 : (int64_t)h->c.t_calls[stage];
  return SCLENS_OK;
}

#define CTX_GUARD(h) \
  if (!(h)) return SCLENS_ERR_ARG; \
  hipSetDevice((h)->c.device)

int sclens_hip_wishart_matrix_f32(sclens_hip_ctx* h, const float* X, int64_t N, int64_t M, int dims, float* Y) {
  CTX_GUARD(h);
  return scl::wishart_host(&h->c, X, N, M, dims, Y);
}
int sclens_hip_get_eigen_f32(sclens_hip_ctx* h, const float* Y, int64_t n, float* L, float* V) {
  CTX_GUARD(h);
  return scl::get_eigen_host(&h->c, Y, n, L, V);
}
int sclens_hip_corr_mat_f32(sclens_hip_ctx* h, const float* X, int64_t n, int64_t p, const float* Y, int64_t q, float* out) {
  CTX_GUARD(h);
  return scl::corr_mat_host(&h->c, X, n, p, Y, q, out);
}
int sclens_hip_preprocess_csc(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, const uint8_t* is_mito, const uint8_t* is_ribo, double min_tp_c,
                              double min_tp_g, double max_tp_c, double max_tp_g, int64_t min_genes_per_cell,
                              int64_t max_genes_per_cell, int64_t min_cells_per_gene, double mito_percent,
                              double ribo_percent, uint8_t* keep_cell, int64_t* gene_order, int64_t* n_cells,
                              int64_t* n_genes, int64_t* nnz_out) {
  CTX_GUARD(h);
  const scl::PpParams P{min_tp_c, min_tp_g, max_tp_c, max_tp_g, min_genes_per_cell, max_genes_per_cell, min_cells_per_gene,
                        mito_percent, ribo_percent};
  return scl::preprocess_stats(&h->c, N, M, colptr, rowval, nzval, is_mito, is_ribo, P, keep_cell, gene_order, n_cells,
                               n_genes, nnz_out);
}
int sclens_hip_preprocess_gather(sclens_hip_ctx* h, int64_t* out_colptr, int32_t* out_rowval, float* out_nzval) {
  CTX_GUARD(h);
  return scl::preprocess_gather(&h->c, out_colptr, out_rowval, out_nzval);
}
/* ---- device-resident count matrices (preprocess -> sclens without the host round trip) */
int sclens_hip_preprocess_keep(sclens_hip_ctx* h, sclens_hip_counts** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Counts* c = nullptr;
  const int rc = scl::preprocess_keep(&h->c, &c);
  if (rc == SCLENS_OK) *out = reinterpret_cast<sclens_hip_counts*>(c);
  return rc;
}
int sclens_hip_counts_upload(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                             sclens_hip_counts** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Counts* c = nullptr;
  const int rc = scl::counts_upload(&h->c, N, M, colptr, rowval, nzval, &c);
  if (rc == SCLENS_OK) *out = reinterpret_cast<sclens_hip_counts*>(c);
  return rc;
}
int sclens_hip_counts_info(const sclens_hip_counts* counts, int64_t* N, int64_t* M, int64_t* nnz) {
  if (!counts) return SCLENS_ERR_ARG;
  const scl::Counts* c = reinterpret_cast<const scl::Counts*>(counts);
  if (N) *N = c->N;
  if (M) *M = c->M;
  if (nnz) *nnz = c->nnz;
  return SCLENS_OK;
}
int sclens_hip_counts_download(sclens_hip_ctx* h, const sclens_hip_counts* counts, int64_t* colptr, int32_t* rowval, float* nzval) {
  CTX_GUARD(h);
  if (!counts) return SCLENS_ERR_ARG;
  return scl::counts_download(&h->c, reinterpret_cast<const scl::Counts*>(counts), colptr, rowval, nzval);
}
void sclens_hip_counts_destroy(sclens_hip_counts* counts) {
  if (!counts) return;
  hipSetDevice(reinterpret_cast<scl::Counts*>(counts)->device);
  scl::counts_free(reinterpret_cast<scl::Counts*>(counts));
}
int sclens_hip_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) return SCLENS_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return SCLENS_ERR_NO_DEVICE;
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable) != hipSuccess || !p) {
    (void)hipGetLastError();
    return SCLENS_ERR_OOM;
  }
  *out = p;
  return SCLENS_OK;
}
void sclens_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
int sclens_hip_session_create_from_counts(sclens_hip_ctx* h, const sclens_hip_counts* counts, sclens_hip_session** out) {
  CTX_GUARD(h);
  if (!out || !counts) return SCLENS_ERR_ARG;
  scl::Session* s = nullptr;
  const int rc = scl::session_create_from_counts(&h->c, reinterpret_cast<const scl::Counts*>(counts), &s);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
int sclens_hip_pattern_create_drawn_from_counts(sclens_hip_ctx* h, const sclens_hip_counts* counts, uint64_t seed, sclens_hip_pattern** out,
                                                int64_t* n_cand) {
  CTX_GUARD(h);
  if (!out || !counts) return SCLENS_ERR_ARG;
  scl::PatternOwner* p = nullptr;
  const int rc = scl::pattern_create_drawn_from_counts(&h->c, reinterpret_cast<const scl::Counts*>(counts), seed, &p, n_cand);
  if (rc == SCLENS_OK) *out = reinterpret_cast<sclens_hip_pattern*>(p);
  return rc;
}
int sclens_hip_scale_csc_f32(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                             const float* nzval, int centering, int f32path, float* out, double* rec_tgc, double* rec_mean,
                             double* rec_std, double* rec_norm, double* rec_cent) {
  CTX_GUARD(h);
  scl::ScaleVecs k{rec_tgc, rec_mean, rec_std, rec_norm, rec_cent};
  const bool any = rec_tgc || rec_mean || rec_std || rec_norm || rec_cent;
  if (any && !(rec_tgc && rec_mean && rec_std && rec_norm && rec_cent))
    return h->c.fail(SCLENS_ERR_ARG, "scale_csc: pass all rec_* buffers or none");
  return scl::scale_csc_host(&h->c, N, M, colptr, rowval, nzval, centering, f32path, out, any ? &k : nullptr);
}
int sclens_hip_corr_colmax_f32(sclens_hip_ctx* h, const float* X, int64_t n, int64_t p, const float* Y, int64_t q, int use_split,
                               float* out) {
  CTX_GUARD(h);
  return scl::corr_colmax_host(&h->c, X, n, p, Y, q, use_split, out);
}
int sclens_hip_gram_binary_f32(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                               const float* nzval, int use_bits, float divisor, float* out) {
  CTX_GUARD(h);
  return scl::gram_binary_host(&h->c, N, M, colptr, rowval, nzval, use_bits, divisor, out);
}
int sclens_hip_gram_counts_f32(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                               int mode, int f32path, int binary, float divisor, float* out) {
  CTX_GUARD(h);
  return scl::gram_counts_host(&h->c, N, M, colptr, rowval, nzval, mode, f32path, binary, divisor, out);
}
int sclens_hip_get_eigvec_f32(sclens_hip_ctx* h, const float* X, int64_t N, int64_t M, int64_t keep_top, float* nL,
                              float* nV, int64_t* r) {
  CTX_GUARD(h);
  return scl::get_eigvec_host(&h->c, X, N, M, keep_top, nL, nV, r);
}

int sclens_hip_get_denoised_f32(sclens_hip_ctx* h, const float* pca_n1, int64_t N, int64_t n_sig, const float* gene_basis_sig,
                                int64_t M, const double* tgc, const double* mat2_mean, const double* mat2_std,
                                const double* norm_tgc, const double* cent, float* out) {
  CTX_GUARD(h);
  return scl::denoise_host(&h->c, pca_n1, N, n_sig, gene_basis_sig, M, tgc, mat2_mean, mat2_std, norm_tgc, cent, out);
}

int sclens_hip_session_create(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, int64_t n_cand, const uint32_t* z1, const uint32_t* z2,
                              sclens_hip_session** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Session* s = nullptr;
  int rc = scl::session_create(&h->c, N, M, colptr, rowval, nzval, n_cand, z1, z2, &s);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
int sclens_hip_session_create_sharded(sclens_hip_ctx* h, int64_t N_global, int64_t row0, int64_t N_local, int64_t M,
                                      const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t n_cand,
                                      const uint32_t* z1, const uint32_t* z2, sclens_hip_allreduce_fn allreduce, void* user,
                                      sclens_hip_session** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Session* s = nullptr;
  int rc = scl::session_create_sharded(&h->c, N_global, row0, N_local, M, colptr, rowval, nzval, n_cand, z1, z2, allreduce,
                                       user, &s);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
int sclens_hip_session_clone(sclens_hip_ctx* h, sclens_hip_session* src, sclens_hip_session** out) {
  CTX_GUARD(h);
  if (!src || !src->s || !out) return SCLENS_ERR_ARG;
  if (h->c.device != src->ctx->c.device) return h->c.fail(SCLENS_ERR_ARG, "session_clone: contexts must share the device");
  scl::Session* s = nullptr;
  int rc = scl::session_clone(&h->c, src->s, &s);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
void sclens_hip_session_destroy(sclens_hip_session* w) {
  if (!w) return;
  hipSetDevice(w->ctx->c.device);
  scl::session_destroy(w->s);
  delete w;
}
#define SES_GUARD(w) \
  if (!(w) || !(w)->s) return SCLENS_ERR_ARG; \
  hipSetDevice((w)->ctx->c.device)

int sclens_hip_session_create_sharded_drawn(sclens_hip_ctx* h, int64_t N_global, int64_t row0, int64_t N_local, int64_t M,
                                            const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t nnz_global,
                                            uint64_t seed, sclens_hip_allreduce_fn allreduce, void* user, sclens_hip_session** out,
                                            int64_t* n_cand_local) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Session* s = nullptr;
  const int rc = scl::session_create_sharded_drawn(&h->c, N_global, row0, N_local, M, colptr, rowval, nzval, nnz_global, seed, allreduce,
                                                   user, &s, n_cand_local);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
int sclens_hip_session_create_chunked(sclens_hip_ctx* h, int64_t N_global, int64_t M, int n_chunks, int64_t nnz_global, uint64_t seed,
                                      sclens_hip_session** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::Session* s = nullptr;
  const int rc = scl::session_create_chunked(&h->c, N_global, M, n_chunks, nnz_global, seed, &s);
  if (rc != SCLENS_OK) return rc;
  sclens_hip_session* w = new sclens_hip_session();
  w->s = s;
  w->ctx = h;
  *out = w;
  return SCLENS_OK;
}
int sclens_hip_session_chunk_add(sclens_hip_session* w, int which, int g, int64_t row0, int64_t N_local, const int64_t* colptr,
                                 const int32_t* rowval, const float* nzval) {
  SES_GUARD(w);
  return scl::session_chunk_add(w->s, which, g, row0, N_local, colptr, rowval, nzval);
}
int sclens_hip_session_chunk_commit(sclens_hip_session* w) {
  SES_GUARD(w);
  return scl::session_chunk_commit(w->s);
}
int sclens_hip_session_null_spectrum_chunked(sclens_hip_session* w, double* Lr) {
  SES_GUARD(w);
  return scl::session_null_spectrum_chunked(w->s, Lr);
}
int sclens_hip_session_set_reduce_to(sclens_hip_session* w, sclens_hip_reduce_fn reduce, void* user) {
  SES_GUARD(w);
  return scl::session_set_reduce_to(w->s, reduce, user);
}
int sclens_hip_session_set_candidate_range(sclens_hip_session* w, int64_t cand_off, int64_t n_cand_global) {
  SES_GUARD(w);
  return scl::session_set_candidate_range(w->s, cand_off, n_cand_global);
}
int sclens_hip_session_local_candidates(sclens_hip_session* w, uint32_t* z1, uint32_t* z2) {
  SES_GUARD(w);
  return scl::session_local_candidates(w->s, z1, z2);
}
int sclens_hip_session_search_round_seeded(sclens_hip_session* w, const uint64_t* seeds, const int64_t* m, const int32_t* roots, int count,
                                           int my_slot, int64_t n_2, double* d5, int64_t* r_it) {
  SES_GUARD(w);
  return scl::session_search_round_seeded(w->s, seeds, m, roots, count, my_slot, n_2, d5, r_it);
}
int sclens_hip_session_perturb_round_seeded(sclens_hip_session* w, const int64_t* t, const uint64_t* seeds, const int64_t* m,
                                            const int32_t* roots, int count, int my_slot, int64_t min_pc, double* nL_top, int64_t* ncols) {
  SES_GUARD(w);
  return scl::session_perturb_round_seeded(w->s, t, seeds, m, roots, count, my_slot, min_pc, nL_top, ncols);
}
int sclens_hip_session_set_reducer(sclens_hip_session* w, sclens_hip_allreduce_fn allreduce, void* user) {
  SES_GUARD(w);
  return scl::session_set_reducer(w->s, allreduce, user);
}
int sclens_hip_session_spectrum(sclens_hip_session* w, const int64_t* rc, const int32_t* rr, const float* rv, double* L,
                                double* Lr, double* rec_tgc, double* rec_mean, double* rec_std, double* rec_norm,
                                double* rec_cent) {
  SES_GUARD(w);
  // scale_to_dense copies all five vectors or none: route missing ones to scratch
  scl::ScaleVecs k{rec_tgc, rec_mean, rec_std, rec_norm, rec_cent};
  bool any = rec_tgc || rec_mean || rec_std || rec_norm || rec_cent;
  if (any && !(rec_tgc && rec_mean && rec_std && rec_norm && rec_cent))
    return w->ctx->c.fail(SCLENS_ERR_ARG, "spectrum: pass all rec_* buffers or none");
  return scl::session_spectrum(w->s, rc, rr, rv, L, Lr, any ? &k : nullptr);
}
int sclens_hip_session_null_spectrum(sclens_hip_session* w, const int64_t* rc, const int32_t* rr, const float* rv,
                                     double* Lr) {
  SES_GUARD(w);
  if (!rc || !rr || !rv) return SCLENS_ERR_ARG;
  return scl::session_null_spectrum(w->s, rc, rr, rv, Lr);
}
int sclens_hip_session_null_spectrum_pattern(sclens_hip_session* w, sclens_hip_pattern* p, double* Lr) {
  SES_GUARD(w);
  if (!p) return SCLENS_ERR_ARG;
  return scl::session_null_spectrum_pattern(w->s, reinterpret_cast<scl::PatternOwner*>(p), Lr);
}
int sclens_hip_session_data_spectrum(sclens_hip_session* w, double* L, double* rec_tgc, double* rec_mean, double* rec_std,
                                     double* rec_norm, double* rec_cent) {
  SES_GUARD(w);
  scl::ScaleVecs k{rec_tgc, rec_mean, rec_std, rec_norm, rec_cent};
  bool any = rec_tgc || rec_mean || rec_std || rec_norm || rec_cent;
  if (any && !(rec_tgc && rec_mean && rec_std && rec_norm && rec_cent))
    return w->ctx->c.fail(SCLENS_ERR_ARG, "data_spectrum: pass all rec_* buffers or none");
  return scl::session_data_spectrum(w->s, L, any ? &k : nullptr);
}
int sclens_hip_pattern_create(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, int64_t n_cand, const uint32_t* z1, const uint32_t* z2,
                              sclens_hip_pattern** out) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::PatternOwner* p = nullptr;
  const int rc = scl::pattern_create(&h->c, N, M, colptr, rowval, nzval, n_cand, z1, z2, &p);
  if (rc == SCLENS_OK) *out = reinterpret_cast<sclens_hip_pattern*>(p);
  return rc;
}
int sclens_hip_pattern_create_drawn(sclens_hip_ctx* h, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                                    const float* nzval, uint64_t seed, sclens_hip_pattern** out, int64_t* n_cand) {
  CTX_GUARD(h);
  if (!out) return SCLENS_ERR_ARG;
  scl::PatternOwner* p = nullptr;
  const int rc = scl::pattern_create_drawn(&h->c, N, M, colptr, rowval, nzval, seed, &p, n_cand);
  if (rc == SCLENS_OK) *out = reinterpret_cast<sclens_hip_pattern*>(p);
  return rc;
}
int sclens_hip_pattern_candidates(sclens_hip_ctx* h, sclens_hip_pattern* p, uint32_t* z1, uint32_t* z2) {
  CTX_GUARD(h);
  if (!p) return SCLENS_ERR_ARG;
  return scl::pattern_candidates(&h->c, reinterpret_cast<scl::PatternOwner*>(p), z1, z2);
}
int sclens_hip_pattern_download(sclens_hip_ctx* h, sclens_hip_pattern* p, int which, void* dst) {
  CTX_GUARD(h);
  if (!p) return SCLENS_ERR_ARG;
  return scl::pattern_download(&h->c, reinterpret_cast<scl::PatternOwner*>(p), which, dst);
}
void sclens_hip_pattern_destroy(sclens_hip_pattern* p) { scl::pattern_destroy(reinterpret_cast<scl::PatternOwner*>(p)); }
int sclens_hip_session_set_pattern(sclens_hip_session* w, sclens_hip_pattern* p) {
  SES_GUARD(w);
  if (!p) return SCLENS_ERR_ARG;
  return scl::session_set_pattern(w->s, reinterpret_cast<scl::PatternOwner*>(p));
}
int sclens_hip_session_shared_buffer(sclens_hip_session* w, int what, int64_t rows, int64_t k, double* theta0, void** dev_ptr,
                                     int64_t* rows_out, int64_t* k_out, int64_t* ld) {
  SES_GUARD(w);
  return scl::session_shared_buffer(w->s, what, rows, k, theta0, dev_ptr, rows_out, k_out, ld);
}
int sclens_hip_session_adopt(sclens_hip_session* dst, sclens_hip_session* src, int what) {
  SES_GUARD(dst);
  if (!src || !src->s) return SCLENS_ERR_ARG;
  return scl::session_adopt(dst->s, src->s, what);
}
int sclens_hip_session_refine_eigenvalues(sclens_hip_session* w, int64_t idx_lo, int64_t idx_hi, double* rho) {
  SES_GUARD(w);
  return scl::session_refine_eigenvalues(w->s, idx_lo, idx_hi, rho);
}
int sclens_hip_session_signal_vectors(sclens_hip_session* w, int64_t k, float* nV) {
  SES_GUARD(w);
  return scl::session_signal_vectors(w->s, k, nV);
}
int sclens_hip_session_binary_basis(sclens_hip_session* w, double* L_bin, int64_t* r) {
  SES_GUARD(w);
  return scl::session_binary_basis(w->s, L_bin, r);
}
int sclens_hip_session_search_step(sclens_hip_session* w, const uint32_t* sample, int64_t m, int64_t n_2, double* d5,
                                   int64_t* r_it) {
  SES_GUARD(w);
  if (!d5 || (m > 0 && !sample)) return SCLENS_ERR_ARG;
  return scl::session_search_step(w->s, sample, m, n_2, d5, r_it);
}
int sclens_hip_session_perturb(sclens_hip_session* w, int64_t t, const uint32_t* sample, int64_t m, int64_t min_pc,
                               double* nL_top, int64_t* ncols) {
  SES_GUARD(w);
  if (!nL_top || (m > 0 && !sample)) return SCLENS_ERR_ARG;
  return scl::session_perturb(w->s, t, sample, m, min_pc, nL_top, ncols);
}
int sclens_hip_session_search_step_seeded(sclens_hip_session* w, uint64_t seed, int64_t m, int64_t n_2, double* d5,
                                          int64_t* r_it) {
  SES_GUARD(w);
  if (!d5) return SCLENS_ERR_ARG;
  return scl::session_search_step_seeded(w->s, seed, m, n_2, d5, r_it);
}
int sclens_hip_session_perturb_seeded(sclens_hip_session* w, int64_t t, uint64_t seed, int64_t m, int64_t min_pc,
                                      double* nL_top, int64_t* ncols) {
  SES_GUARD(w);
  if (!nL_top) return SCLENS_ERR_ARG;
  return scl::session_perturb_seeded(w->s, t, seed, m, min_pc, nL_top, ncols);
}
int sclens_hip_session_get_perturbed(sclens_hip_session* w, int64_t t, float* out) {
  SES_GUARD(w);
  if (!out) return SCLENS_ERR_ARG;
  return scl::session_get_perturbed(w->s, t, out);
}
int sclens_hip_session_robustness(sclens_hip_session* w, int64_t P, int32_t* a_b, double* b) {
  SES_GUARD(w);
  if (!a_b || !b) return SCLENS_ERR_ARG;
  return scl::session_robustness(w->s, P, a_b, b);
}
int sclens_hip_session_gene_basis(sclens_hip_session* w, const double* nL, float* out) {
  SES_GUARD(w);
  if (!nL || !out) return SCLENS_ERR_ARG;
  return scl::session_gene_basis(w->s, nL, out);
}

int sclens_hip_session_set_int(sclens_hip_session* w, const char* name, int64_t value) {
  SES_GUARD(w);
  return scl::session_set_int(w->s, name, value);
}
int sclens_hip_session_get_int(sclens_hip_session* w, const char* name, int64_t* value) {
  SES_GUARD(w);
  if (!value) return SCLENS_ERR_ARG;
  return scl::session_get_int(w->s, name, value);
}
int64_t sclens_hip_session_slot_ld(sclens_hip_session* w) { return (w && w->s) ? scl::session_slot_ld(w->s) : -1; }
int sclens_hip_session_export_slot(sclens_hip_session* w, int64_t t, int64_t min_pc, void* dst) {
  SES_GUARD(w);
  return scl::session_export_slot(w->s, t, min_pc, dst);
}
int sclens_hip_session_import_slot(sclens_hip_session* w, int64_t t, int64_t min_pc, int64_t ncols, const void* src) {
  SES_GUARD(w);
  return scl::session_import_slot(w->s, t, min_pc, ncols, src);
}

int sclens_hip_symv_probe(sclens_hip_ctx* h, int64_t n, int64_t* launches, double* total_ms, double* total_bytes) {
  CTX_GUARD(h);
  if (!launches || !total_ms || !total_bytes) return SCLENS_ERR_ARG;
  return scl::symv_probe(&h->c, n, launches, total_ms, total_bytes);
}
int sclens_hip_symv_profile(sclens_hip_ctx* h, int enable) {
  CTX_GUARD(h);
  h->c.prof_symv = enable != 0;
  h->c.prof_used = 0;
  h->c.prof_bytes = 0.0;
  return SCLENS_OK;
}
int sclens_hip_symv_profile_read(sclens_hip_ctx* h, int64_t* launches, double* total_ms, double* total_bytes) {
  CTX_GUARD(h);
  SCL_HIP(&h->c, hipStreamSynchronize(h->c.stream));
  double ms = 0.0;
  for (size_t q = 0; q + 1 < h->c.prof_used; q += 2) {
    float t = 0.f;
    SCL_HIP(&h->c, hipEventElapsedTime(&t, h->c.prof_ev[q], h->c.prof_ev[q + 1]));
    ms += (double)t;
  }
  if (launches) *launches = (int64_t)(h->c.prof_used / 2);
  if (total_ms) *total_ms = ms;
  if (total_bytes) *total_bytes = h->c.prof_bytes;
  h->c.prof_used = 0;
  h->c.prof_bytes = 0.0;
  return SCLENS_OK;
}

// ---- device-level entry points
int sclens_hip_dev_gemm_f32(sclens_hip_ctx* h, const float* P, const float* Q, float* C, int64_t M, int64_t N, int64_t K,
                            int64_t ldp, int64_t ldq, int64_t ldc, float alpha, float beta, int q_kcontig, int lower,
                            uint32_t* colabsmax) {
  CTX_GUARD(h);
  scl::GemmArgs g{P, Q, C, M, N, K, ldp, ldq, ldc, alpha, beta, q_kcontig, lower, colabsmax};
  return scl::gemm_f32(&h->c, g);
}
int sclens_hip_dev_gram_f32(sclens_hip_ctx* h, const float* B, int64_t n, int64_t K, int64_t ldb, float divisor, float* A,
                            int64_t lda) {
  CTX_GUARD(h);
  return scl::gram_f32(&h->c, B, n, K, ldb, divisor, A, lda);
}
int sclens_hip_dev_sy2sb_f32(sclens_hip_ctx* h, float* A, int64_t n, int64_t lda, float* T, int* breakdown) {
  CTX_GUARD(h);
  return scl::sy2sb_f32(&h->c, A, n, lda, T, breakdown);
}
int sclens_hip_dev_sbr_apply_q1_f32(sclens_hip_ctx* h, const float* A, int64_t n, int64_t lda, const float* T, float* Zt, int64_t m,
                                    int64_t ldz) {
  CTX_GUARD(h);
  return scl::sbr_apply_q1(&h->c, A, n, lda, T, Zt, m, ldz);
}
int sclens_hip_dev_sbr_apply_q2_f32(sclens_hip_ctx* h, int64_t n, float* Zt, int64_t m, int64_t ldz) {
  CTX_GUARD(h);
  return scl::sbr_apply_q2(&h->c, n, Zt, m, ldz);
}
int sclens_hip_dev_sb2st_f32(sclens_hip_ctx* h, const float* A, int64_t n, int64_t lda, double* d, double* e) {
  CTX_GUARD(h);
  return scl::sb2st_f32(&h->c, A, n, lda, d, e);
}
int sclens_hip_dev_sytrd_f32(sclens_hip_ctx* h, float* A, int64_t n, int64_t lda, double* d, double* e, float* tau) {
  CTX_GUARD(h);
  return scl::sytrd_f32(&h->c, A, n, lda, d, e, tau);
}
int sclens_hip_dev_stebz_f64(sclens_hip_ctx* h, const double* d, const double* e, int64_t n, double* w) {
  CTX_GUARD(h);
  return scl::stebz_f64(&h->c, d, e, n, w);
}
int sclens_hip_dev_eigh_f32(sclens_hip_ctx* h, float* A, int64_t n, int64_t lda, double* w, int64_t vec_lo, int64_t vec_hi,
                            float* Zt, int64_t ldz) {
  CTX_GUARD(h);
  return scl::eigh_f32(&h->c, A, n, lda, w, vec_lo, vec_hi, Zt, ldz);
}
int sclens_hip_trim(int device_id) {
  scl::stein_shared_release(device_id, nullptr);  // the device-wide inverse-iteration block is idle between its users' kernels
  scl::pool_trim(device_id);
  return SCLENS_OK;
}
int sclens_hip_release_scratch(sclens_hip_ctx* h, const char* family) {
  CTX_GUARD(h);
  if (!family) return SCLENS_ERR_ARG;
  const std::string f(family);
  std::vector<std::string> pre;
  if (f == "eigensolver" || f == "all") for (const char* q : {"sbr.", "stein.", "tri.", "trd.", "orm.", "eig."}) pre.push_back(q);
  if (f == "gram" || f == "all") for (const char* q : {"gram.", "gb."}) pre.push_back(q);
  if (f == "chefsi" || f == "all") pre.push_back("che.");
  if (f == "corr" || f == "all") for (const char* q : {"c.", "ses.Vr2h", "ses.Zh", "ses.cmax"}) pre.push_back(q);  // the search statistic's images
  scl::Ctx& c = h->c;
  if (f == "everything") {  // every named workspace of the context: only between calls (no live session holds pointers into them)
    if (c.live_sessions > 0) return c.fail(SCLENS_ERR_STATE, "release_scratch(everything): a session of this context is still alive");
    c.release_all();
    scl::stein_shared_release(c.device, c.stream);
    c.ws_epoch += 1;
    c.q2_tg_n = -1; c.q2_built_variant = -1; c.q1p_n = -1; c.last_two_stage = false;
    return SCLENS_OK;
  }
  if (pre.empty()) return c.fail(SCLENS_ERR_ARG, "release_scratch: family must be eigensolver, gram, chefsi, corr, all or everything");
  if (c.opt.debug >= 2) {  // what this context's named workspaces hold at the moment a family goes back (footprint table of INTEGRATION.md 6)
    std::vector<std::pair<size_t, std::string>> tab;
    size_t tot = 0;
    for (const auto& kv : c.ws) {
      tab.emplace_back(kv.second.second, kv.first);
      tot += kv.second.second;
    }
    std::sort(tab.rbegin(), tab.rend());
    fprintf(stderr, "[workspaces] context %p before release_scratch(%s): %.2f GB in %zu blocks\n", (void*)h, family, tot / 1e9, tab.size());
    for (const auto& t : tab)
      if (t.first >= ((size_t)64 << 20)) fprintf(stderr, "[workspaces]   %-16s %8.2f GB\n", t.second.c_str(), t.first / 1e9);
  }
  std::vector<std::string> names;
  for (const auto& kv : c.ws)
    for (const std::string& q : pre)
      if (kv.first.compare(0, q.size(), q) == 0) names.push_back(kv.first);
  if (f == "eigensolver" || f == "all") scl::stein_shared_release(c.device, c.stream);
  if (names.empty()) return SCLENS_OK;
  scl::ctx_quiesce(&c);
  for (const std::string& nm : names) c.release(nm);
  if (f == "eigensolver" || f == "all") {  // nothing a later call could continue from: the next decomposition starts from scratch
    c.q2_tg_n = -1;
    c.q2_built_variant = -1;
    c.q1p_n = -1;
    c.last_two_stage = false;
  }
  return SCLENS_OK;
}
int64_t sclens_hip_pool_peak(int device_id, int reset) { return (int64_t)scl::pool_peak(device_id, reset != 0); }
int sclens_hip_pool_set_cap(int device_id, int64_t bytes) {
  scl::pool_set_cap(device_id, (long long)bytes);
  return SCLENS_OK;
}
int sclens_hip_pool_stats(int device_id, int64_t* cached_bytes, int64_t* live_bytes, int64_t* hits, int64_t* misses) {
  size_t c = 0, l = 0, h = 0, m = 0;
  scl::pool_stats(device_id, &c, &l, &h, &m);
  if (cached_bytes) *cached_bytes = (int64_t)c;
  if (live_bytes) *live_bytes = (int64_t)l;
  if (hits) *hits = (int64_t)h;
  if (misses) *misses = (int64_t)m;
  return SCLENS_OK;
}
void* sclens_hip_dev_malloc(sclens_hip_ctx* h, int64_t bytes) {
  if (!h || bytes < 0) return nullptr;
  hipSetDevice(h->c.device);
  void* p = nullptr;
  if (scl::pool_malloc(&p, bytes > 0 ? (size_t)bytes : 16) != hipSuccess) {
    h->c.fail(SCLENS_ERR_OOM, "dev_malloc failed");
    return nullptr;
  }
  return p;
}
void sclens_hip_dev_free(sclens_hip_ctx* h, void* p) {
  if (!h || !p) return;
  hipSetDevice(h->c.device);
  scl::pool_free(p, h->c.stream);
}
int sclens_hip_dev_memcpy(sclens_hip_ctx* h, void* dst, const void* src, int64_t bytes, int kind) {
  CTX_GUARD(h);
  hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  SCL_HIP(&h->c, hipMemcpyAsync(dst, src, (size_t)bytes, k, h->c.stream));
  SCL_HIP(&h->c, hipStreamSynchronize(h->c.stream));
  return SCLENS_OK;
}
int sclens_hip_dev_memset(sclens_hip_ctx* h, void* dst, int value, int64_t bytes) {
  CTX_GUARD(h);
  SCL_HIP(&h->c, hipMemsetAsync(dst, value, (size_t)bytes, h->c.stream));
  return SCLENS_OK;
}
int sclens_hip_dev_sync(sclens_hip_ctx* h) {
  CTX_GUARD(h);
  SCL_HIP(&h->c, hipStreamSynchronize(h->c.stream));
  return SCLENS_OK;
}

}  // extern "C"
