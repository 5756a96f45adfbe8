/* sclens_hip.h -- C ABI of libsclens_hip.so: the MI355X (gfx950) replacement for the device side of
 * scLENS.sclens() (reference: Mathbiomed/scLENS v2.0.1, src/scLENS.jl).
 *
 * The reference dispatches on a string kwarg `device` inside five functions and three inline sites
 * (scLENS.jl:332, :363, :375, :489, :526, :505, :558-561, :813-816); every GPU call there uploads a
 * host matrix, runs one cuBLAS/cuSOLVER routine and downloads the result. This library offers
 *   (A) the same five operations with the same host-array semantics (drop-in per call site), and
 *   (B) a session that keeps the count matrix, the scaled matrices, the Gram matrix and the
 *       eigenvectors resident in HBM for the whole sclens() call (what a `device_="hip"` branch of
 *       sclens() would call; see INTEGRATION.md for the Julia ccall shim).
 *
 * Conventions: plain pointers and sizes, no C++/torch types. Host matrices are COLUMN-MAJOR (Julia
 * layout) unless stated. Sparse input is CSC with 0-BASED int64 column pointers and int32 row indices
 * (Julia's SparseMatrixCSC{Float32,UInt32} minus one, scLENS.jl:103-117). All functions return 0 on
 * success or an SCLENS_ERR_* code and never throw; sclens_hip_last_error() gives the message.
 * Blocking calls; one host thread per context. The caller owns every host buffer; the library owns
 * device memory behind the opaque handles.
 */
#ifndef SCLENS_HIP_H
#define SCLENS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCLENS_OK 0
#define SCLENS_ERR_ARG 1       /* bad argument */
#define SCLENS_ERR_NO_DEVICE 2 /* no HIP device / device init failed (shim falls back to "cpu") */
#define SCLENS_ERR_OOM 3       /* device out of memory (reference: catch -> CPU, scLENS.jl:504-508) */
#define SCLENS_ERR_HIP 4       /* other HIP runtime error */
#define SCLENS_ERR_NOCONV 5    /* eigensolver did not converge */
#define SCLENS_ERR_NAN 6       /* NaN eigenvalue (reference: redo in Float64 on CPU, scLENS.jl:379-381) */
#define SCLENS_ERR_STATE 7     /* session call out of order */

typedef struct sclens_hip_ctx sclens_hip_ctx;
typedef struct sclens_hip_session sclens_hip_session;

/* ---------------------------------------------------------------- context ---------------------- */
int sclens_hip_create(sclens_hip_ctx** ctx, int device_id);
void sclens_hip_destroy(sclens_hip_ctx* ctx);
const char* sclens_hip_last_error(const sclens_hip_ctx* ctx);
const char* sclens_hip_version(void);
/* per-stage HIP-event timing ("scale","gram","sytrd","stebz","stein","ormtr","chefsi","corr","recover") */
int sclens_hip_set_timing(sclens_hip_ctx* ctx, int enabled);
int sclens_hip_get_timing(sclens_hip_ctx* ctx, const char* stage, double* total_ms, int64_t* calls);
int sclens_hip_reset_timing(sclens_hip_ctx* ctx);
/* Per-launch HIP-event timing of the dominant kernel (trd_colB, the HBM-bound symmetric matrix-vector product of
 * the tridiagonalisation). enable=1 starts recording; sclens_hip_symv_profile_read() synchronises, returns the
 * number of recorded launches, the sum of their durations and of their algorithmic bytes (the lower triangle of the
 * symmetric trailing matrix, 2 n'(n'+1) with n' = n-j-1, each), and
 * clears the record. */
int sclens_hip_symv_profile(sclens_hip_ctx* ctx, int enable);
int sclens_hip_symv_profile_read(sclens_hip_ctx* ctx, int64_t* launches, double* total_ms, double* total_bytes);
/* Roofline probe: all n-1 trd_colB launches of one tridiagonalisation of order n, back to back between one pair of HIP
 * events on the context's stream (synthetic finite data; same grids and arguments as the real reduction). */
int sclens_hip_symv_probe(sclens_hip_ctx* ctx, int64_t n, int64_t* launches, double* total_ms, double* total_bytes);
/* Context options: every tunable of the library is a named integer of the context (table with defaults: csrc/common.h,
 * SCL_OPTION_TABLE; INTEGRATION.md section 3 lists them). A new context starts from the defaults; the environment variable
 * SCLENS_HIP_OPTIONS ("name=value,name=value") is applied once, inside sclens_hip_create -- nothing is read from the environment per
 * call. Unknown names return SCLENS_ERR_ARG. The ones a host touches:
 *   "precision"  (1 / 0) what the reference selects with `device_` (scLENS.jl:649: "gpu" = Float32 cuBLAS / cuSOLVER, :335-343, :377):
 *                0 = every dense product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), the reference GPU path's
 *                arithmetic; the Gram matrix of a BINARISED matrix (scLENS.jl:735-738) is still formed as the exact co-occurrence product
 *                of its 0/1 pattern (exact in fp16) with the cell weights as three fp16 pieces = 33 bits -- no operand narrower than fp32
 *                ("gram_bits_strict" = 0 sends it through the fp32 product as well);
 *                1 (default) = the large products (Gram, search statistic, trailing updates and W of the band reduction, both
 *                back-transformations) from operands split into two fp16 pieces on the fp16 matrix cores: 22-bit operands, fp32
 *                accumulation, measured as accurate as the fp32 products they replace;
 *   "two_stage"  (-1 / 0 / 1) which reduction the eigensolver that replaces cuSOLVER syevd! (scLENS.jl:377) uses: 1 = two-stage (dense ->
 *                band of half-width 64 -> tridiagonal by bulge chasing, sbr.hip; falls back to the one-stage reduction for orders below
 *                128 and when a panel is numerically rank deficient), 0 = one-stage (tridiag.hip), -1 = two-stage from "two_stage_min_n"
 *                (8 192) upwards;
 *   "gram_bits"  (-1 / 0 / 1) the Gram matrices of the binarised search matrices as exact co-occurrence products and the search statistic
 *                from split operands (gram_bits.hip); -1 = from "gram_bits_min_n" (16 000); "gram_bits_terms" 2 / 3 = 22 / 33 bits of the
 *                cell weights.
 * sclens_hip_copy_options hands one context's settings to another (the worker contexts a host creates for concurrent decompositions;
 * sclens_hip_session_clone does it for the worker it is given). */
int sclens_hip_set_option(sclens_hip_ctx* ctx, const char* name, int64_t value);
int sclens_hip_get_option(sclens_hip_ctx* ctx, const char* name, int64_t* value);
int sclens_hip_copy_options(sclens_hip_ctx* dst, const sclens_hip_ctx* src);
/* Device memory of the library is pooled per device: the sessions, worker sessions and contexts of successive sclens() calls ask for
 * the same block sizes, so freed blocks are kept and handed out again instead of going through hipFree / hipMalloc (about 2 s per
 * call at 100 000 x 30 000). The cache is for back-to-back calls only: sclens_hip_trim gives the idle blocks back to the driver
 * (device_id < 0: every device) and a host calls it when sclens() returns unless it is about to call again (api.sclens(keep_warm=...),
 * the Julia shim's `keep_warm` keyword). Idle bytes kept per device: what was free on the device when the pool first looked, less an
 * eighth of the device for everybody else; sclens_hip_pool_set_cap overrides it (bytes < 0: back to that rule) -- a host that puts
 * several ranks on one device gives each its share. Environment (process level, read once): SCLENS_HIP_POOL=0 disables pooling,
 * SCLENS_HIP_POOL_MAX_GB = the cap. With the option stein_shared = 1 the inverse iteration's workspaces are one block per device outside
 * any context; sclens_hip_trim (and release_scratch "eigensolver" / "all" / "everything") waits for its last user and frees it too. */
int sclens_hip_trim(int device_id);
int sclens_hip_pool_set_cap(int device_id, int64_t bytes);
/* the largest number of bytes the library held at once on the device (live blocks, the idle cache not counted) since the last reset */
int64_t sclens_hip_pool_peak(int device_id, int reset);
/* A context keeps its scratch (grow-only named workspaces) until it is destroyed; between the phases of one sclens() call most of it is
 * idle -- the eigensolver's 30-40 GB per context during the ensemble, the Gram images after the first decompositions. This hands one
 * family of a context's scratch back to the pool ("eigensolver", "gram", "chefsi", "corr" -- the search statistic's images -- or "all" of
 * these), where the next phase's requests of the same sizes find it; the next call that needs it allocates it again. Not while a
 * session of the context is between session_spectrum / eig values and the vectors that continue from them. "everything": every named
 * workspace of the context (vector blocks and ensemble slots included) -- only when no session of the context is alive (SCLENS_ERR_STATE
 * otherwise): what a host does before sclens_hip_trim when its context outlives the call. */
int sclens_hip_release_scratch(sclens_hip_ctx* ctx, const char* family);
int sclens_hip_pool_stats(int device_id, int64_t* cached_bytes, int64_t* live_bytes, int64_t* hits, int64_t* misses);
/* raw stream handle (hipStream_t) so a host framework can order its own work after ours */
void* sclens_hip_stream(sclens_hip_ctx* ctx);

/* ---------------------------------------------------------------- (A) per-call drop-ins -------- */
/* _wishart_matrix(X; device, dims)  (scLENS.jl:332-361): X is N x M; dims=2 -> X'X / M (M x M),
 * dims=1 -> XX' / M (N x N). Both divide by size(X,2). Y is caller-allocated. */
int sclens_hip_wishart_matrix_f32(sclens_hip_ctx* ctx, const float* X, int64_t N, int64_t M, int dims, float* Y);
/* _get_eigen(Y; device)  (scLENS.jl:375-387): all eigenvalues ascending in L[n], eigenvectors as the
 * columns of V (n x n). Returns SCLENS_ERR_NAN if an eigenvalue is NaN.
 * Accuracy: that of an fp32 solver, ~sqrt(n) eps32 |lambda|max on the eigenvalues, at any norm of Y: from n = 8 192 the two-stage
 * reduction forms its trailing updates from operands split into two fp16 pieces, the reflector columns and the columns that
 * scale with Y under separate power-of-two scales (round 4; tests/test_gpu_sbr.py at norms 1, 2^14, 2^20).
 * The context option precision = 0 keeps those products on the fp32 matrix cores. */
int sclens_hip_get_eigen_f32(sclens_hip_ctx* ctx, const float* Y, int64_t n, float* L, float* V);
/* corr_mat(X, Y; device)  (scLENS.jl:363-373): out = X' * Y, X is n x p, Y is n x q, out is p x q. */
int sclens_hip_corr_mat_f32(sclens_hip_ctx* ctx, const float* X, int64_t n, int64_t p, const float* Y, int64_t q,
                            float* out);
/* preprocess(tmp_df; min_tp_c, min_tp_g, max_tp_c, max_tp_g, min_genes_per_cell, max_genes_per_cell, min_cells_per_gene,
 * mito_percent, ribo_percent)  (scLENS.jl:160-236), the QC filter that precedes sclens(), on a raw count
 * matrix given as 0-based CSC (N cells x M genes). is_mito / is_ribo: per-gene flags (the reference's regexes
 * r"^(?i)mt-." / r"^(?i)RP[SL]." evaluated by the caller on the gene names; NULL = none). Call 1 computes the masks:
 * keep_cell[N] (fc_idx), gene_order[0..*n_genes) = original indices of the kept genes in OUTPUT order (all-zero genes
 * dropped, sorted by mean count ascending, ties in original order), and the sizes of the filtered matrix.
 * *n_cells == 0 or *n_genes == 0 is the reference's "no high quality cells and genes" (returns nothing).
 * Call 2 writes the filtered matrix (n_cells x n_genes CSC, cells renumbered in order) into caller-allocated arrays
 * out_colptr[n_genes+1], out_rowval[nnz_out], out_nzval[nnz_out]; SCLENS_ERR_STATE without a preceding call 1. */
int sclens_hip_preprocess_csc(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, const uint8_t* is_mito, const uint8_t* is_ribo, double min_tp_c,
                              double min_tp_g, double max_tp_c, double max_tp_g, int64_t min_genes_per_cell,
                              int64_t max_genes_per_cell, int64_t min_cells_per_gene, double mito_percent,
                              double ribo_percent, uint8_t* keep_cell, int64_t* gene_order, int64_t* n_cells,
                              int64_t* n_genes, int64_t* nnz_out);
int sclens_hip_preprocess_gather(sclens_hip_ctx* ctx, int64_t* out_colptr, int32_t* out_rowval, float* out_nzval);
/* Call 2': the filtered matrix STAYS IN HBM as a count-matrix handle (what df2sparr(pre_df) is to the reference, scLENS.jl:90-120,
 * :662) instead of travelling to the host and back: sessions and patterns are built from it in place
 * (sclens_hip_session_create_from_counts, sclens_hip_pattern_create_drawn_from_counts). sclens_hip_counts_upload makes the same
 * handle from host arrays; sclens_hip_counts_download copies any of the three arrays back (NULL = skip), e.g. colptr + nzval for
 * a host that draws the null matrix itself (sclens_draw_null_matrix needs no row indices). A handle may be used by every
 * context of its device; destroy it after the sessions / pattern builds that read it. */
typedef struct sclens_hip_counts sclens_hip_counts;
int sclens_hip_preprocess_keep(sclens_hip_ctx* ctx, sclens_hip_counts** out);
int sclens_hip_counts_upload(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                             const float* nzval, sclens_hip_counts** out);
int sclens_hip_counts_info(const sclens_hip_counts* counts, int64_t* N, int64_t* M, int64_t* nnz);
int sclens_hip_counts_download(sclens_hip_ctx* ctx, const sclens_hip_counts* counts, int64_t* colptr, int32_t* rowval, float* nzval);
void sclens_hip_counts_destroy(sclens_hip_counts* counts);
/* Page-locked host memory for arrays the host fills and the library uploads (the null matrix drawn by sclens_draw_null_matrix: 1.1 GB of
 * row indices and values at 100 000 x 30 000, which cross PCIe at a third of the link's rate out of pageable memory). No reference
 * counterpart (the reference uploads dense Float32 matrices with CuArray(), scLENS.jl:346-352). SCLENS_ERR_NO_DEVICE without a HIP device. */
int sclens_hip_host_alloc(int64_t bytes, void** out);
void sclens_hip_host_free(void* p);

/* logn_scale(pre_scale(x))  (scLENS.jl:650-654: proj_l :607 -> log1p -> zscore_with_l2 :596-605 -> scaled_gdata "cent"
 * :300-305 for centering="mean"; scaled_gdata "median" :291-298 -> norm_l :608 for centering="median") and, with
 * f32path = 0 and the five rec_* vectors, the inline Float64 twin of the data matrix (scLENS.jl:676-696).
 * Input: N x M counts as 0-based CSC (colptr[M+1], rowval, nzval). centering: 0 mean, 1 median. f32path: 1 = the closure
 * (Float32 proj_l / log1p / std), 0 = the inline Float64 statistics. out: dense N x M, column-major, fp32.
 * rec_*: all five (TGC[N], mat2_mean[M], mat2_std[M], norm_tgc[N], cent_[M]) or all NULL; must be NULL for median. */
int sclens_hip_scale_csc_f32(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                             const float* nzval, int centering, int f32path, float* out, double* rec_tgc,
                             double* rec_mat2_mean, double* rec_mat2_std, double* rec_norm_tgc, double* rec_cent);
/* Piece exposed for unit tests: out[j] = max_i |X_i' Y_j| for X n x p and Y n x q (column-major), the statistic of the
 * sparsity search (scLENS.jl:742 `maximum(abs.(corr_mat(...)), dims=1)`). use_split = 0: fp32 product; 1: operands split into
 * two fp16 pieces (22 significant bits), three fp16 MFMA products with fp32 accumulation (gram_bits.hip); the split scales by
 * 2^12, so entries must stay below 15 in magnitude (the search passes unit vectors). */
int sclens_hip_corr_colmax_f32(sclens_hip_ctx* ctx, const float* X, int64_t n, int64_t p, const float* Y, int64_t q, int use_split,
                               float* out);
/* Piece exposed for unit tests: the M x M Gram matrix (row-major, fp32) of logn_scale(pre_scale(P)) / divisor for the BINARISED
 * counts P (every stored value counts as 1; scLENS.jl:664), N > M. use_bits = 0: scaled matrix + fp32 product (the general
 * path); 1: weighted co-occurrence product on the fp16 MFMA without forming the scaled matrix (gram_bits.hip), which the
 * sparsity search uses for large problems (context option "gram_bits"). */
int sclens_hip_gram_binary_f32(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                               const float* nzval, int use_bits, float divisor, float* out);
/* Piece exposed for unit tests and for the A/B of SURVEY 8f-1: the M x M Gram matrix (row-major, fp32) of logn_scale(pre_scale(X)) /
 * divisor (f32path = 1; scLENS.jl:650-652) or of the inline Float64 twin (f32path = 0; :676-696) for a COUNT-VALUED N x M matrix, N >= M
 * (binary != 0: stored counts as ones). mode 0: scaled dense matrix + the dense product the context's precision selects (the general
 * path); mode 1: from the sparse structure of the scaled matrix, sparse + rank two -- the identity of scLENS.jl:601-603 applied to the
 * Gram product: fp32 products of the stored entries accumulated in 64-bit fixed point per 128 x 128 tile of gene pairs, rank-two terms
 * in fp64 (gram_sparse.hip; context option "gram_sparse" selects it inside sessions). */
int sclens_hip_gram_counts_f32(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                               const float* nzval, int mode, int f32path, int binary, float divisor, float* out);
/* get_eigvec(X; device)  (scLENS.jl:489-524): X is N x M scaled data. nL[r] positive eigenvalues
 * descending, nV is N x r (cell-side eigenvectors, unit columns). On input *r = capacity in columns
 * (min(N,M) always suffices); on output the number of positive eigenvalues. keep_top > 0 limits the
 * eigenvectors to the first keep_top columns (what the caller consumes, scLENS.jl:776). */
int sclens_hip_get_eigvec_f32(sclens_hip_ctx* ctx, const float* X, int64_t N, int64_t M, int64_t keep_top,
                              float* nL, float* nV, int64_t* r);

/* get_denoised_df(inp_obj; device_)  (scLENS.jl:889-931): pca_n1 is N x n_sig, gene_basis_sig =
 * gene_basis[sig_id,:] as n_sig rows of M genes (row-major, what sclens_hip_session_gene_basis writes), the five vectors
 * are rec_vals. out is N x M (cells x genes), the denoised count means (the DataFrame body of :927). */
int sclens_hip_get_denoised_f32(sclens_hip_ctx* ctx, const float* pca_n1, int64_t N, int64_t n_sig, const float* gene_basis_sig,
                                int64_t M, const double* tgc, const double* mat2_mean, const double* mat2_std,
                                const double* norm_tgc, const double* cent, float* out);

/* ---------------------------------------------------------------- host statistics (no GPU) ----- */
/* _mp_calculation(L, Lr)  (scLENS.jl:424-459). L_mp_mask[n] gets 1 where b_minus < L < b_plus. */
int sclens_mp_calculation(const double* L, int64_t n, const double* Lr, int64_t nr, double* b_plus, double* b_minus,
                          uint8_t* L_mp_mask);
/* _tw(L, L_mp)  (scLENS.jl:461-467): lambda_c, gamma, p, sigma. n_all = length(L). */
int sclens_tw(int64_t n_all, const double* L_mp, int64_t n_mp, double* lambda_c, double* gamma, double* p,
              double* sigma);
/* mp_check(test_L)  (scLENS.jl:469-487) */
int sclens_mp_check(const double* L_mp, int64_t n_mp, double p_val, double* ks_static, int* pass);
/* robustness statistics from b_ (k x npairs, row-major)  (scLENS.jl:797-806): Tukey fence, median, std */
int sclens_robust_scores(const double* b, int64_t k, int64_t npairs, double* m_score, double* sd_score);
/* expected max |N(0,1/n)| over n draws, the quantity scLENS.jl:709-712 estimates with 5000 trials */
double sclens_noise_baseline_exact(int64_t n);

/* ---------------------------------------------------------------- random draws (host, no GPU) -- */
/* R1, scLENS.jl:668-673: nnz uniform (i,j) draws minus the stored entries, de-duplicated in first-occurrence order.
 * z1/z2 need capacity nnz; *count receives the number of candidates. 0-based. */
int sclens_draw_zero_candidates(int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, uint64_t seed,
                                uint32_t* z_idx1, uint32_t* z_idx2, int64_t* count);
/* R2, scLENS.jl:701 (random_nz, rmix=true): stored values shuffled globally, each gene keeps its number of entries at
 * uniformly drawn distinct cells (rows ascending). Same colptr as the input. */
int sclens_draw_null_matrix(int64_t N, int64_t M, const int64_t* colptr, const float* nzval, uint64_t seed,
                            int32_t* out_rowval, float* out_nzval);
/* R4/R5, scLENS.jl:731, :772: m distinct indices of [0, len) -- the same keyed permutation the *_seeded session calls
 * evaluate on the device, so host and device samples are identical for equal (len, m, seed). */
int sclens_sample_without_replacement(uint64_t len, int64_t m, uint64_t seed, uint32_t* out);

/* ---------------------------------------------------------------- (B) device-resident session -- */
/* all-reduce supplied by the host for row-sharded sessions (see sclens_hip_session_create_sharded) */
typedef int (*sclens_hip_allreduce_fn)(void* user, void* dev_ptr, int64_t count, int dtype /*0 fp64, 1 fp32*/);
/* optional companion: the sum lands on rank `root` only (the other ranks' buffers are then undefined); see
 * sclens_hip_session_set_reduce_to and the *_round calls of a row-sharded session */
typedef int (*sclens_hip_reduce_fn)(void* user, void* dev_ptr, int64_t count, int dtype, int root);

/* Count matrix (what df2sparr(inp_df) returns, scLENS.jl:662) + the zero-candidate list
 * (z_idx1, z_idx2 of scLENS.jl:668-673, 0-based, disjoint from the stored entries, unique). */
int sclens_hip_session_create(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, int64_t n_cand, const uint32_t* z_idx1, const uint32_t* z_idx2,
                              sclens_hip_session** out);
void sclens_hip_session_destroy(sclens_hip_session* s);
/* A worker session on another context of the SAME device (its own stream and scratch) sharing the read-only device data
 * of `src` (pattern, and Vr2 / seed block if already computed; see sclens_hip_session_adopt). Valid calls on it:
 * null_spectrum, binary_basis, search_step*, perturb*, export_slot. Destroy it before `src`. Lets two independent decompositions overlap on one GPU (one host thread each). */
int sclens_hip_session_clone(sclens_hip_ctx* ctx2, sclens_hip_session* src, sclens_hip_session** out);

/* Row-sharded session for the atlas configuration (cells > genes, so the Gram matrix X'X is a sum over
 * cells): this process holds the cells [row0, row0 + N_local) of an N_global x M count matrix (colptr / rowval / nzval:
 * the LOCAL cells, row indices 0 .. N_local-1; z_idx1: GLOBAL cell indices of the whole candidate list, identical on
 * every rank). `allreduce` sums a device buffer over all ranks in place (RCCL via torch.distributed in the Python host;
 * any equivalent); the library calls it, on the calling thread and with its stream synchronised, where the path has a
 * real exchange: three O(M) fp64 vectors + one scalar inside each normalisation (column mean / variance / centring,
 * mean row norm), the M x M fp32 partial Gram matrix of each decomposition (scLENS.jl:332-361), the norms of the
 * recovered cell-side vectors (:503-508), and the small k x min_pc / (P k)^2 / k x M products of the robustness scoring
 * (:788-795) and the gene basis (:813-818). The eigendecompositions run replicated on identical inputs. Every session
 * call must be made by all ranks in the same order. Outputs on the gene side are identical on every rank; signal_vectors,
 * get_perturbed, rec_tgc, rec_norm_tgc cover the local cells. Mean centring only; no late candidate attachment.
 * A worker clone needs its own exchange channel: sclens_hip_session_set_reducer. */
int sclens_hip_session_create_sharded(sclens_hip_ctx* ctx, int64_t N_global, int64_t row0, int64_t N_local, int64_t M,
                                      const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t n_cand,
                                      const uint32_t* z_idx1, const uint32_t* z_idx2, sclens_hip_allreduce_fn allreduce,
                                      void* user, sclens_hip_session** out);
int sclens_hip_session_set_reducer(sclens_hip_session* s, sclens_hip_allreduce_fn allreduce, void* user);
int sclens_hip_session_set_reduce_to(sclens_hip_session* s, sclens_hip_reduce_fn reduce, void* user);

/* Row-sharded session whose zero candidates are LOCAL: this rank draws, on the device, its own part of the global draw sequence
 * R1 (scLENS.jl:668-673; draw t is a pure function of (seed, t), so the pairs that land in the cells [row0, row0 + N_local) are
 * found without the other ranks) from nnz_global = the stored entries of the WHOLE matrix. *n_cand_local = the length of this
 * rank's list. The global candidate list that the samples of the search / the ensemble index is the concatenation of the ranks'
 * lists in rank order: the host all-gathers the counts (a handful of integers) and tells every session its window with
 * sclens_hip_session_set_candidate_range(cand_off = sum of the counts of the lower ranks, n_cand_global = their total).
 * No rank ever holds another rank's candidates (1M x 30k on 8 GPUs: 2.6 GB of list + slots per rank instead of 21 GB). */
int sclens_hip_session_create_sharded_drawn(sclens_hip_ctx* ctx, int64_t N_global, int64_t row0, int64_t N_local, int64_t M,
                                            const int64_t* colptr, const int32_t* rowval, const float* nzval,
                                            int64_t nnz_global, uint64_t seed, sclens_hip_allreduce_fn allreduce, void* user,
                                            sclens_hip_session** out, int64_t* n_cand_local);
int sclens_hip_session_set_candidate_range(sclens_hip_session* s, int64_t cand_off, int64_t n_cand_global);
/* Local candidate list of such a session (z1 = GLOBAL cell indices, z2 = genes; n_cand_local entries each): tests / hosts
 * that want to replay the run unsharded. */
int sclens_hip_session_local_candidates(sclens_hip_session* s, uint32_t* z1, uint32_t* z2);

/* Chunked session (BASELINE configs[4], "chunked Gram accumulation"; cells > genes): ALL cells on this device, held as CSC chunks of
 * consecutive cells (8 bytes per stored entry in HBM) because the forms the plain session keeps resident do not fit -- at 1 000 000 x
 * 30 000 the union pattern is 175 GB and one scaled matrix 120 GB. Every decomposition then visits the chunks: the chunk's sparse
 * pattern is built on the device (with ITS part of the global candidate draw R1, scLENS.jl:668-673, when the matrix carries candidate
 * ones: the convention of sclens_hip_session_create_sharded_drawn; kept between visits as far as the context option chunk_cache_gb
 * allows), the statistics that span all cells (scLENS.jl:597-603, :682-690) are summed over the chunks in three passes, and the Gram
 * matrix is the sum of the chunks' contributions (scLENS.jl:332-361, X'X as a sum over cell blocks). The gene side -- eigensolver,
 * search statistic, partial eigensolver, scoring -- is that of the plain session; cell-side outputs cover all N_global cells.
 *   create_chunked: the shell; nnz_global = stored entries of the whole matrix, seed = the candidate seed.
 *   chunk_add(which = 0): chunk g = cells [row0, row0 + N_local) of the count matrix (LOCAL row indices), uploaded at once -- the host
 *     needs one chunk at a time; chunk_commit when all are in. which = 1 (after commit): the same cells of the null matrix X_r
 *     (scLENS.jl:701), consumed and released by null_spectrum_chunked (the chunked form of sclens_hip_session_null_spectrum).
 *   Then the calls of the plain session: data_spectrum, refine_eigenvalues, signal_vectors, binary_basis, search_step[_seeded],
 *   perturb[_seeded], robustness, gene_basis (mean centring; no worker clones; samples index the global candidate list, whose length
 *   session_get_int "n_cand" returns -- counted on first use). get_int "chunk_builds" / "chunk_visits": pattern builds / chunk visits. */
int sclens_hip_session_create_chunked(sclens_hip_ctx* ctx, int64_t N_global, int64_t M, int n_chunks, int64_t nnz_global, uint64_t seed,
                                      sclens_hip_session** out);
int sclens_hip_session_chunk_add(sclens_hip_session* s, int which, int g, int64_t row0, int64_t N_local, const int64_t* colptr,
                                 const int32_t* rowval, const float* nzval);
int sclens_hip_session_chunk_commit(sclens_hip_session* s);
int sclens_hip_session_null_spectrum_chunked(sclens_hip_session* s, double* Lr);

/* One ROUND of the sparsity search of a row-sharded session (scLENS.jl:725-761 evaluated `count` sparsities at a time, SURVEY
 * 8e-ii + 8e-iii): for evaluation e (sample seeds[e] of m[e] candidates) every rank forms its partial Gram matrix, which is summed
 * onto rank roots[e] ONLY (sclens_hip_session_set_reduce_to; all-reduce otherwise); after the last one every rank decomposes the
 * evaluation it is the root of (my_slot = its index in the round, -1: none) -- the eigensolves of a round run in parallel on
 * different GPUs instead of replicated on all. d5 / r_it receive the statistics of evaluation my_slot. All ranks make the call
 * with the same seeds / m / roots. */
int sclens_hip_session_search_round_seeded(sclens_hip_session* s, const uint64_t* seeds, const int64_t* m, const int32_t* roots,
                                           int count, int my_slot, int64_t n_2, double* d5, int64_t* r_it);
/* The same for the perturbation ensemble (scLENS.jl:771-778): members t[e] (slots), eigen-solve of member e on rank roots[e],
 * whose leading gene-side vectors are then shared (summed from one rank) so that every rank recovers ITS cells of the member's
 * cell-side vectors. nL_top: count x min_pc, ncols: count (identical on every rank). */
int sclens_hip_session_perturb_round_seeded(sclens_hip_session* s, const int64_t* t, const uint64_t* seeds, const int64_t* m,
                                            const int32_t* roots, int count, int my_slot, int64_t min_pc, double* nL_top,
                                            int64_t* ncols);

/* Late candidate attachment. The data / null / binarised decompositions (scLENS.jl:676-721) do not involve the zero
 * candidates, so a session may be created with n_cand = 0 and start them at once, while the host still draws the
 * candidates (:668-673) and merges them into the sparse pattern: sclens_hip_pattern_create builds the union pattern
 * (counts + candidates; host work + one upload on `ctx`'s stream, any context of the device, thread-safe w.r.t. work on
 * other contexts); sclens_hip_session_set_pattern hands it to the idle owner session (which takes ownership: the handle
 * is empty afterwards and only needs sclens_hip_pattern_destroy); worker clones then call session_adopt(what = 4). */
typedef struct sclens_hip_pattern sclens_hip_pattern;
int sclens_hip_pattern_create(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                              const float* nzval, int64_t n_cand, const uint32_t* z_idx1, const uint32_t* z_idx2,
                              sclens_hip_pattern** out);
/* The same with the zero candidates drawn ON THE DEVICE (R1, scLENS.jl:668-673: nnz uniform (i, j) pairs minus the stored
 * entries, first occurrences in draw order; the list sclens_draw_zero_candidates returns for the same seed) and merged into
 * the union pattern there: the host passes only the counts' CSC. *n_cand = length of the candidate list.
 * sclens_hip_pattern_candidates copies that list to the host (z1 = cells, z2 = genes, 0-based, n_cand entries each);
 * sclens_hip_pattern_download copies one device array of the pattern (tests: which = 0 colptr[M+1] i64, 1 row[nU] i32,
 * 2 rowptr[N+1] i64, 3 csr2csc[nU] i64, 4 csrcol[nU] i32, 5 cand_pos[n_cand] i64, 6 base_val[nU] f32). */
int sclens_hip_pattern_create_drawn(sclens_hip_ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval,
                                    const float* nzval, uint64_t seed, sclens_hip_pattern** out, int64_t* n_cand);
int sclens_hip_pattern_candidates(sclens_hip_ctx* ctx, sclens_hip_pattern* p, uint32_t* z1, uint32_t* z2);
int sclens_hip_pattern_download(sclens_hip_ctx* ctx, sclens_hip_pattern* p, int which, void* dst);
void sclens_hip_pattern_destroy(sclens_hip_pattern* p);
int sclens_hip_session_set_pattern(sclens_hip_session* s, sclens_hip_pattern* p);
/* The same two constructors from a device-resident count matrix (no upload): a counts-only session, and the union pattern with the
 * zero candidates drawn on the device. */
int sclens_hip_session_create_from_counts(sclens_hip_ctx* ctx, const sclens_hip_counts* counts, sclens_hip_session** out);
int sclens_hip_pattern_create_drawn_from_counts(sclens_hip_ctx* ctx, const sclens_hip_counts* counts, uint64_t seed,
                                                sclens_hip_pattern** out, int64_t* n_cand);

/* First half of get_sigev (scLENS.jl:526-537, :569-576): eigenvalues (ascending, length min(N,M)) of the
 * Gram matrix of the scaled data (L) and of the scaled null matrix X_r (Lr; CSC, same shape).
 * rec_* receive rec_vals (scLENS.jl:676-696); any of them may be NULL. */
int sclens_hip_session_spectrum(sclens_hip_session* s, const int64_t* r_colptr, const int32_t* r_rowval,
                                const float* r_nzval, double* L, double* Lr, double* rec_tgc, double* rec_mat2_mean,
                                double* rec_mat2_std, double* rec_norm_tgc, double* rec_cent);
/* The two halves of session_spectrum as separate calls, so that the null matrix can be decomposed by a worker session
 * (sclens_hip_session_clone) while the main session decomposes the data matrix. */
int sclens_hip_session_null_spectrum(sclens_hip_session* s, const int64_t* r_colptr, const int32_t* r_rowval,
                                     const float* r_nzval, double* Lr);
/* null_spectrum from a pattern of X_r built beforehand with sclens_hip_pattern_create (n_cand = 0), so that the host part of
 * the upload can run while the session is still being created; the pattern stays with the caller. */
int sclens_hip_session_null_spectrum_pattern(sclens_hip_session* s, sclens_hip_pattern* x_r, double* Lr);
int sclens_hip_session_data_spectrum(sclens_hip_session* s, double* L, double* rec_tgc, double* rec_mat2_mean,
                                     double* rec_mat2_std, double* rec_norm_tgc, double* rec_cent);
/* Share read-only device results between sessions of one GPU: what = 1 Vr2 (after binary_basis on src), 2 the seed
 * block of the partial eigensolver (after signal_vectors on src), 4 the sparse pattern (after set_pattern on src);
 * flags may be or-ed. `src` must outlive `dst`'s use of them. */
int sclens_hip_session_adopt(sclens_hip_session* dst, sclens_hip_session* src, int what);
/* The same between GPUs: device buffers of Vr2 (what = 1) and of the seed block of the partial eigensolver (what = 2: its
 * b0 rows, their eigenvalues theta0[b0] and the signal count k), for a host that spreads the data / null / binarised
 * decompositions (scLENS.jl:704, :717-721) over ranks and broadcasts the results (RCCL on *dev_ptr: rows x *ld floats).
 * rows = 0: owner side, query (theta0 is filled); rows > 0: receiver side, allocate for `rows` rows and record k / theta0
 * before the broadcast lands. For the seed block *dev_ptr may come back NULL (*rows_out = 0): no seed, full solver. */
int sclens_hip_session_shared_buffer(sclens_hip_session* s, int what, int64_t rows, int64_t k, double* theta0,
                                     void** dev_ptr, int64_t* rows_out, int64_t* k_out, int64_t* ld);
/* Guard band of the signal threshold: `sum(L .> lambda_c)` (scLENS.jl:539, :541, :580) is a hard cut on eigenvalues that the
 * fp32 solver (the reference: cuSOLVER syevd!, scLENS.jl:377) delivers with an error of ~ sqrt(n) eps32 lambda_max. For the
 * ascending eigen-indices idx_lo .. idx_hi-1 of the data matrix, rho[q] receives the float64 Rayleigh quotient of the fp32
 * eigenvector against the resident scaled matrix, which the host substitutes for L before the comparison. Valid between
 * session_data_spectrum (session_spectrum) and session_signal_vectors. */
int sclens_hip_session_refine_eigenvalues(sclens_hip_session* s, int64_t idx_lo, int64_t idx_hi, double* rho);
/* Second half (scLENS.jl:541-558, :580-590): cell-side eigenvectors of the k largest eigenvalues,
 * descending; nV is N x k (may be NULL: they also stay on the device for the later steps). */
int sclens_hip_session_signal_vectors(sclens_hip_session* s, int64_t k, float* nV);

/* Vr2 of scLENS.jl:717-721: eigenbasis of the binarised pattern. L_bin[min(N,M)] ascending; *r = number of
 * positive eigenvalues (columns of Vr2, kept on the device). */
int sclens_hip_session_binary_basis(sclens_hip_session* s, double* L_bin, int64_t* r);
/* One iteration of the sparsity search (scLENS.jl:731-747): sample = indices into the candidate list
 * (0-based, distinct); n_2 as at scLENS.jl:722. d5 = the five smallest values of d_arr, ascending. */
int sclens_hip_session_search_step(sclens_hip_session* s, const uint32_t* sample, int64_t m, int64_t n_2, double* d5,
                                   int64_t* r_it);
/* The same with the sample drawn on the device: indices sclens_sample_without_replacement(n_cand, m, seed). */
int sclens_hip_session_search_step_seeded(sclens_hip_session* s, uint64_t seed, int64_t m, int64_t n_2, double* d5,
                                          int64_t* r_it);
int sclens_hip_session_perturb_seeded(sclens_hip_session* s, int64_t t, uint64_t seed, int64_t m, int64_t min_pc,
                                      double* nL_top, int64_t* ncols);
/* One member of the perturbation ensemble (scLENS.jl:772-777): keeps the first min(min_pc, r) cell-side
 * eigenvectors in device slot t; nL_top[min_pc] their eigenvalues (descending); *ncols their number. */
int sclens_hip_session_perturb(sclens_hip_session* s, int64_t t, const uint32_t* sample, int64_t m, int64_t min_pc,
                               double* nL_top, int64_t* ncols);
/* Download slot t (N x ncols, column-major) -- for tests and for callers that score on the host. */
int sclens_hip_session_get_perturbed(sclens_hip_session* s, int64_t t, float* nV_t);
/* Options / counters. set: "chefsi" (1 = ensemble members use the leading-eigenpair subspace iteration seeded with the
 * data matrix's eigenvectors, falling back to the full solver if it does not converge; 0 = always the full solver).
 * get: "chefsi", "chefsi_used", "chefsi_fallback".
 * "chefsi_tail_gap_milli" (set / get): the eigenpairs k .. min_pc-1 of a member additionally held to value/1000 x their gap to the
 * block's smallest Ritz value (a host switches it on to solve a member again whose tail vector the matching :788 picked).
 * "chefsi_tail_free" (set / get, round 5): 1 = those tail pairs are not converged at all (a member then costs a third of the passes
 * over its scaled matrix); sclens_hip_session_robustness then records per member whether the matching provably does not depend on
 * them -- signal i correlates with any unit vector orthogonal to the member's first k eigenvectors by at most
 * sqrt(1 - sum_{j<k} c_ij^2), so the argmax of :788 lies among the first k whenever the best of them beats that bound (with a margin of
 * 2e-2 for the angle error of the strict pairs) --:
 * get "match_uncertain:<t>" = 1 for a member t without that proof (or whose matching picked a column >= k), "match_uncertain_count"
 * their number. The host solves such members again with "chefsi_tail_free" = 0 and repeats sclens_hip_session_robustness. */
/* Row-sharded sessions, set: "shard_rank" (this rank's index, once) and "solve_root" (-1 = every rank, the default): the rank that runs
 * the eigensolver of the NEXT null_spectrum / data_spectrum (+ the refine_eigenvalues / signal_vectors that continue from it) /
 * binary_basis -- the partial Gram matrix is summed onto that rank only (sclens_hip_session_set_reduce_to) and its eigenvalues, the
 * vectors the other ranks recover their cells from, and Vr2 reach them as sums in which they contribute zeros: the three first
 * decompositions (scLENS.jl:704, :717-721) run on three ranks side by side instead of replicated on all (sclens_amd/atlas.py).
 * get: "n_cand", "chunks", "chunk_builds", "chunk_visits", "chunk_cached" (chunked sessions), "gram_sparse_used". */
int sclens_hip_session_set_int(sclens_hip_session* s, const char* name, int64_t value);
int sclens_hip_session_get_int(sclens_hip_session* s, const char* name, int64_t* value);
/* Multi-GPU ensemble sharding: copy slot t to / from a contiguous device buffer of min_pc x ldn floats
 * (ldn = N rounded up to 32; sclens_hip_session_slot_ld). The buffer may belong to the host framework (e.g. a torch
 * tensor that RCCL all-gathers). */
int64_t sclens_hip_session_slot_ld(sclens_hip_session* s);
int sclens_hip_session_export_slot(sclens_hip_session* s, int64_t t, int64_t min_pc, void* dst_dev);
int sclens_hip_session_import_slot(sclens_hip_session* s, int64_t t, int64_t min_pc, int64_t ncols, const void* src_dev);
/* Robustness matching (scLENS.jl:788-795) over slots 0..P-1: a_b is k x P (0-based column picks,
 * column-major), b is k x P(P-1)/2 ROW-major (pair order i<j as at :792). */
int sclens_hip_session_robustness(sclens_hip_session* s, int64_t P, int32_t* a_b, double* b);
/* gene_basis (scLENS.jl:813-818): (nL^-1/2 .* nV') * scaled_X / sqrt(M), written as out[q*M + j]
 * (k rows of M genes). */
int sclens_hip_session_gene_basis(sclens_hip_session* s, const double* nL, float* out);

/* ---------------------------------------------------------------- multi-GPU: RCCL inside the library */
/* One process per GPU. The perturbation ensemble (scLENS.jl:771-778) runs member t on rank t mod G and needs ONE gather of the
 * min_pc x N blocks at the end; the data / null / binarised decompositions (:704, :717-721) may run on different ranks and
 * broadcast Vr2 and the seed block; row-sharded cells (sclens_hip_session_create_sharded) all-reduce their statistics and partial
 * Gram matrices. These collectives are RCCL calls made BY THE LIBRARY on its own device buffers and on the owning context's
 * stream; the host only launches the ranks and ships the 128-byte unique id from rank 0 to the others (any channel).
 * All calls block until the result is in place. `*_host` variants take host buffers (the few doubles of the control flow). */
#define SCLENS_HIP_COMM_ID_BYTES 128
typedef struct sclens_hip_comm sclens_hip_comm;
int sclens_hip_comm_unique_id(sclens_hip_ctx* ctx, uint8_t* id /*[128]*/);                 /* rank 0: ncclGetUniqueId */
int sclens_hip_comm_create(sclens_hip_ctx* ctx, const uint8_t* id, int rank, int world, sclens_hip_comm** out);
void sclens_hip_comm_destroy(sclens_hip_comm* comm);
/* what RCCL reports for the communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion) */
int sclens_hip_comm_info(sclens_hip_comm* comm, int* world, int* rank, int* rccl_version);
int sclens_hip_comm_stats(sclens_hip_comm* comm, int64_t* calls, double* bytes);
const char* sclens_hip_comm_last_error(sclens_hip_comm* comm);
int sclens_hip_comm_allreduce(sclens_hip_comm* comm, void* dev_ptr, int64_t count, int dtype /*0 fp64, 1 fp32*/); /* in-place sum */
int sclens_hip_comm_broadcast(sclens_hip_comm* comm, void* dev_ptr, int64_t nbytes, int root);
int sclens_hip_comm_allgather(sclens_hip_comm* comm, const void* send_dev, void* recv_dev, int64_t nbytes_per_rank);
int sclens_hip_comm_allgather_host(sclens_hip_comm* comm, const void* send, void* recv, int64_t nbytes_per_rank);
int sclens_hip_comm_broadcast_host(sclens_hip_comm* comm, void* buf, int64_t nbytes, int root);
/* an sclens_hip_allreduce_fn whose `user` is the communicator: pass (sclens_hip_comm_allreduce_cb, comm) to
 * sclens_hip_session_create_sharded / _set_reducer and the row-sharded session reduces over RCCL with no host callback */
int sclens_hip_comm_allreduce_cb(void* user, void* dev_ptr, int64_t count, int dtype);
/* an sclens_hip_reduce_fn (ncclReduce onto rank `root`) with user = the communicator */
int sclens_hip_comm_reduce_cb(void* user, void* dev_ptr, int64_t count, int dtype, int root);

/* ---------------------------------------------------------------- device-level entry points ---- */
/* Used by the repository's own tests and bench.py (device pointers, row-major; see csrc/common.h). */
int sclens_hip_dev_gemm_f32(sclens_hip_ctx* ctx, const float* P, const float* Q, float* C, int64_t M, int64_t N,
                            int64_t K, int64_t ldp, int64_t ldq, int64_t ldc, float alpha, float beta, int q_kcontig,
                            int lower, uint32_t* colabsmax);
int sclens_hip_dev_gram_f32(sclens_hip_ctx* ctx, const float* B, int64_t n, int64_t K, int64_t ldb, float divisor,
                            float* A, int64_t lda);
/* stage 1 of the two-stage reduction (work in progress, see sbr.hip): dense symmetric -> band of half-width 64; n must be a
 * multiple of 64. Lower band of A = the band matrix, upper part = panel reflectors, T[(n/64-1)][64][64] their factors. */
int sclens_hip_dev_sy2sb_f32(sclens_hip_ctx* ctx, float* A, int64_t n, int64_t lda, float* T, int* breakdown);
/* first back-transformation: the m rows of Zt (length n, leading dimension ldz) are multiplied by Q1 of sy2sb (A, T) */
int sclens_hip_dev_sbr_apply_q1_f32(sclens_hip_ctx* ctx, const float* A, int64_t n, int64_t lda, const float* T, float* Zt,
                                    int64_t m, int64_t ldz);
/* second back-transformation: the m rows of Zt are multiplied by Q2 of the preceding sb2st on this context */
int sclens_hip_dev_sbr_apply_q2_f32(sclens_hip_ctx* ctx, int64_t n, float* Zt, int64_t m, int64_t ldz);
/* stage 2 (work in progress): band (output of sy2sb) -> tridiagonal d[n], e[n] (fp64, device) by bulge chasing */
int sclens_hip_dev_sb2st_f32(sclens_hip_ctx* ctx, const float* A, int64_t n, int64_t lda, double* d, double* e);
int sclens_hip_dev_sytrd_f32(sclens_hip_ctx* ctx, float* A, int64_t n, int64_t lda, double* d, double* e, float* tau);
int sclens_hip_dev_stebz_f64(sclens_hip_ctx* ctx, const double* d, const double* e, int64_t n, double* w);
int sclens_hip_dev_eigh_f32(sclens_hip_ctx* ctx, float* A, int64_t n, int64_t lda, double* w, int64_t vec_lo,
                            int64_t vec_hi, float* Zt, int64_t ldz);
void* sclens_hip_dev_malloc(sclens_hip_ctx* ctx, int64_t bytes);
void sclens_hip_dev_free(sclens_hip_ctx* ctx, void* p);
int sclens_hip_dev_memcpy(sclens_hip_ctx* ctx, void* dst, const void* src, int64_t bytes, int kind /*1 H2D, 2 D2H, 3 D2D*/);
int sclens_hip_dev_memset(sclens_hip_ctx* ctx, void* dst, int value, int64_t bytes);
int sclens_hip_dev_sync(sclens_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SCLENS_HIP_H */
