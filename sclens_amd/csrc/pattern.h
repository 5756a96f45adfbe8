// Sparse pattern shared by every matrix of one sclens() call: stored counts UNION zero candidates
// (scLENS.jl:664-673), held on the device as a CSC view plus a CSR view of the same slots.
#pragma once
#include "common.h"

namespace scl {

struct PatternDev {
  int64_t N = 0, M = 0, nU = 0, ncand = 0;
  const int64_t* colptr = nullptr;   // [M+1]  CSC: per column the stored counts (rows ascending), then candidates
  const int32_t* row = nullptr;      // [nU]
  const int64_t* rowptr = nullptr;   // [N+1]  CSR view (columns ascending inside a row)
  const int64_t* csr2csc = nullptr;  // [nU]   CSC slot of each CSR slot
  const int32_t* csrcol = nullptr;   // [nU]
  const int64_t* cand_pos = nullptr; // [ncand] CSC slot of candidate t (-1: not in this session's cells, row-sharded mode)
  // Samples index a GLOBAL candidate list of ncand_global entries of which this pattern holds [cand_off, cand_off + ncand)
  // (row-sharded sessions with local candidates: the global list is the concatenation of the ranks' lists in rank order).
  // ncand_global == 0: the pattern holds the whole list (cand_off = 0, ncand_global = ncand).
  int64_t cand_off = 0, ncand_global = 0;
  int64_t population() const { return ncand_global > 0 ? ncand_global : ncand; }
  // CSR companions (pattern_add_csr_companions; nullptr = off). Every value array over such a pattern is 2 nU floats long: the
  // values in CSC slot order, then THE SAME values in CSR slot order -- the row reductions of the normalisation (row sums of the
  // counts, row norms of the scaled rows: scLENS.jl:607, :603) then stream their operand instead of gathering it through
  // csr2csc (10 ms each per decomposition at 100 000 x 30 000, 0.066 of the HBM roofline for the stage in round 2).
  const float* base_val_csr = nullptr;     // [nU] base_val in CSR slot order
  const int64_t* cand_pos_csr = nullptr;   // [ncand] CSR slot of candidate t (-1 as cand_pos)
  size_t val_floats() const { return (size_t)nU * (base_val_csr ? 2 : 1); }
};

struct PatternOwner {  // owns the device arrays of a PatternDev
  PatternDev dev;
  float* base_val = nullptr;  // [nU] stored counts, 0 in candidate slots
  const uint32_t* z1_dev = nullptr;  // [ncand] candidate list on the device (device-built patterns; nullptr otherwise)
  const uint32_t* z2_dev = nullptr;
  std::vector<void*> allocs;
};

// Build on the host (O(nU)) and upload. Indices are 0-based. z1/z2 may be null when ncand == 0.
// row0 / N_global: row-sharded session, z1 holds GLOBAL cell indices of which [row0, row0 + N) are local (N_global = 0: N).
int pattern_build(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval,
                  int64_t ncand, const uint32_t* z1, const uint32_t* z2, PatternOwner* out, int64_t row0 = 0,
                  int64_t N_global = 0);
void pattern_free(PatternOwner* p, Ctx* busy = nullptr);  // busy: a context whose streams may still hold work on the blocks (error paths)
// device helper shared by both builders: base_val_csr / cand_pos_csr from csr2csc, base_val, cand_pos (context option val_csr = 0: skip)
int pattern_add_csr_companions(Ctx* ctx, PatternOwner* out);
// The same arrays built on the device from the counts' CSC (pattern_dev.hip); draw != 0 also draws the candidate list there
// (R1, scLENS.jl:668-673; the list of sclens_draw_zero_candidates for the same seed). Sessions that hold all cells only.
// blk != nullptr (with draw): the matrix is the block [row0, row0 + N) of the cells of an N_global x M matrix with nnz_global stored
// entries in all; the candidates are this block's part of the GLOBAL draw sequence (local cell indices in z1_dev)
struct BlockDraw {
  int64_t N_global, row0, nnz_global;
};
int pattern_build_device(Ctx* ctx, int64_t N, int64_t M, const int64_t* colptr, const int32_t* rowval, const float* nzval, int64_t ncand,
                         const uint32_t* z1_h, const uint32_t* z2_h, int draw, uint64_t seed, PatternOwner* out, int64_t nnz_dev = -1,
                         const BlockDraw* blk = nullptr);

// A count matrix that lives in HBM (CSC, 0-based): the filtered output of preprocess (sclens_hip_preprocess_keep) or an upload.
// Sessions and patterns built from it read it in place; it may be shared by the contexts of its device.
struct Counts {
  int device = 0;
  int64_t N = 0, M = 0, nnz = 0;
  int64_t* colptr = nullptr;  // [M + 1]
  int32_t* row = nullptr;     // [nnz]
  float* val = nullptr;       // [nnz]
};
void counts_free(Counts* c, Ctx* busy = nullptr);

// Row-sharded session: the all-reduce the host supplies. dtype 0 = fp64, 1 = fp32; sum over all ranks, in
// place, on a device buffer; called from the thread that made the session call, after the session's stream has been
// synchronised, and must return only when the buffer holds the result.
struct ShardReduce {
  int64_t N_global = 0, row0 = 0;
  sclens_hip_allreduce_fn fn = nullptr;
  void* user = nullptr;
  // optional: sum onto ONE rank (`root`; the other ranks' buffers are left undefined). Used where only one rank consumes the sum
  // (the Gram matrix of a search evaluation / ensemble member that `root` decomposes); absent: all-reduce.
  sclens_hip_reduce_fn rfn = nullptr;
  void* ruser = nullptr;
  // a worker clone starts with its parent's description of the sharding but must not use the parent's channel (another host thread
  // would issue collectives on the parent's communicator and stream): every collective is refused until session_set_reducer has
  // given the clone a channel of its own
  bool inherited = false;
  bool on() const { return fn != nullptr; }
  int sum(Ctx* ctx, void* dev, int64_t count, int dtype) const {
    if (!fn) return SCLENS_OK;
    if (inherited) return ctx->fail(SCLENS_ERR_STATE, "row-sharded worker session: call set_reducer before any collective");
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return ctx->fail(SCLENS_ERR_HIP, std::string("allreduce: ") + hipGetErrorString(e));
    const int rc = fn(user, dev, count, dtype);
    return rc == 0 ? SCLENS_OK : ctx->fail(SCLENS_ERR_HIP, "allreduce callback failed with code " + std::to_string(rc));
  }
  int sum_to(Ctx* ctx, void* dev, int64_t count, int dtype, int root) const {
    if (inherited) return ctx->fail(SCLENS_ERR_STATE, "row-sharded worker session: call set_reducer before any collective");
    if (!rfn) return sum(ctx, dev, count, dtype);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return ctx->fail(SCLENS_ERR_HIP, std::string("reduce: ") + hipGetErrorString(e));
    const int rc = rfn(ruser, dev, count, dtype, root);
    return rc == 0 ? SCLENS_OK : ctx->fail(SCLENS_ERR_HIP, "reduce callback failed with code " + std::to_string(rc));
  }
};

struct ScaleVecs {  // host destinations for rec_vals (scLENS.jl:676-696); all fp64
  double* tgc;        // [N]
  double* mat2_mean;  // [M]
  double* mat2_std;   // [M]
  double* norm_tgc;   // [N]
  double* cent;       // [M]
};

// Dense scaled matrix from a value array over the pattern.
//   f32path = 1 : closure path logn_scale(pre_scale(x)) (Float32 proj_l/log1p/std)
//   f32path = 0 : inline Float64 path of the data matrix (scLENS.jl:676-696)
//   centering = 0 : mean (zscore_with_l2 + final centring), 1 : median (scLENS.jl:653-654; no rec_vals there)
//   cells_major = 1 : B[i*ldb + j] (N rows of M genes); 0 : B[j*ldb + i] (M rows of N cells)
int scale_to_dense(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, int cells_major,
                   float* B, int64_t ldb, ScaleVecs* keep);

// the statistics alone (device pointers into the context's workspaces, valid until the next normalisation on this context)
struct ScaleStats {
  double *tgc, *lg, *mean, *stdv, *mu, *l2, *srow, *cent;
  double* red;  // [0] = ||mu||^2, [1] = sum_i l_i, [2] = sum_i s_i
};
int scale_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, ScaleStats* out);
// scale_to_dense that also hands out the statistics it computed; B == nullptr: the statistics (and rec_vals) only
int scale_to_dense_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, int centering, int cells_major, float* B, int64_t ldb,
                         ScaleVecs* keep, ScaleStats* out);
// Gram matrix of the scaled matrix from its SPARSE structure (gram_sparse.hip; cells > genes layout, mean centring); gram_sparse_macs:
// its multiply-adds, sum_i r_i^2 / 2 over the slots of the rows
int gram_sparse_macs(Ctx* ctx, const PatternDev& p, double* macs);
int gram_sparse(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* tgc, const double* lg, const double* stdv,
                const double* mu, const double* l2, const double* cent, const double* lsum, double n_all, double alpha, double beta,
                float* A, int64_t lda, bool accumulate);

// Gram matrix of the scaled matrix of a BINARY value array (val in {0, 1}; mean centring, all cells on this device, N > M
// layout) without forming the scaled matrix (gram_bits.hip): A (M x M, lda, zero padded, exactly symmetric) =
// scaled(P)' scaled(P) / divisor. `scratch` holds the M x round_up(N, 64) fp16 image of P.
size_t gram_binary_scratch_bytes(int64_t N, int64_t M);
struct ShardReduce;
int gram_binary(Ctx* ctx, const PatternDev& p, const float* val, int f32path, void* scratch, float divisor, float* A, int64_t lda,
                const ShardReduce* sh = nullptr);
int gram_binary_stats(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const ScaleStats* given, void* scratch, float divisor,
                      float* A, int64_t lda, const ShardReduce* sh, bool accumulate);
int scale_stats_sharded(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const ShardReduce& sh, ScaleStats* out);

// the same for a session that holds p.N of sh.N_global cells (N > M layout: B[j][i_local]); mean centring only
int scale_to_dense_sharded(Ctx* ctx, const PatternDev& p, const float* val, int f32path, float* B, int64_t ldb,
                           ScaleVecs* keep, const ShardReduce& sh);

// chunked session (session.hip): the normalisation of a matrix whose cells are visited in chunks (scale.hip, "chunked variant")
int chunk_pass_sum(Ctx* ctx, const PatternDev& p, const float* val, int f32path, double* acc);
int chunk_pass_var(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* acc, double n_global, double* acc_s2);
int chunk_stats_finish(Ctx* ctx, int64_t M, const double* acc, const double* acc_s2, double n_global, int f32path, double* mean, double* stdv,
                       double* mu, double* red);
int chunk_dense(Ctx* ctx, const PatternDev& p, const float* val, int f32path, const double* stdv, const double* mu, const double* red, double num,
                const double* cent, double* accT, float* B, int64_t ldb, double** tgc_out, double** l2_out, double** lg_out = nullptr,
                double** srow_out = nullptr);
int chunk_gram_finish(Ctx* ctx, float* A, int64_t n, int64_t lda, const double* T, const double* stdv, const double* mu, double n_global,
                      double divisor, double* cent);

// val = (binary ? pattern-of-counts : counts), then 1 at the candidate slots idx_dev[0..m)
int make_values(Ctx* ctx, const PatternDev& p, const float* base_val, int binary, const uint32_t* idx_dev, int64_t m,
                float* out);
// the same with idx = the first m values of the keyed Feistel permutation of [0, ncand) (rng.h), evaluated on the device
int make_values_seeded(Ctx* ctx, const PatternDev& p, const float* base_val, int binary, uint64_t seed, int64_t m,
                       float* out);

}  // namespace scl
