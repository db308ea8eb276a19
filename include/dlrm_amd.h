/*
 * dlrm_amd -- C ABI of the MI355X (gfx950) DLRM hot path.
 *
 * One shared library (libdlrm_amd.so) exports everything below.  It replaces,
 * for the data-parallel hot path only, the C++ model interface the reference
 * driver is written against (dlrm_libtorch/src/dlrmtest.cpp with
 * dlrm_libtorch/include/data.h, and src/dlrmtest2.cpp with include/data2.h for the
 * second cache policy); each entry point cites the reference interface it stands in for.
 *
 * Sections: tables - EmbeddingBag (forward, fused sparse update) - MLP layers - interaction and losses -
 * page-LRU cache (dlrm_cache: declarations, then its calls further down) - Criteo record decode - recorded
 * step (dlrm_plan) - vector-granular cache (dlrm_vcache) - collectives (dlrm_comm) - whole model (dlrm_model).
 *
 * Conventions
 *  - Every function returns DLRM_OK (0) or an error code; the message of the
 *    last error on the calling thread is dlrm_last_error().  Nothing throws.
 *  - Buffers named d_* are device (HBM) pointers owned by the caller; h_* are
 *    host pointers.  The library allocates nothing unless a function says so
 *    (the cache objects and the handles are the only owners of device memory).
 *  - stream is a hipStream_t passed as void*; NULL is the null stream.  All
 *    launches are asynchronous on it; no entry point synchronises unless it
 *    says so.
 *  - Index dtype is selectable (DLRM_I32 / DLRM_I64); indices and offsets of
 *    one call share it, as in torch::embedding_bag.
 *  - Bags are numbered g = t*batch + b (table major).  Pooled vectors and
 *    their gradients live at  base + b*ld + t*dim,  so with ld = T*dim this is
 *    the [B, T*D] tensor the reference builds with torch::cat
 *    (src/dlrmtest.cpp:196) and with ld = (T+1)*dim, base advanced by dim, it
 *    is the [B, 1+T, D] input of the dot interaction.
 *  - One host thread drives one GPU; handles are not shared between threads.
 */
#ifndef DLRM_AMD_H
#define DLRM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dlrm_stream_t;

enum { DLRM_OK = 0, DLRM_ERR_INVALID = 1, DLRM_ERR_HIP = 2, DLRM_ERR_NOMEM = 3, DLRM_ERR_STATE = 4 };
enum { DLRM_F32 = 0, DLRM_BF16 = 1 };
enum { DLRM_I32 = 0, DLRM_I64 = 1 };
enum { DLRM_POOL_SUM = 0, DLRM_POOL_MEAN = 1 };
enum { DLRM_DUP_ACCUMULATE = 0, DLRM_DUP_LAST_WINS = 1 };
enum { DLRM_OPT_SGD = 0, DLRM_OPT_ADAGRAD_ROWWISE = 1 };
enum { DLRM_ACT_NONE = 0, DLRM_ACT_RELU = 1 };

/* Message of the last failing call on this thread ("" if none). */
const char *dlrm_last_error(void);
/* ABI version: major*1000 + minor.  It changes whenever a struct of this header changes layout.
 * dlrm_version() is what the loaded library was built with; a caller compiled against another
 * DLRM_ABI_VERSION must not pass structs to it (the adapter and the Python binding check this
 * before their first call). */
#define DLRM_ABI_VERSION 1018
int dlrm_version(void);
/* Name of the device architecture the kernels were built for ("gfx950"). */
const char *dlrm_target_arch(void);

/* ------------------------------------------------------------------ tables */

/*
 * Fill n elements (first .. first+n of a table's flat element stream) with
 * uniform [0,1): value(i) = (splitmix64(splitmix64(seed) + i) >> 40) * 2^-24 (the seed is hashed, so the
 * streams of nearby seeds -- one per table -- are unrelated).
 * Replaces embedding_table_init::write() -> DATA::initial()
 * (include/data.h:217, src/data_transfer.cpp:47-61: rand()/(RAND_MAX+1.0)).
 * dst may be device memory or host-mapped pinned memory.
 */
int dlrm_table_init_uniform(void *dst, int row_dtype, int64_t n, uint64_t seed, int64_t first,
                            dlrm_stream_t stream);

/*
 * Strided 2-D copy of float32: dst[r*dst_ld + c] = src[r*src_ld + c] for
 * r < rows, c < cols.  Packs / unpacks the per-rank blocks of the pooled
 * embedding exchange (the reference has no counterpart: it is single process;
 * its closest relative is the torch::cat at src/dlrmtest.cpp:196).
 */
int dlrm_copy_2d(float *d_dst, int64_t dst_ld, const float *d_src, int64_t src_ld, int64_t rows, int64_t cols,
                 dlrm_stream_t stream);
/* Several such copies in one launch (the per-rank blocks of one exchange: a rank packs or unpacks the blocks of all
 * its peers at once instead of one short launch per peer). */
#define DLRM_COPY_MAX_ITEMS 16   /* per launch; longer lists take several launches */
typedef struct dlrm_copy_item {
    float *d_dst;
    int64_t dst_ld;
    const float *d_src;
    int64_t src_ld;
    int64_t rows, cols;
} dlrm_copy_item;
int dlrm_copy_2d_many(const dlrm_copy_item *items, int32_t n_items, dlrm_stream_t stream);

/* ------------------------------------------------------------ EmbeddingBag */

/*
 * A batch of lookups into T tables that share dim and dtype and sit in one
 * allocation (table t = rows [table_row_offsets[t], table_row_offsets[t+1]));
 * indices are table-local row ids.  Replaces struct emb: create_emb builds a
 * torch::nn::EmbeddingBag per table (src/dlrmtest.cpp:53-86, :144-169) and
 * emb::forward runs them one by one and concatenates (:170-206).
 */
typedef struct dlrm_embbag {
    void *d_weight;                     /* [total_rows, dim], row_dtype */
    const int64_t *d_table_row_offsets; /* device int64[T+1]: table t holds rows [o[t], o[t+1]) of d_weight. */
                                        /* NULL: every table addresses the whole row space [0, total_rows) */
                                        /* -- one table, or indices that are already rows of a shared pool */
                                        /* (the slot rows dlrm_cache_lookup returns) */
    int64_t total_rows;
    int32_t num_tables;
    int32_t dim;
    int32_t row_dtype;           /* DLRM_F32 | DLRM_BF16 */
    int32_t idx_dtype;           /* DLRM_I32 | DLRM_I64 */
    const void *d_indices;       /* idx[n_indices] */
    const void *d_offsets;       /* idx[T*batch] (+1 when include_last_offset): positions in d_indices.  With */
                                 /* include_last_offset the bags may address a part of a longer index array */
                                 /* (a descriptor for some of a batch's tables); the rest is ignored */
    int64_t n_indices;           /* length of d_indices */
    int64_t batch;
    int32_t include_last_offset;
    int32_t pooling;             /* DLRM_POOL_SUM | DLRM_POOL_MEAN */
    int32_t *d_error_flag;       /* optional device int32, set to 1 when an index is out of range */
                                 /* (such lookups are skipped); NULL = do not report */
    const int64_t *d_shard_row_begin; /* optional device int64[T] for row-wise sharded tables: table t of */
                                 /* this descriptor holds rows [begin[t], begin[t] + rows_t) of the full */
                                 /* table; indices are global, those outside the range belong to another */
                                 /* shard and are skipped without raising the error flag (mean pooling */
                                 /* still divides by the full bag length, so shard outputs add up). */
                                 /* NULL = tables are whole.  The reference stripes tables over parallel */
                                 /* units by page (include/data2.h:150, src/emb_cache2.cpp:701-709). */
} dlrm_embbag;

/*
 * out[b*out_ld + t*dim + d] = pool_{p in bag (t,b)} weight[t][indices[p]][d], float32.
 * Replaces emb::forward (src/dlrmtest.cpp:170-206) -> torch::embedding_bag.
 */
int dlrm_embbag_forward(const dlrm_embbag *e, float *d_out, int64_t out_ld, dlrm_stream_t stream);

/*
 * Sparse backward fused with the row-wise optimizer step: replaces autograd of
 * embedding_bag + torch::optim::SGD::step (src/dlrmtest.cpp:505-506, :531-533)
 * + the write-back save_emb / update_parameter (:88-107, :336-355).
 * Occurrences are sorted by row (stable radix sort), duplicates are reduced in
 * position order, and every touched row is read and written exactly once.
 *   grad row of bag (t,b) = d_grad + b*grad_ld + t*dim
 *   DLRM_DUP_ACCUMULATE: row gradient = sum over occurrences (libtorch autograd)
 *   DLRM_DUP_LAST_WINS : only the last occurrence counts (what the reference's
 *                        write-back loop does, src/dlrmtest.cpp:96-101)
 *   DLRM_OPT_SGD: w -= lr*g.  DLRM_OPT_ADAGRAD_ROWWISE: state[row] += mean(g^2),
 *   w -= lr*g/(sqrt(state[row])+eps), d_opt_state = float[total_rows].
 * d_workspace: dlrm_embbag_backward_workspace_bytes() bytes of device scratch.
 */
typedef struct dlrm_sparse_opt {
    int32_t optimizer; /* DLRM_OPT_* */
    int32_t dup_mode;  /* DLRM_DUP_* */
    float lr;
    float eps;
    float *d_opt_state; /* Adagrad: float[total_rows]; SGD: NULL */
} dlrm_sparse_opt;

int dlrm_embbag_backward_workspace_bytes(const dlrm_embbag *e, size_t *bytes);
int dlrm_embbag_backward_update(const dlrm_embbag *e, const float *d_grad, int64_t grad_ld,
                                const dlrm_sparse_opt *opt, void *d_workspace,
                                size_t workspace_bytes, dlrm_stream_t stream);
/*
 * The same in two halves.  _prepare needs the ids only (occurrences expanded and sorted by row into the
 * workspace), so it can run as soon as a batch's ids are on the device -- beside the forward pass, on
 * another stream.  _apply reduces the gradients over the sorted occurrences and updates the rows; it
 * must follow a _prepare with the same descriptor fields (ids, offsets, tables) and workspace.
 */
int dlrm_embbag_backward_prepare(const dlrm_embbag *e, void *d_workspace, size_t workspace_bytes,
                                 dlrm_stream_t stream);
int dlrm_embbag_backward_apply(const dlrm_embbag *e, const float *d_grad, int64_t grad_ld, const dlrm_sparse_opt *opt,
                               void *d_workspace, size_t workspace_bytes, dlrm_stream_t stream);

/* ------------------------------------------------------------- dense MLP */

/*
 * How a layer's float32 products are evaluated on the matrix cores (no counterpart in the reference,
 * whose GEMMs are libtorch sgemm calls).  Operands and results are float32 in memory either way.
 *   DLRM_GEMM_BF16X3 (what DLRM_GEMM_DEFAULT selects): every operand element is split exactly into three
 *      bf16 pieces and the product is formed from the six leading cross terms on v_mfma_f32_32x32x16_bf16
 *      with float32 accumulation.  Products of pieces are exact in float32 and the dropped terms are below
 *      2^-26 of a product, so the result has float32 accuracy (element by element within a few 2^-24 of
 *      sum_k |a_ik b_kj|, like a float32 fma chain) at 8/3 of the float32 matrix-pipe rate.
 *   DLRM_GEMM_F32_MFMA: v_mfma_f32_32x32x2_f32, bit for bit a k-ordered float32 fma chain.
 *   DLRM_GEMM_F16X2: each operand is multiplied by the power of two that puts its largest magnitude just below
 *      2^14 and every element is split into two f16 pieces, h = f16(x s) and l = f16((x s - h) 2^11) (the low
 *      piece is carried 2^11 times too large so that it stays a normal f16 number); h_a h_b goes to one float32
 *      accumulator, h_a l_b + l_a h_b to a second one that enters the result times 2^-11, on
 *      v_mfma_f32_32x32x16_f16, and the result is scaled back.  GUARANTEED RANGE: every element within 2^27 of its
 *      operand's largest magnitude keeps 22 mantissa bits whatever else the tensor holds; an element further down
 *      has an absolute error of at most 2^-48 of that largest magnitude (float32, BF16X3 and F32_MFMA have no such
 *      floor: operands spanning more than eight decades want one of those).  Within that range the
 *      representation error is about one float32 ulp of each product (7.5e-8 relative rms, against a
 *      float64 product; the float32 accumulation of ANY of the modes, this one included, contributes 2e-7 to 4e-7 at
 *      reduction lengths of 512 to 65536), at half the matrix work of BF16X3: 1.3 to 1.6 times faster on the
 *      65536-row layers.  It needs max |x| of both operands before the kernel starts: handed in through the
 *      descriptor's *_absmax fields (a GEMM of this library can leave its output's maximum behind for the next
 *      one), else measured by a pass over the operand inside the call, which costs most of what the mode gains.
 *      dlrm_model_* uses it for the layers of models with a local batch of 8192 and above when its gemm_mode is
 *      DLRM_GEMM_DEFAULT.
 * The mode is a field of the layer descriptor: a recorded step (dlrm_plan) keeps the mode it was recorded
 * with, and two host threads can use different modes.
 */
enum { DLRM_GEMM_DEFAULT = 0, DLRM_GEMM_F32_MFMA = 1, DLRM_GEMM_BF16X3 = 2, DLRM_GEMM_F16X2 = 3 };

/*
 * DLRM_GEMM_F16X2 with a layer's weights cut into their f16 pieces ONCE per step instead of by every workgroup of
 * every product that reads them (no counterpart in the reference).  dlrm_split_weights writes, for each item, the
 * two f16 planes of W and of W^T (scaled by the power of two that *d_w_absmax implies, zero padded to whole tiles)
 * into d_planes; a dlrm_linear / dlrm_linear_bwd whose w_planes points at the item then runs its forward /
 * backward-data product through a kernel that reads those planes (same arithmetic, same results as the in-kernel
 * split up to the order of the float32 accumulation).  The planes describe W as it was when dlrm_split_weights
 * ran: split again after every optimizer step.  The item is host memory that must stay valid while descriptors
 * (or a recorded dlrm_plan) refer to it.
 */
typedef struct dlrm_weight_planes {
    const float *d_w;          /* [N,K] */
    int64_t w_ld;
    int32_t N, K;
    const float *d_w_absmax;   /* device scalar: max |W| (dlrm_absmax), complete on the stream before dlrm_split_weights runs */
    void *d_planes;            /* dlrm_weight_planes_bytes(N, K) bytes of device memory, 16-byte aligned */
} dlrm_weight_planes;

int dlrm_weight_planes_bytes(int32_t N, int32_t K, size_t *bytes);
int dlrm_split_weights(const dlrm_weight_planes *items, int32_t n, dlrm_stream_t stream);

/*
 * y = act(x W^T + bias): one torch::nn::Linear + torch::relu of Net2::forward
 * (src/dlrmtest.cpp:254-272; layers made by create_fc, :119-140).  W is [N,K]
 * row-major (torch layout).  Any shape and leading dimension is accepted (K = 13, 479, ...);
 * operands whose rows start on 16-byte boundaries (pointer 16-byte aligned, leading dimension
 * a multiple of 4 floats) are read with 16-byte loads, others with 4-byte loads.  A leading dimension is a
 * promise that every row, the last one included, is readable up to it: when x_ld and w_ld both reach K rounded
 * up to 32, the columns between K and that bound are loaded too (and discarded; they need not be zero).
 * Layers with at most 16 inputs or a single output are no matrix product and run in one-pass float32 fma kernels.
 */
typedef struct dlrm_linear {
    int64_t M;          /* rows of x (batch) */
    int32_t N, K;       /* out features, in features */
    const float *d_x;   /* [M,K] */
    int64_t x_ld;
    const float *d_w;   /* [N,K] */
    int64_t w_ld;
    const float *d_bias; /* [N] or NULL */
    float *d_y;         /* [M,N] */
    int64_t y_ld;
    int32_t act;        /* DLRM_ACT_NONE | DLRM_ACT_RELU */
    int32_t gemm_mode;  /* DLRM_GEMM_* */
    /* DLRM_GEMM_F16X2 only, each may be NULL: device scalars holding max |x| and max |W| (NULL: the call measures
     * the operand in a pass of its own), and a scalar into which max |y| is folded with atomicMax (start it at 0) */
    const float *d_x_absmax, *d_w_absmax;
    float *d_y_absmax;
    /* DLRM_GEMM_F16X2 only, may be NULL: the pre-split planes of d_w (dlrm_split_weights); needs d_w_absmax = the item's */
    const dlrm_weight_planes *w_planes;
} dlrm_linear;

int dlrm_linear_forward(const dlrm_linear *l, dlrm_stream_t stream);

/*
 * max |x| over a [rows, cols] float32 matrix (row stride ld), folded into *d_slot with an atomic maximum on the bit
 * pattern: the slot must hold 0 (or an earlier maximum that is to be kept).  No counterpart in the reference; it
 * produces the operand maxima DLRM_GEMM_F16X2 takes for tensors that no GEMM of this library wrote.
 */
int dlrm_absmax(const float *d_x, int64_t rows, int32_t cols, int64_t ld, float *d_slot, dlrm_stream_t stream);

/*
 * Backward of the same layer; replaces the autograd pass loss.backward()
 * (src/dlrmtest.cpp:531) for the MLP.  dz is the gradient w.r.t. the
 * pre-activation z = x W^T + b.
 *   backward_data  : dx = dz W, then, if d_relu_mask != NULL,
 *                    dx = (relu_mask > 0) ? dx : 0  -- relu_mask is the output
 *                    of the previous layer (= this layer's x), which turns dx
 *                    directly into the previous layer's dz.
 *   backward_weight: dW = dz^T x  [N,K] and, if d_db != NULL, db = column
 *                    sums of dz.  The reduction over the batch is split over
 *                    workgroups; partial sums go through d_workspace
 *                    (dlrm_linear_backward_weight_workspace_bytes) and are
 *                    combined in a fixed order (reproducible).
 */
typedef struct dlrm_linear_bwd {
    int64_t M;
    int32_t N, K;
    const float *d_dz;  /* [M,N] */
    int64_t dz_ld;
    const float *d_w;   /* [N,K]  (backward_data) */
    int64_t w_ld;
    const float *d_x;   /* [M,K]  (backward_weight) */
    int64_t x_ld;
    float *d_dx;        /* [M,K]  (backward_data) */
    int64_t dx_ld;
    const float *d_relu_mask; /* [M,K] or NULL (backward_data) */
    int64_t relu_mask_ld;
    float *d_dw;        /* [N,K]  (backward_weight) */
    int64_t dw_ld;
    float *d_db;        /* [N] or NULL (backward_weight) */
    int32_t gemm_mode;  /* DLRM_GEMM_* */
    /* DLRM_GEMM_F16X2 only, each may be NULL (see dlrm_linear): max |dz|, max |W| (backward_data), max |x|
     * (backward_weight); max |dx| is folded into *d_dx_absmax (backward_data) */
    const float *d_dz_absmax, *d_w_absmax, *d_x_absmax;
    float *d_dx_absmax;
    /* DLRM_GEMM_F16X2 only, may be NULL: the pre-split planes of d_w (backward_data; see dlrm_linear) */
    const dlrm_weight_planes *w_planes;
} dlrm_linear_bwd;

int dlrm_linear_backward_data(const dlrm_linear_bwd *l, dlrm_stream_t stream);
int dlrm_linear_backward_weight_workspace_bytes(int64_t M, int32_t N, int32_t K, size_t *bytes);
int dlrm_linear_backward_weight(const dlrm_linear_bwd *l, void *d_workspace, size_t workspace_bytes,
                                dlrm_stream_t stream);
/*
 * dlrm_linear_backward_weight for callers that cannot allocate device memory (code under HIP graph capture,
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels, same
 * results bit for bit; the partial sums of the split reduction go to a DLRM_STATIC_SCRATCH_BYTES array that is
 * part of the library's device code (every GPU the library runs on holds one until the process ends) instead of
 * a caller workspace.  The call allocates and frees nothing; its first call on a device looks the array's
 * address up (hipGetSymbolAddress), so make one call before starting a capture.  A layer whose
 * dlrm_linear_backward_weight_workspace_bytes() exceeds the array is refused with DLRM_ERR_INVALID (every
 * layer of the Criteo-Kaggle and MLPerf models fits, at any batch).  All calls on one GPU share the array:
 * calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream do not.
 * Nothing in the array has to be zero between calls: every partial sum that is read was written by the same call.
 */
#define DLRM_STATIC_SCRATCH_BYTES (40u << 20)
int dlrm_linear_backward_weight_static(const dlrm_linear_bwd *l, dlrm_stream_t stream);

/*
 * w -= lr * g on a dense tensor.  Replaces torch::optim::SGD::step
 * (src/dlrmtest.cpp:505-506, :533); the reference builds a new optimizer every
 * batch, so its momentum buffer never carries over and the step is plain SGD.
 */
int dlrm_sgd_step(float *d_w, const float *d_g, int64_t n, float lr, dlrm_stream_t stream);

/* ------------------------------------------------- interaction and losses */

/*
 * Pairwise dot interaction (absent from the reference, whose interaction is
 * torch::cat, src/dlrmtest.cpp:265; required for the Criteo shapes):
 *   x [B, F, D] (feature 0 = bottom-MLP output, 1..F-1 = pooled embeddings)
 *   out[b] = [ x[b][0][0..D) , <x[b][i], x[b][j]> for i in 1..F-1, j in 0..i-1 ]
 *   out row length D + F(F-1)/2, row stride out_ld (pad columns are zeroed up
 *   to out_ld so the result can feed dlrm_linear_forward directly).
 * backward: dx = d(dense copy) + (S + S^T) x with S the strict lower triangle
 * of the pair gradients; if relu_mask_feature0 != 0 the gradient of feature 0
 * is masked with (x[b][0] > 0), which makes it the bottom MLP's last dz.
 */
int dlrm_interact_dot_forward(const float *d_x, int64_t B, int32_t F, int32_t D, float *d_out, int64_t out_ld,
                              dlrm_stream_t stream);
int dlrm_interact_dot_backward(const float *d_x, const float *d_dout, int64_t dout_ld, int64_t B, int32_t F,
                               int32_t D, float *d_dx, int32_t relu_mask_feature0, dlrm_stream_t stream);

/*
 * The dot interaction's forward pass with the EmbeddingBag lookup folded in, for batches whose bags hold exactly
 * ONE id each (Criteo's categorical features are one-hot): feature f >= 1 of sample b is row
 * indices[(f-1) batch + b] of table f-1, read straight from the tables; feature 0 is d_x0 + b x0_ld (the bottom
 * MLP's output).  The same result, bit for bit, as dlrm_embbag_forward into x[:, 1:, :] followed by
 * dlrm_interact_dot_forward on x, without writing and re-reading the [B, T, dim] block of pooled vectors
 * (create_emb + emb::forward + the concat, src/dlrmtest.cpp:53-86, :170-206, :265, in one pass).  `e` describes
 * tables and ids as for dlrm_embbag_forward: float32 rows, whole tables (no d_shard_row_begin), dim a power of two,
 * F = num_tables + 1 <= 32; pooling is irrelevant (one id).  Every bag is checked on the device: offsets[g] must
 * equal g and the id must lie inside its table; a violation sets *d_error_flag (2: a bag that is not one id, 1: id
 * out of range) and the lookup counts as a zero row.
 * If d_x_copy != NULL the kernel also writes the gathered block x [B, F, dim] (dense) there, for a backward pass
 * through dlrm_interact_dot_backward (d_x_copy may be the buffer d_x0 points into: feature 0 is rewritten with
 * itself); NULL (inference) saves that write.
 */
int dlrm_interact_dot_forward_gather(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, float *d_out, int64_t out_ld,
                                     float *d_x_copy, dlrm_stream_t stream);

/*
 * The backward pass of dlrm_interact_dot_forward_gather with the same lookup folded in: the rows are gathered from the
 * tables again (the caller has not updated them since the forward pass) instead of being read from a copy that the
 * forward pass would have to write (d_x_copy = NULL there: F * dim * 4 bytes per sample less to write, the same bytes
 * to read here).  `e`, d_x0 / x0_ld as in the forward call; d_dout [B, dout_ld] is the gradient of the forward output,
 * d_dx [B, F, dim] (dense) receives the gradient of every feature -- bit for bit what dlrm_interact_dot_backward writes
 * when given the gathered block -- feature 0's zeroed where d_x0 <= 0 if relu_mask_feature0 != 0
 * (src/dlrmtest.cpp:265-267 concatenates a ReLU output).  The ids are checked as in the forward pass and a
 * violation counts as a zero row.  The reference's autograd differentiates the concat and the gathered rows
 * separately; it has no fused counterpart.
 */
int dlrm_interact_dot_backward_gather(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, const float *d_dout,
                                      int64_t dout_ld, float *d_dx, int32_t relu_mask_feature0, dlrm_stream_t stream);

/*
 * The dot interaction on features that do not sit in one [B, F, dim] block: feature f of sample b is dim floats at
 * views[f].d_base + b * views[f].stride (floats).  A sharded rank receives the pooled vectors as one block per source
 * rank and returns their gradients the same way: with views the interaction reads / writes those exchange buffers
 * in place and the unpack / pack passes between exchange and interaction disappear (the reference is one process and
 * has no counterpart; its concat is src/dlrmtest.cpp:265).  Same arithmetic and results as dlrm_interact_dot_forward /
 * _backward on the assembled block.  dim must be a power of two >= 8, F <= 32, every base 16-byte aligned, every
 * stride a multiple of 4 floats; the view arrays are host arrays of F entries.  _backward writes the gradient of
 * feature f at dx_views[f] (a sample's gradients may go back to where its features were read from).
 */
typedef struct dlrm_feature_view {
    float *d_base;
    int64_t stride;
} dlrm_feature_view;
int dlrm_interact_dot_forward_views(const dlrm_feature_view *x_views, int64_t B, int32_t F, int32_t D, float *d_out,
                                    int64_t out_ld, dlrm_stream_t stream);
int dlrm_interact_dot_backward_views(const dlrm_feature_view *x_views, const float *d_dout, int64_t dout_ld, int64_t B,
                                     int32_t F, int32_t D, const dlrm_feature_view *dx_views, int32_t relu_mask_feature0,
                                     dlrm_stream_t stream);

/*
 * Loss forward + backward in one pass, mean reduction over n predictions.
 *   mse : torch::mse_loss (src/dlrmtest.cpp:522);  d_dpred = 2 (pred - target) / n
 *   bce : binary cross entropy with logits (MLPerf DLRM loss; not in the
 *         reference);  d_dpred = (sigmoid(pred) - target) / n
 * If relu_mask_pred != 0, d_dpred is additionally zeroed where pred <= 0 (the
 * reference applies ReLU to the last layer too, src/dlrmtest.cpp:270, so pred
 * is a ReLU output and this yields the last layer's dz).
 * d_loss receives the scalar loss (device float); preds/targets have stride 1.
 * Either of d_loss and d_dpred may be NULL.  The gradient is a chip-wide launch of a few microseconds;
 * the scalar is summed by one workgroup in a fixed order (tens of microseconds at a batch of 65536) and
 * nothing downstream needs it, so a host may ask for it in a call of its own on a side stream.
 * mean_over is the divisor of the mean: n on one GPU; the GLOBAL batch when the
 * batch is sharded over ranks, so that per-rank losses and gradients simply add
 * up to those of the whole batch.
 */
int dlrm_loss_mse(const float *d_pred, const float *d_target, int64_t n, int64_t mean_over, float *d_loss,
                  float *d_dpred, int32_t relu_mask_pred, dlrm_stream_t stream);
int dlrm_loss_bce_logits(const float *d_logit, const float *d_target, int64_t n, int64_t mean_over,
                         float *d_loss, float *d_dlogit, dlrm_stream_t stream);

/* ------------------------------------------------------- embedding cache */

/*
 * HBM-resident page cache in front of a pinned host-DRAM backing store.
 * Replaces class cache (include/data.h:242-298, src/emb_cache.cpp) and the
 * policy driver reclaim_emb_cache (src/dlrmtest.cpp:397-412); host DRAM stands
 * in for the OCSSD.  A page is rows_per_page consecutive rows of one table
 * (the reference: 4096 / (dim*4) rows, include/data.h:211; 1 gives a row
 * granular hot-row cache).  Resident pages live in num_slots fixed slots of
 * one HBM allocation which the EmbeddingBag entry points use as their weight
 * matrix; dlrm_cache_lookup returns, for every id, its row inside that matrix.
 *
 * Replacement is the reference's: per-table LRU, recency advanced once per
 * lookup call with pages of one call ordered by ascending page id (the
 * reference driver sorts ids before get_emb_vectors, src/dlrmtest.cpp:62-63,
 * which gives its std::list exactly this order).  All counters and the
 * page -> slot map are exact and reproducible (see dlrm_cache_counters).
 */
typedef struct dlrm_cache dlrm_cache;

typedef struct dlrm_cache_config {
    int32_t num_tables;          /* <= 64 */
    int32_t dim;
    int32_t row_dtype;           /* DLRM_F32 | DLRM_BF16 */
    int32_t rows_per_page;       /* cache::vector_page_offset */
    int32_t evict_group;         /* pages dropped per table per reclaim pass: nvm_dev_get_ws_opt() (8 on the */
                                 /* reference's device; src/emb_cache.cpp:220) */
    const int64_t *table_rows;   /* host int64[T] */
    void *const *h_table_store;  /* host array[T]: pinned, device-accessible [rows_t, dim] row-major store of */
                                 /* each table (dlrm_host_alloc or hipHostMalloc/hipHostRegister memory) */
    int64_t num_slots;           /* page slots in HBM: capacity + headroom for the pages one call can read */
    int64_t cache_size;          /* cache::cache_size: reclaim brings page_count down to at most this */
    void *d_slot_pool;           /* optional caller-owned HBM for the slots, num_slots*rows_per_page*dim */
                                 /* elements, 16-byte aligned; NULL = the cache allocates it */
    int64_t max_lookup_ids;      /* most ids one dlrm_cache_lookup call will carry: its buffers and those of a */
                                 /* reclaim are allocated at create and no later call allocates; 0 = they grow */
                                 /* on demand (a call may then hipMalloc / hipFree: not while another stream */
                                 /* still uses the cache) */
    int32_t writeback_dirty_only; /* 0: every evicted page is copied back to the store, like the reference */
                                 /* (src/emb_cache.cpp:259-305 has no dirty bit); 1: only pages that */
                                 /* dlrm_cache_mark_dirty has marked since they were read (inference and */
                                 /* read-mostly tables then cost no device-to-host traffic).  Residency, */
                                 /* eviction order and counters 0..5 are the same either way */
    float *const *h_row_state_store; /* NULL, or host array[T]: pinned, device-accessible float[rows_t] per table: */
                                 /* one float of optimizer state per row (row-wise Adagrad's sum of squared */
                                 /* gradients) that is read, kept and written back with the row's page.  The */
                                 /* reference's cache carries no optimizer state (its tables train with plain */
                                 /* SGD, src/dlrmtest.cpp:505-506) */
    float *d_row_state_pool;     /* optional caller-owned float[num_slots*rows_per_page]; NULL = the cache allocates */
} dlrm_cache_config;

typedef struct dlrm_cache_state_t {
    int64_t page_count;             /* cache::page_count (reference arithmetic, include/data.h:289) */
    int64_t cache_size;
    int64_t ring_head, ring_tail;   /* free-slot ring cursors (monotonic) */
    int64_t epoch;                  /* lookup calls so far */
    int64_t error_flags;            /* 1 id out of range, 2 slot pool exhausted, 4 page of the current call evicted */
    int64_t last_lookup_pages_read; /* pages the last lookup read from the store */
    int64_t total_pages, num_slots;
} dlrm_cache_state_t;

/* ---- input decoding ------------------------------------------------------------------------
 *
 * Criteo click logs in the MLPerf-DLRM binary layout (per sample: int32 label, n_dense int32 counters, then the
 * tables' hashed category ids as int32: one per table in the Criteo Terabyte files; hot[t] per table in the
 * fixed-hotness multi-hot variant, d_hot_offsets = device int32[n_sparse+1] prefix sums of hot[], ids_per_sample
 * their total), decoded on the device into the step's inputs: dense features log(max(x,0)+1) [n, dense_ld]
 * float32, labels float32, ids table-major (bag g = t'*n + b holds hot ids, taken modulo d_table_rows[t] when
 * given) and CSR offsets (n_sparse*n + 1 entries).  Output table t' is record table d_table_order[t'] (device
 * int32[n_sparse]; NULL = record order; d_table_rows and d_hot_offsets are indexed by RECORD table): with a
 * sharding plan's order the decoded ids of a local batch are, block by block, what the index exchange sends to
 * each rank.  Replaces the float-encoded feature tensor of MyDataset (src/dlrmtest.cpp:299-334), which carries
 * ids as floats and is split again by create_emb (:53-63).
 */
int dlrm_criteo_decode(const int32_t *d_records, int64_t n_samples, int32_t n_dense, int32_t n_sparse,
                       const int64_t *d_table_rows, const int32_t *d_table_order, const int32_t *d_hot_offsets,
                       int32_t ids_per_sample, float *d_dense, int64_t dense_ld, float *d_labels,
                       void *d_indices, void *d_offsets, int32_t idx_dtype, dlrm_stream_t stream);
/*
 * After the index exchange of a sharded run (the reference is one process; its closest relative is create_emb
 * taking one table's columns out of the batch, src/dlrmtest.cpp:53-63): d_recv holds, from every rank r in
 * order, the ids of this rank's n_tables tables for r's local_batch samples ([r][table][sample][hot]);
 * d_indices receives them table-major over the global batch (sample r*local_batch + b), d_offsets the CSR
 * offsets (n_tables*world*local_batch + 1), each raised by offset_base (d_indices may be the tail of a longer
 * index array whose head holds other tables).  d_hot_offsets: device int32[n_tables+1] prefix sums of these
 * tables' hotness IN THE ORDER THE TABLES HAVE INSIDE d_recv (this rank's local-table order, not the record order
 * dlrm_criteo_decode indexes its array by), NULL = one id per bag; ids_per_sample = their total (n_tables when
 * NULL).  With 32-bit ids the largest offset, offset_base + ids_per_sample * world * local_batch, must stay below
 * 2^31 (refused otherwise: it would wrap into negative CSR offsets).
 */
int dlrm_ids_regroup(const void *d_recv, int32_t world, int64_t local_batch, int32_t n_tables,
                     const int32_t *d_hot_offsets, int64_t ids_per_sample, int64_t offset_base, void *d_indices,
                     void *d_offsets, int32_t idx_dtype, dlrm_stream_t stream);

/* ---- recorded step ------------------------------------------------------------------------
 *
 * No counterpart in the reference (it re-creates modules and an autograd graph for every batch,
 * src/dlrmtest.cpp:473-545).  The launches of a step are described once -- dlrm_plan_add_* take the
 * same arguments as the calls above and copy the descriptors -- and replayed per batch:
 * dlrm_plan_set_batch patches the index / offset pointers of every EmbeddingBag entry, dlrm_plan_run
 * replays entries [first, last) on a stream (a sharded step replays one range per stretch between
 * two collectives).  Every add returns the entry's number (>= 0), or -1 with the message set.  EmbeddingBag entries take the position of their first offset
 * inside the batch's offset array instead of pointers (indices are addressed absolutely by the
 * offsets).  Optional per-entry timing with events on the replay stream: dlrm_plan_enable_timing
 * (max_runs replays of an entry between two collects; 0 switches it off), dlrm_plan_collect_timing
 * waits for the recorded events and returns milliseconds summed per entry and the replay counts.
 */
typedef struct dlrm_plan dlrm_plan;
enum {
    DLRM_OP_EMBBAG_FORWARD = 0, DLRM_OP_EMBBAG_BACKWARD_UPDATE = 1, DLRM_OP_LINEAR_FORWARD = 2,
    DLRM_OP_LINEAR_BACKWARD_DATA = 3, DLRM_OP_LINEAR_BACKWARD_WEIGHT = 4, DLRM_OP_SGD_STEP = 5,
    DLRM_OP_INTERACT_DOT_FORWARD = 6, DLRM_OP_INTERACT_DOT_BACKWARD = 7, DLRM_OP_LOSS = 8, DLRM_OP_COPY_2D = 9,
    DLRM_OP_EMBBAG_BACKWARD_PREPARE = 10, DLRM_OP_EMBBAG_BACKWARD_APPLY = 11
};
int dlrm_plan_create(dlrm_plan **out);
int dlrm_plan_destroy(dlrm_plan *p);
int dlrm_plan_size(const dlrm_plan *p);
int dlrm_plan_op(const dlrm_plan *p, int entry);
int dlrm_plan_set_batch(dlrm_plan *p, const void *d_indices, const void *d_offsets, int64_t n_indices);
int dlrm_plan_add_embbag_forward(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first, float *d_out,
                                 int64_t out_ld);
int dlrm_plan_add_embbag_backward_update(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first,
                                         const float *d_grad, int64_t grad_ld, const dlrm_sparse_opt *opt,
                                         void *d_workspace, size_t workspace_bytes);
int dlrm_plan_add_embbag_backward_prepare(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first, void *d_workspace,
                                          size_t workspace_bytes);
int dlrm_plan_add_embbag_backward_apply(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first, const float *d_grad,
                                        int64_t grad_ld, const dlrm_sparse_opt *opt, void *d_workspace,
                                        size_t workspace_bytes);
int dlrm_plan_add_linear_forward(dlrm_plan *p, const dlrm_linear *l);
int dlrm_plan_add_linear_backward_data(dlrm_plan *p, const dlrm_linear_bwd *l);
int dlrm_plan_add_linear_backward_weight(dlrm_plan *p, const dlrm_linear_bwd *l, void *d_workspace,
                                         size_t workspace_bytes);
int dlrm_plan_add_sgd_step(dlrm_plan *p, float *d_w, const float *d_g, int64_t n, float lr);
int dlrm_plan_add_interact_dot_forward(dlrm_plan *p, const float *d_x, int64_t B, int32_t F, int32_t D, float *d_out,
                                       int64_t out_ld);
int dlrm_plan_add_interact_dot_backward(dlrm_plan *p, const float *d_x, const float *d_dout, int64_t dout_ld,
                                        int64_t B, int32_t F, int32_t D, float *d_dx, int32_t relu_mask_feature0);
int dlrm_plan_add_loss_mse(dlrm_plan *p, const float *d_pred, const float *d_target, int64_t n, int64_t mean_over,
                           float *d_loss, float *d_dpred, int32_t relu_mask_pred);
int dlrm_plan_add_loss_bce_logits(dlrm_plan *p, const float *d_logit, const float *d_target, int64_t n,
                                  int64_t mean_over, float *d_loss, float *d_dlogit);
int dlrm_plan_add_copy_2d(dlrm_plan *p, float *d_dst, int64_t dst_ld, const float *d_src, int64_t src_ld, int64_t rows,
                          int64_t cols);
int dlrm_plan_enable_timing(dlrm_plan *p, int32_t max_runs);
int dlrm_plan_run(dlrm_plan *p, int32_t first, int32_t last, dlrm_stream_t stream);
int dlrm_plan_collect_timing(dlrm_plan *p, double *h_ms_sum, int32_t *h_runs);

/* Pinned, device-mapped host memory for backing stores (hipHostMalloc). */
int dlrm_host_alloc(void **h_ptr, size_t bytes);
int dlrm_host_free(void *h_ptr);

/* cache::cache(table_num, size, dimention) (include/data.h:250).  Allocates device memory; synchronises. */
int dlrm_cache_create(const dlrm_cache_config *cfg, dlrm_cache **out);
int dlrm_cache_destroy(dlrm_cache *c);
/* The slot pool as an embedding matrix: [num_slots*rows_per_page, dim] (replaces cache::get_vec(), :268). */
int dlrm_cache_storage(dlrm_cache *c, void **d_data, int64_t *total_rows);
/* The state of the pool's rows, float[num_slots*rows_per_page] indexed like the rows dlrm_cache_lookup returns (the
 * d_opt_state of dlrm_sparse_opt for updates inside the pool); NULL without h_row_state_store. */
int dlrm_cache_row_state(dlrm_cache *c, float **d_state);
/*
 * cache::get_emb_vectors for all tables of a batch (include/data.h:247,
 * src/emb_cache.cpp:358-371): ids of table t are d_ids[h_table_offsets[t] ..
 * h_table_offsets[t+1]).  Missing pages are read from the store into free
 * slots (unique pages, ascending), every referenced page becomes most
 * recently used, and d_rows_out[i] (same dtype as the ids) is the row of id i
 * in the slot pool (-1 for an id out of range).  Equals the reference when
 * ids are sorted per table, as its driver passes them.
 */
int dlrm_cache_lookup(dlrm_cache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                      void *d_rows_out, dlrm_stream_t stream);
/*
 * reclaim_emb_cache (src/dlrmtest.cpp:397-403): while page_count > cache_size,
 * every table drops its evict_group least recently used pages
 * (cache::Reclaim_vector_page, include/data.h:267); dropped pages are written
 * back to the store.  Reads one small scalar block back (synchronises).
 */
int dlrm_cache_reclaim(dlrm_cache *c, dlrm_stream_t stream);
/*
 * The same in two halves, so that the host never waits for the device in a pipelined loop (the
 * reference's reclaim_emb_cache is synchronous, src/dlrmtest.cpp:397-403).  _begin queues the choice of
 * victims and an asynchronous read-back of their count; _finish waits for that read-back only, then
 * queues write-back and eviction.  Between the two the host may queue anything that does not call into
 * this cache.  Both halves go to the same stream.
 */
int dlrm_cache_reclaim_begin(dlrm_cache *c, dlrm_stream_t stream);
int dlrm_cache_reclaim_finish(dlrm_cache *c, dlrm_stream_t stream);
/* Write every resident page back to the store without dropping it. */
int dlrm_cache_flush(dlrm_cache *c, dlrm_stream_t stream);
/*
 * cache::Reclaim_vector_page(table_ID) (include/data.h:267, src/emb_cache.cpp:208-316): the evict_group least
 * recently used pages of ONE table leave, whatever page_count is; page_count -= evict_group and the table's
 * write_page_count += evict_group as in the reference (:312, :315), also when the table holds fewer pages.
 * Synchronises like dlrm_cache_reclaim.
 */
int dlrm_cache_reclaim_table(dlrm_cache *c, int32_t table, dlrm_stream_t stream);
/*
 * Pages whose last use is fewer than `age` lookup calls old are never chosen as victims by the reclaims that
 * follow; a table with fewer older pages than a reclaim wants gives what it has (page_count and
 * write_page_count still move by evict_group per pass, the reference's arithmetic).  0 (default) is the
 * reference: any page may go.  A pipelined host that looks batch n+1 up BEFORE it reclaims for batch n
 * (the reverse of src/dlrmtest.cpp:484-494) sets 1: the rows batch n+1 will pool from must stay.
 */
int dlrm_cache_set_protect_age(dlrm_cache *c, int32_t age);
/*
 * Victims travel to the store BESIDE the calls that follow a reclaim instead of inside it (the reference writes a
 * victim page in Reclaim_vector_page, src/emb_cache.cpp:259-305, before get_emb_vectors reads the next batch's
 * misses, :451-512: one direction of the link at a time).  max_pages > 0: a reclaim with at most that many victims
 * (passes * evict_group * tables) parks them in an HBM area of max_pages pages -- its slots are free at once -- and a
 * stream of the cache's own writes the area to the store; the next lookup brings in its misses meanwhile, except
 * pages that are still on their way (flagged per page), which it fills once that write-back has ended.  Larger
 * reclaims write back inside the call as before.  dlrm_cache_flush and dlrm_cache_destroy wait for the write-back,
 * so the store is whole behind a flush; a host that reads the store directly must flush first.  Residency, eviction
 * order, slot assignment and all counters are the same in both modes.  max_pages = 0 (the default) turns it off.
 * Synchronises (stream and the cache's own); allocates.  All calls of one cache go to one stream, as before.
 */
int dlrm_cache_set_async_writeback(dlrm_cache *c, int64_t max_pages, dlrm_stream_t stream);
/*
 * The slot-pool rows d_rows[0..n) (what dlrm_cache_lookup returned; negative entries are skipped) are about to
 * be updated: their pages will be copied back when they leave the cache.  Stands in for the reference's
 * unconditional write-back of every evicted page; only looked at when writeback_dirty_only is set.
 */
int dlrm_cache_mark_dirty(dlrm_cache *c, const void *d_rows, int32_t idx_dtype, int64_t n, dlrm_stream_t stream);
/*
 * h_out[t*DLRM_CACHE_NUM_COUNTERS + k], k = 0 pages read (write_buffer::read_page_count), 1 pages
 * written (write_page_count, advances by evict_group per table per pass as in
 * src/emb_cache.cpp:315), 2 ids that hit, 3 ids that missed, 4 pages evicted,
 * 5 pages resident, 6 pages copied back to the store on eviction (= k 4 unless writeback_dirty_only).
 * reset_read_write zeroes k = 0,1 afterwards
 * (cache::read_page / write_page, include/data.h:274-285).  Synchronises.
 */
#define DLRM_CACHE_NUM_COUNTERS 7
int dlrm_cache_counters(dlrm_cache *c, int64_t *h_out, int32_t reset_read_write, dlrm_stream_t stream);
int dlrm_cache_state(dlrm_cache *c, dlrm_cache_state_t *h_state, dlrm_stream_t stream);
/* cache_size = v (the reference freezes it to page_count, src/dlrmtest.cpp:496-498). */
int dlrm_cache_set_cache_size(dlrm_cache *c, int64_t cache_size, dlrm_stream_t stream);
/* Slot of every page of one table (-1 = not resident) into host int32[pages of the table]. Synchronises. */
int dlrm_cache_export_page_map(dlrm_cache *c, int32_t table, int32_t *h_out, dlrm_stream_t stream);

/* ------------------------------------------------------------------------
 * Vector-granular cache over a log-structured page store: the reference's second policy.
 * Replaces class cache2 (include/data2.h:219-287, src/emb_cache2.cpp) as src/dlrmtest2.cpp drives it:
 *   - rows are resident one by one; a table lists the rows that were asked for, oldest request first
 *     (lru_page, src/emb_cache2.cpp:638-659: a hit counts but does not move the entry);
 *   - a table whose list exceeds cache_size rows evicts its oldest entries (reclaim_emb_cache,
 *     src/dlrmtest2.cpp:462-481); evicted rows collect in a buffer of evict_group * rows_per_page rows per
 *     table, a full buffer is written as evict_group new pages (Reclaim_vector_page, src/emb_cache2.cpp:397-633);
 *   - a missing row is read together with the page that holds it, and every row that page lists becomes
 *     resident (cache2::read, :729-892) -- unlisted; such rows go when a listed row of their home page
 *     (rows [p*rows_per_page, (p+1)*rows_per_page)) is evicted (:492-496).
 * The store of table t is pinned host memory of (ceil(rows/rows_per_page) + log_pages_per_table) pages: the
 * table itself, row-major, then the region the cache writes pages to.  Flash placement below the page
 * (chunks, write pointers, garbage collection: src/emb_cache2.cpp:28-285) has no counterpart.
 * ------------------------------------------------------------------------ */
typedef struct dlrm_vcache dlrm_vcache;

typedef struct dlrm_vcache_config {
    int32_t num_tables;
    int32_t dim;                      /* a row (dim elements of row_dtype) must be a multiple of 16 bytes */
    int32_t rows_per_page;            /* vector_page_offset = page bytes / row bytes (include/data2.h:161) */
    int32_t evict_group;              /* ws_opt: pages per write (the reference's device reports 8) */
    const int64_t *table_rows;        /* host int64[T] */
    void *const *h_table_store;       /* host pointers [T], from dlrm_host_alloc, sized as described above */
    int64_t log_pages_per_table;
    int64_t num_slots;                /* rows of the HBM pool (listed rows + rows that came along with a page) */
    int64_t cache_size;               /* listed rows per table (src/dlrmtest2.cpp:53, :468); >= evict_group * rows_per_page */
    void *d_slot_pool;                /* optional caller-owned [num_slots, dim] device buffer of row_dtype */
    int32_t row_dtype;                /* DLRM_F32 (the reference's rows) | DLRM_BF16 */
    float *const *h_row_state_store;  /* NULL, or host array[T] of pinned, device-accessible float[(pages of the */
                                      /* store) * rows_per_page]: one float of optimizer state per row position of */
                                      /* the store; it moves with the row (store <-> write buffer <-> slot pool). */
                                      /* No counterpart in the reference (plain SGD, src/dlrmtest2.cpp) */
    float *d_row_state_pool;          /* optional caller-owned float[num_slots]; NULL = the cache allocates it */
} dlrm_vcache_config;

typedef struct dlrm_vcache_state_t {
    int64_t page_count;               /* cache2::page_count: only ever lowered, by evict_group per write (:614) */
    int64_t cache_size;
    int64_t free_slots;
    int64_t epoch;                    /* lookups so far */
    int64_t error_flags;              /* 1 id out of range, 2 slot pool exhausted, 8 a table's written region is full */
    int64_t last_lookup_pages_read;
    int64_t num_slots;
    int64_t total_rows;
} dlrm_vcache_state_t;

/* per table: pages read, pages written (both reset by dlrm_vcache_counters(reset) like cache2::read_page /
 * write_page, include/data2.h:263-274), ids found resident, ids not, rows evicted, rows resident now, rows
 * listed now, unlisted rows dropped, rows waiting in the write buffer, pages written in total */
#define DLRM_VCACHE_NUM_COUNTERS 10

int dlrm_vcache_create(const dlrm_vcache_config *cfg, dlrm_vcache **out);
int dlrm_vcache_destroy(dlrm_vcache *c);
/* The slot pool [num_slots, dim]: the weight matrix for dlrm_embbag (d_table_row_offsets = NULL). */
int dlrm_vcache_storage(dlrm_vcache *c, void **d_data, int64_t *total_rows);
/* The state of the pool's rows, float[num_slots] (d_opt_state for updates inside the pool); NULL without a state store. */
int dlrm_vcache_row_state(dlrm_vcache *c, float **d_state);
/* cache2::get_emb_vectors (src/emb_cache2.cpp:686-699) for all tables of a batch: ids of table t at
 * [h_table_offsets[t], h_table_offsets[t+1]), in the caller's order (dlrmtest2 does not sort them,
 * src/dlrmtest2.cpp:63-67); d_rows_out[i] = row of id i in the slot pool (same integer type as the ids). */
int dlrm_vcache_lookup(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                       void *d_rows_out, dlrm_stream_t stream);
/* reclaim_emb_cache of src/dlrmtest2.cpp:462-481.  Synchronises once (the number of listed rows sizes a sort). */
int dlrm_vcache_reclaim(dlrm_vcache *c, dlrm_stream_t stream);
/* save_emb (src/dlrmtest2.cpp:169-177): rows of the batch that the reclaim has dropped are made resident
 * again, unlisted, with their current values.  d_rows_out optional (as in lookup). */
int dlrm_vcache_restore(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                        void *d_rows_out, dlrm_stream_t stream);
/* d_out[i][dim] (row_dtype) = current value of id i, wherever it lives (slot pool, write buffer, store), and, when
 * d_state_out is not NULL and rows carry state, d_state_out[i] = its state.  Changes nothing in the cache. */
int dlrm_vcache_read_rows(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                          void *d_out, float *d_state_out, dlrm_stream_t stream);
/* h_out: host int64[T][DLRM_VCACHE_NUM_COUNTERS].  Synchronises. */
int dlrm_vcache_counters(dlrm_vcache *c, int64_t *h_out, int32_t reset_read_write, dlrm_stream_t stream);
int dlrm_vcache_state(dlrm_vcache *c, dlrm_vcache_state_t *h_state, dlrm_stream_t stream);
int dlrm_vcache_set_cache_size(dlrm_vcache *c, int64_t cache_size, dlrm_stream_t stream);
/* Per row of one table (any output may be NULL; host int32[rows]): slot (-1 = not resident), page that holds
 * the stored value (initial pages first, then written pages in order), position in the write buffer (-1 =
 * not waiting).  Synchronises. */
int dlrm_vcache_export_table(dlrm_vcache *c, int32_t table, int32_t *h_row_slot, int32_t *h_row_page,
                             int32_t *h_row_pending, dlrm_stream_t stream);
/* Per slot (host arrays of num_slots entries, any may be NULL): global row, list key (0 = not listed, else
 * (epoch << 32) | position of the first request: smaller = older), hit count.  Synchronises. */
int dlrm_vcache_export_slots(dlrm_vcache *c, int32_t *h_slot_row, uint64_t *h_slot_key, int32_t *h_slot_count,
                             dlrm_stream_t stream);

/* ------------------------------------------------------------------------
 * Collectives of the sharded step (RCCL over xGMI), for hosts without a Python runtime.  No counterpart in the
 * reference, which runs on one host; the placement follows its striping of tables over parallel units
 * (include/data2.h:150, src/emb_cache2.cpp:701-709): whole tables per rank -> all-to-all of the pooled vectors
 * [B_local, tables of the sender, dim] and of their gradients; row ranges per rank
 * (dlrm_embbag.d_shard_row_begin) -> reduce-scatter of partial sums, all-gather of gradients; dense
 * gradients -> all-reduce.  One communicator per process, created with the HIP device current that its
 * calls will use; calls are enqueued on the given stream; every rank makes the same calls in the same order.
 * ------------------------------------------------------------------------ */
typedef struct dlrm_comm dlrm_comm;
#define DLRM_COMM_ID_BYTES 128
/* Rank 0 draws an id and hands it to the other ranks by any means (file, environment, socket). */
int dlrm_comm_unique_id(uint8_t *id /* [DLRM_COMM_ID_BYTES] */);
/* An id that selects the rehearsal transport instead of RCCL: the ranks are processes of one machine that meet in
 * shared host memory, every call drains its stream and is carried out on the host (mailboxes of 64 MiB per rank,
 * DLRM_COMM_REHEARSAL_MB changes that; at most 16 ranks).  Same calls, same results (sums in rank order); for
 * checking a multi-rank host on a box with fewer GPUs than ranks -- RCCL refuses two ranks on one device. */
int dlrm_comm_unique_id_rehearsal(uint8_t *id /* [DLRM_COMM_ID_BYTES] */);
int dlrm_comm_init(const uint8_t *id, int32_t rank, int32_t world, dlrm_comm **out);
int dlrm_comm_destroy(dlrm_comm *c);
/* d_buf[i] = sum over ranks, in place (the flat MLP gradient buffer). */
int dlrm_comm_all_reduce_sum(dlrm_comm *c, float *d_buf, int64_t n, dlrm_stream_t stream);
/* Floats for peer p are h_send_counts[p] consecutive elements of d_send (peers in rank order); what peer p
 * sends arrives as h_recv_counts[p] consecutive elements of d_recv.  The exchange moves 4-byte words without
 * looking at them (as does dlrm_comm_all_gather): the index exchange passes its int32 ids through the same calls
 * (int64 ids: twice the counts). */
int dlrm_comm_all_to_all(dlrm_comm *c, const float *d_send, const int64_t *h_send_counts, float *d_recv,
                         const int64_t *h_recv_counts, dlrm_stream_t stream);
/* d_recv[i] = sum over ranks of their d_send[rank * recv_count + i]. */
int dlrm_comm_reduce_scatter_sum(dlrm_comm *c, const float *d_send, float *d_recv, int64_t recv_count,
                                 dlrm_stream_t stream);
/* d_recv[p * send_count + i] = rank p's d_send[i]. */
int dlrm_comm_all_gather(dlrm_comm *c, const float *d_send, float *d_recv, int64_t send_count, dlrm_stream_t stream);

/* ------------------------------------------------------------------------
 * The whole model behind one handle: what a C or C++ host needs to run the train / inference step of the
 * Criteo-shaped models without assembling it from the operators above.  Stands for the loop body of the
 * reference's main() (src/dlrmtest.cpp:473-545: build emb, forward, loss, backward, optimizer step,
 * update_parameter) with struct emb (:144-210) and struct Net2 (:217-278) inside.  One handle per process and
 * GPU; with world > 1 (or force_exchange) the tables are sharded over the ranks of `comm` -- whole tables by
 * table_owner[t] = rank (pooled vectors travel with an all-to-all), tables with owner -1 split into one
 * contiguous row range per rank (reduce-scatter / all-gather) -- and the MLPs are data parallel (all-reduce
 * of the flat gradient buffer).  Every rank passes the ids of ITS tables (dlrm_model_local_table order: whole
 * tables ascending, then the row-wise ones) for the GLOBAL batch, table-major with include_last_offset
 * offsets, as dlrm_ids_regroup produces them.  The step's kernels run on the caller's stream and up to three
 * streams of the model; on return everything the step depends on is ordered behind the caller's stream.
 * ------------------------------------------------------------------------ */
typedef struct dlrm_model dlrm_model;
enum { DLRM_INTERACT_DOT = 0, DLRM_INTERACT_CAT = 1 };   /* cat: torch::cat of src/dlrmtest.cpp:265 */
enum { DLRM_LOSS_BCE = 0, DLRM_LOSS_MSE = 1 };           /* mse: torch::mse_loss of :522 */

typedef struct dlrm_model_config {
    int32_t num_tables;
    const int64_t *table_rows;   /* host int64[T] */
    int32_t dim;
    int32_t dense_in;
    int32_t n_bottom;
    const int32_t *bottom;       /* host: widths of the bottom MLP's layers (all ReLU); dot needs the last == dim */
    int32_t n_top;
    const int32_t *top;          /* host: widths of the top MLP's layers, the last one 1 */
    int32_t interaction;         /* DLRM_INTERACT_* */
    int32_t loss;                /* DLRM_LOSS_* */
    int32_t final_relu;          /* ReLU on the last layer too, as the reference has it (:270) */
    int32_t pooling;             /* DLRM_POOL_* */
    int32_t row_dtype;           /* DLRM_F32 | DLRM_BF16 */
    int32_t idx_dtype;           /* DLRM_I32 | DLRM_I64 */
    int32_t emb_optimizer;       /* DLRM_OPT_* */
    int32_t gemm_mode;           /* DLRM_GEMM_* */
    float lr, eps;
    int64_t local_batch;         /* samples whose MLP runs on this rank */
    int32_t rank, world;
    dlrm_comm *comm;             /* required when world > 1 or force_exchange */
    const int32_t *table_owner;  /* host int32[T]: owning rank, -1 = split by rows; NULL = whole tables placed */
                                 /* greedily (biggest first, to the rank with the fewest tables, then bytes) */
    int32_t force_exchange;      /* run the sharded schedule (every exchange, with one peer) even when world == 1 */
    int32_t overlap;             /* embedding side on its own stream: 1 / 0, -1 = when sharded or 8192 <= local_batch <= 32768 */
    int32_t presort;             /* sort the sparse update's occurrences beside the forward pass (own stream) */
    int32_t wgrad_stream;        /* weight gradients on their own stream: 1 / 0, -1 = when local_batch <= 32768 */
    uint64_t seed;               /* tables: uniform [0,1), table t seeded (seed << 20) + 7919 (t + 1) */
    int32_t init_tables;         /* 0 = leave the table memory as allocated (the caller fills it) */
    int32_t one_id_per_bag;      /* promise: every bag of every batch holds exactly one id (one-hot features).  A */
                                 /* single-GPU model with the dot interaction and float32 rows then reads the rows */
                                 /* inside the interaction's forward kernel (dlrm_interact_dot_forward_gather) instead */
                                 /* of pooling them into a [B, T, dim] block first; a batch that breaks the promise raises */
                                 /* *d_error_flag (2) and its offending bags count as zero rows */
    int32_t graph_steps;         /* 1: dlrm_model_train_step captures its launches, stream forks and joins into a HIP */
                                 /* graph the first time it sees a set of batch buffers (d_indices, d_offsets, */
                                 /* n_indices, stream) and replays the graph afterwards (the 8 most recent sets are */
                                 /* kept): for hosts that cycle resident buffers.  The caller's stream must not be the */
                                 /* NULL stream.  Sharded models and steps with operator timing launch directly. */
    dlrm_comm *comm_dense;       /* NULL, or a SECOND communicator over the same ranks for the all-reduce of the dense */
                                 /* gradients: RCCL serialises the operations of one communicator, so only with its own */
                                 /* one does that all-reduce run beside the backward exchange of the embedding side.  The */
                                 /* top MLP's gradients are then reduced as soon as that stack's backward pass is done */
                                 /* (beside the interaction's and the bottom MLP's), the bottom MLP's at the end */
} dlrm_model_config;

typedef struct dlrm_model_info_t {
    float *d_params, *d_grads;   /* flat dense parameters / gradients: see dlrm_model_layer */
    int64_t n_params;
    void *d_weight;              /* [total_rows, dim]: this rank's tables back to back */
    int64_t total_rows;
    float *d_opt_state;          /* Adagrad: float[total_rows], else NULL */
    int32_t num_local_tables, num_whole_tables;
    float *d_pred;               /* [local_batch] predictions (logits for DLRM_LOSS_BCE) of the last step */
    float *d_loss;               /* this rank's share of the global-mean loss of the last train step */
    int32_t *d_error_flag;       /* raised by the EmbeddingBag kernels on an out-of-range id */
    float *d_dense;              /* [local_batch, dense_ld] input buffer (dlrm_model_load_batch fills it) */
    int64_t dense_ld;
    float *d_labels;             /* [local_batch] */
    int32_t n_bottom, n_top;
} dlrm_model_info_t;

int dlrm_model_create(const dlrm_model_config *cfg, dlrm_model **out);
int dlrm_model_destroy(dlrm_model *m);
int dlrm_model_info(const dlrm_model *m, dlrm_model_info_t *info);
/* local table i (0 .. num_local_tables): its global table id, first row inside d_weight, rows held, and the
 * table row its first held row is (0 for whole tables).  Any output may be NULL. */
int dlrm_model_local_table(const dlrm_model *m, int32_t local_table, int32_t *global_table, int64_t *first_pool_row,
                           int64_t *rows, int64_t *first_table_row);
/* order[f] = table whose pooled vector is feature f + 1 of the interaction (whole tables rank-major, then the
 * row-wise ones): the table order of the top MLP's first-layer columns, and what dlrm_criteo_decode's
 * d_table_order wants for the index exchange.  host int32[T]. */
int dlrm_model_feature_order(const dlrm_model *m, int32_t *order);
/* stack 0 = bottom, 1 = top.  Layer `layer`: W [N, K] at d_params + w_offset with row stride w_ld, bias [N] at
 * d_params + b_offset (gradients at the same offsets of d_grads). */
int dlrm_model_layer(const dlrm_model *m, int32_t stack, int32_t layer, int32_t *N, int32_t *K, int64_t *w_offset,
                     int64_t *w_ld, int64_t *b_offset);
int dlrm_model_set_layer(dlrm_model *m, int32_t stack, int32_t layer, const float *h_w /* [N,K] */, const float *h_b);
int dlrm_model_get_layer(const dlrm_model *m, int32_t stack, int32_t layer, float *h_w, float *h_b);   /* synchronises */
/* Normal(0, 2/(K+N)) weights and Normal(0, 1/N) biases from one seed (give every rank the same seed); the
 * reference binds unscaled randn weights (src/dlrmtest.cpp:110-113). */
int dlrm_model_init_dense(dlrm_model *m, uint64_t seed);
/* copies a batch's dense features [local_batch, dense_in] (row stride dense_ld) and labels into the model */
int dlrm_model_load_batch(dlrm_model *m, const float *d_dense, int64_t dense_ld, const float *d_labels, dlrm_stream_t stream);
/* Net2::forward around emb::forward (src/dlrmtest.cpp:254-272, :170-206): predictions into d_pred */
int dlrm_model_forward(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices, dlrm_stream_t stream);
/* one iteration of the loop body (src/dlrmtest.cpp:505-535): forward, loss, backward, SGD on every dense
 * parameter, fused sparse update of the rows.  The first step of a new batch size may allocate its sort
 * workspace; later steps allocate nothing. */
int dlrm_model_train_step(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices,
                          dlrm_stream_t stream);

/*
 * Optional timing of the step's operators, for benchmarks: every launch of a class whose bit is set in
 * class_mask is bracketed by two events on the stream it runs on (at most max_records launches between two
 * collects; further ones run untimed).  dlrm_model_collect_timing synchronises and returns milliseconds and
 * launch counts summed per class (arrays of DLRM_TIME_NUM_CLASSES), then starts over.  The "_SKINNY" classes
 * are the layers that are no matrix product (at most 16 inputs, or one output).  class_mask 0 switches it off.
 */
enum {
    DLRM_TIME_LINEAR_FORWARD = 0, DLRM_TIME_LINEAR_BACKWARD_DATA, DLRM_TIME_LINEAR_BACKWARD_WEIGHT,
    DLRM_TIME_LINEAR_FORWARD_SKINNY, DLRM_TIME_LINEAR_BACKWARD_DATA_SKINNY, DLRM_TIME_LINEAR_BACKWARD_WEIGHT_SKINNY,
    DLRM_TIME_EMBBAG_FORWARD, DLRM_TIME_EMBBAG_BACKWARD_PREPARE, DLRM_TIME_EMBBAG_BACKWARD_UPDATE,
    DLRM_TIME_INTERACT_DOT_FORWARD, DLRM_TIME_INTERACT_DOT_BACKWARD,
    /* the collectives of a sharded step (time on their stream, waiting for the slowest peer included) */
    DLRM_TIME_COMM_FORWARD_EXCHANGE, DLRM_TIME_COMM_BACKWARD_EXCHANGE, DLRM_TIME_COMM_ALL_REDUCE,
    /* the one launch per MLP stack that adds up the partial sums of all its layers' weight and bias gradients */
    DLRM_TIME_LINEAR_BACKWARD_WEIGHT_REDUCE,
    DLRM_TIME_NUM_CLASSES
};
int dlrm_model_enable_timing(dlrm_model *m, uint32_t class_mask, int32_t max_records);
int dlrm_model_collect_timing(dlrm_model *m, double *h_ms_sum, int64_t *h_launches);

#ifdef __cplusplus
}
#endif
#endif /* DLRM_AMD_H */
