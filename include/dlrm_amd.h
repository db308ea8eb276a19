/*
 * dlrm_amd -- C ABI of the MI355X (gfx950) DLRM hot path.
 *
 * One shared library (libdlrm_amd.so) exports everything below.  It replaces,
 * for the data-parallel hot path only, the C++ model interface the reference
 * driver is written against (dlrm_libtorch/src/dlrmtest.cpp with
 * dlrm_libtorch/include/data.h); each entry point cites the reference
 * interface it stands in for.
 *
 * Conventions
 *  - Every function returns DLRM_OK (0) or an error code; the message of the
 *    last error on the calling thread is dlrm_last_error().  Nothing throws.
 *  - Buffers named d_* are device (HBM) pointers owned by the caller; h_* are
 *    host pointers.  The library allocates nothing unless a function says so
 *    (the cache object and the handles are the only owners of device memory).
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
/* ABI version of the library: major*1000 + minor. */
int dlrm_version(void);
/* Name of the device architecture the kernels were built for ("gfx950"). */
const char *dlrm_target_arch(void);

/* ------------------------------------------------------------------ tables */

/*
 * Fill n elements (first .. first+n of a table's flat element stream) with
 * uniform [0,1): value(i) = (splitmix64(seed + i) >> 40) * 2^-24.
 * Replaces embedding_table_init::write() -> DATA::initial()
 * (include/data.h:217, src/data_transfer.cpp:47-61: rand()/(RAND_MAX+1.0)).
 * dst may be device memory or host-mapped pinned memory.
 */
int dlrm_table_init_uniform(void *dst, int row_dtype, int64_t n, uint64_t seed, int64_t first,
                            dlrm_stream_t stream);

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
    const int64_t *d_table_row_offsets; /* device int64[T+1]; NULL when T == 1 (one table, */
                                        /* rows [0, total_rows)) */
    int64_t total_rows;
    int32_t num_tables;
    int32_t dim;
    int32_t row_dtype;           /* DLRM_F32 | DLRM_BF16 */
    int32_t idx_dtype;           /* DLRM_I32 | DLRM_I64 */
    const void *d_indices;       /* idx[n_indices] */
    const void *d_offsets;       /* idx[T*batch] (+1 when include_last_offset) */
    int64_t n_indices;
    int64_t batch;
    int32_t include_last_offset;
    int32_t pooling;             /* DLRM_POOL_SUM | DLRM_POOL_MEAN */
    int32_t *d_error_flag;       /* optional device int32, set to 1 when an index is out of range */
                                 /* (such lookups are skipped); NULL = do not report */
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

#ifdef __cplusplus
}
#endif
#endif /* DLRM_AMD_H */
