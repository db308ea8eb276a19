// Decoder for Criteo click logs in the MLPerf-DLRM binary layout, and the regrouping of ids after the index
// exchange of a sharded run.
//
// Record layout: one sample = int32 label, n_dense int32 counters, then for every table its ids as int32 --
// one per table in the single-hot Criteo Terabyte files, hot[t] per table in the fixed-hotness multi-hot
// variant (MLPerf DLRM-v2 style).  The raw records of a batch are copied to the device as they are; the decode
// kernel turns them into what the train step consumes -- dense features log(max(x,0)+1) as float32
// [B, dense_ld], labels as float32, ids table-major (bag g = t'*B + b holds hot ids) reduced modulo the table's
// row count, and the CSR offsets.  Output table t' is record table order[t'], so a sharded run can ask for its
// sharding plan's order (whole tables rank-major, then the row-wise ones): the decoded ids of a LOCAL batch are
// then, block by block, what goes to each rank in the index all-to-all.  It replaces the reference's
// float-encoded synthetic feature tensor (MyDataset, src/dlrmtest.cpp:299-334), which create_emb splits by
// table again (:53-63).
#include "common.h"

namespace dlrm {

constexpr int kDecThreads = 256;
constexpr int kDecSamples = 64;   // samples per workgroup

template <typename IDX>
__global__ __launch_bounds__(kDecThreads) void criteo_decode_kernel(const int32_t *__restrict__ rec, int64_t n,
                                                                     int n_dense, int n_sparse,
                                                                     const int64_t *__restrict__ table_rows,
                                                                     const int32_t *__restrict__ order,
                                                                     const int32_t *__restrict__ hot_off, int ids_per_sample,
                                                                     float *__restrict__ dense, int64_t dense_ld,
                                                                     float *__restrict__ labels, IDX *__restrict__ idx,
                                                                     IDX *__restrict__ off) {
    extern __shared__ int32_t s_rec[];   // [kDecSamples][fields + 1]: odd stride, conflict-free transposed reads
    const int fields = 1 + n_dense + ids_per_sample;
    const int ld = fields | 1;
    const int64_t b0 = (int64_t)blockIdx.x * kDecSamples;
    const int nb = (int)min((int64_t)kDecSamples, n - b0);
    // coalesced read of the workgroup's records
    for (int i = threadIdx.x; i < nb * fields; i += kDecThreads) {
        const int s = i / fields, f = i - s * fields;
        s_rec[s * ld + f] = rec[b0 * fields + i];
    }
    __syncthreads();
    // labels and dense features: consecutive threads write consecutive floats of a sample row
    for (int i = threadIdx.x; i < nb * (1 + n_dense); i += kDecThreads) {
        const int s = i / (1 + n_dense), f = i - s * (1 + n_dense);
        const int32_t v = s_rec[s * ld + f];
        if (f == 0) labels[b0 + s] = (float)v;
        else dense[(b0 + s) * dense_ld + (f - 1)] = log1pf((float)(v > 0 ? v : 0));
    }
    // ids: consecutive threads write the consecutive ids of consecutive samples of one output table
    int64_t out_first = 0;   // ids of the output tables before t', per sample
    for (int tp = 0; tp < n_sparse; tp++) {
        const int t = order ? order[tp] : tp;
        const int first = hot_off ? hot_off[t] : t, hot = hot_off ? hot_off[t + 1] - hot_off[t] : 1;
        const int64_t rows = table_rows ? table_rows[t] : 0;
        for (int i = threadIdx.x; i < nb * hot; i += kDecThreads) {
            const int s = i / hot, j = i - s * hot;
            int64_t v = (int64_t)(uint32_t)s_rec[s * ld + 1 + n_dense + first + j];   // hashed ids are unsigned 32-bit values
            if (rows) v %= rows;
            idx[out_first * n + (b0 + s) * hot + j] = (IDX)v;
        }
        for (int s = threadIdx.x; s < nb; s += kDecThreads)
            off[(int64_t)tp * n + b0 + s] = (IDX)(out_first * n + (b0 + s) * hot);
        out_first += hot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[(int64_t)n_sparse * n] = (IDX)((int64_t)ids_per_sample * n);
}

// After the index exchange a rank holds, from every peer r, the block [its tables][B local samples][hot ids].
// The train step wants them table-major over the GLOBAL batch (sample r*B + b): block (r, t) is contiguous on
// both sides, so this is a copy of world * n_tables runs, plus the CSR offsets of the global batch.
template <typename IDX>
__global__ __launch_bounds__(kDecThreads) void ids_regroup_kernel(const IDX *__restrict__ recv, int world, int64_t B,
                                                                  int n_tables, const int32_t *__restrict__ hot_off,
                                                                  int64_t off_base, IDX *__restrict__ idx,
                                                                  IDX *__restrict__ off) {
    const int r = blockIdx.y / n_tables, t = blockIdx.y % n_tables;
    const int64_t first = hot_off ? hot_off[t] : t, hot = hot_off ? hot_off[t + 1] - hot_off[t] : 1;
    const int64_t per_sample = hot_off ? hot_off[n_tables] : n_tables;
    const int64_t Bg = B * world;
    const IDX *src = recv + r * per_sample * B + first * B;
    IDX *dst = idx + first * Bg + r * B * hot;
    for (int64_t i = (int64_t)blockIdx.x * kDecThreads + threadIdx.x; i < B * hot; i += (int64_t)gridDim.x * kDecThreads)
        dst[i] = src[i];
    for (int64_t b = (int64_t)blockIdx.x * kDecThreads + threadIdx.x; b < B; b += (int64_t)gridDim.x * kDecThreads)
        off[(int64_t)t * Bg + r * B + b] = (IDX)(off_base + first * Bg + (r * B + b) * hot);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) off[(int64_t)n_tables * Bg] = (IDX)(off_base + per_sample * Bg);
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_criteo_decode(const int32_t *d_records, int64_t n_samples, int32_t n_dense, int32_t n_sparse,
                                  const int64_t *d_table_rows, const int32_t *d_table_order, const int32_t *d_hot_offsets,
                                  int32_t ids_per_sample, float *d_dense, int64_t dense_ld, float *d_labels,
                                  void *d_indices, void *d_offsets, int32_t idx_dtype, dlrm_stream_t stream) {
    DLRM_REQUIRE(n_samples >= 0 && n_dense >= 0 && n_sparse >= 1, "bad shape");
    DLRM_REQUIRE(d_hot_offsets ? ids_per_sample >= n_sparse : ids_per_sample == n_sparse,
                 "ids_per_sample must be the sum of the tables' hotness (n_sparse without d_hot_offsets)");
    DLRM_REQUIRE(n_dense + ids_per_sample <= 4095, "record too long");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    if (n_samples == 0) return DLRM_OK;
    DLRM_REQUIRE(d_records && d_labels && d_indices && d_offsets && (n_dense == 0 || d_dense), "NULL argument");
    DLRM_REQUIRE(dense_ld >= n_dense, "dense_ld too small");
    DLRM_REQUIRE(idx_dtype == DLRM_I64 || (int64_t)ids_per_sample * n_samples < (1ll << 31), "offsets do not fit int32");
    const int fields = 1 + n_dense + ids_per_sample;
    const size_t lds = sizeof(int32_t) * kDecSamples * (size_t)(fields | 1);
    DLRM_REQUIRE(lds <= 160 * 1024, "record too long for the staging buffer");
    const unsigned blocks = (unsigned)ceil_div(n_samples, kDecSamples);
    hipStream_t s = as_stream(stream);
    if (idx_dtype == DLRM_I64) {
        if (lds > 64 * 1024)
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(criteo_decode_kernel<int64_t>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(criteo_decode_kernel<int64_t>, dim3(blocks), dim3(kDecThreads), lds, s, d_records, n_samples,
                           n_dense, n_sparse, d_table_rows, d_table_order, d_hot_offsets, ids_per_sample, d_dense, dense_ld,
                           d_labels, static_cast<int64_t *>(d_indices), static_cast<int64_t *>(d_offsets));
    } else {
        if (lds > 64 * 1024)
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(criteo_decode_kernel<int32_t>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(criteo_decode_kernel<int32_t>, dim3(blocks), dim3(kDecThreads), lds, s, d_records, n_samples,
                           n_dense, n_sparse, d_table_rows, d_table_order, d_hot_offsets, ids_per_sample, d_dense, dense_ld,
                           d_labels, static_cast<int32_t *>(d_indices), static_cast<int32_t *>(d_offsets));
    }
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_ids_regroup(const void *d_recv, int32_t world, int64_t local_batch, int32_t n_tables,
                                const int32_t *d_hot_offsets, int64_t ids_per_sample, int64_t offset_base, void *d_indices,
                                void *d_offsets, int32_t idx_dtype, dlrm_stream_t stream) {
    DLRM_REQUIRE(world >= 1 && local_batch >= 0 && n_tables >= 0, "bad shape");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    DLRM_REQUIRE(ids_per_sample >= n_tables && (d_hot_offsets != nullptr || ids_per_sample == n_tables) && offset_base >= 0,
                 "ids_per_sample %lld for %d tables", (long long)ids_per_sample, n_tables);
    DLRM_REQUIRE(idx_dtype == DLRM_I64 || offset_base + ids_per_sample * local_batch * world < (1ll << 31),
                 "offsets up to %lld do not fit 32-bit ids", (long long)(offset_base + ids_per_sample * local_batch * world));
    DLRM_REQUIRE((int64_t)world * n_tables <= 65535, "too many (rank, table) blocks");
    if (n_tables == 0 || local_batch == 0) return DLRM_OK;
    DLRM_REQUIRE(d_recv && d_indices && d_offsets, "NULL argument");
    const unsigned bx = (unsigned)(ceil_div(local_batch, kDecThreads) < 64 ? ceil_div(local_batch, kDecThreads) : 64);
    const dim3 grid(bx, (unsigned)(world * n_tables));
    hipStream_t s = as_stream(stream);
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(ids_regroup_kernel<int64_t>, grid, dim3(kDecThreads), 0, s, static_cast<const int64_t *>(d_recv),
                           world, local_batch, n_tables, d_hot_offsets, offset_base, static_cast<int64_t *>(d_indices),
                           static_cast<int64_t *>(d_offsets));
    else
        hipLaunchKernelGGL(ids_regroup_kernel<int32_t>, grid, dim3(kDecThreads), 0, s, static_cast<const int32_t *>(d_recv),
                           world, local_batch, n_tables, d_hot_offsets, offset_base, static_cast<int32_t *>(d_indices),
                           static_cast<int32_t *>(d_offsets));
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
