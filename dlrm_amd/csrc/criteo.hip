// Decoder for the Criteo click logs in the MLPerf-DLRM binary layout: one sample = int32 label,
// n_dense int32 counters, n_sparse int32 category ids.  The raw records of a batch are copied to the
// device as they are; this kernel turns them into what the train step consumes -- dense features
// log(max(x,0)+1) as float32 [B, dense_ld], labels as float32, ids table-major (bag g = t*B + b, one id
// per bag) reduced modulo the table's row count, and the offsets 0..T*B.  It replaces the reference's
// float-encoded synthetic feature tensor (MyDataset, src/dlrmtest.cpp:299-334).
#include "common.h"

namespace dlrm {

constexpr int kDecThreads = 256;
constexpr int kDecSamples = 64;   // samples per workgroup

template <typename IDX>
__global__ __launch_bounds__(kDecThreads) void criteo_decode_kernel(const int32_t *__restrict__ rec, int64_t n,
                                                                     int n_dense, int n_sparse,
                                                                     const int64_t *__restrict__ table_rows,
                                                                     float *__restrict__ dense, int64_t dense_ld,
                                                                     float *__restrict__ labels, IDX *__restrict__ idx,
                                                                     IDX *__restrict__ off) {
    extern __shared__ int32_t s_rec[];   // [kDecSamples][fields + 1]: odd stride, conflict-free transposed reads
    const int fields = 1 + n_dense + n_sparse;
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
    // ids: consecutive threads write consecutive samples of one table
    for (int i = threadIdx.x; i < n_sparse * kDecSamples; i += kDecThreads) {
        const int t = i / kDecSamples, s = i - t * kDecSamples;
        if (s >= nb) continue;
        int64_t v = (int64_t)(uint32_t)s_rec[s * ld + 1 + n_dense + t];   // hashed ids are unsigned 32-bit values
        if (table_rows) v %= table_rows[t];
        idx[(int64_t)t * n + b0 + s] = (IDX)v;
        off[(int64_t)t * n + b0 + s] = (IDX)((int64_t)t * n + b0 + s);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[(int64_t)n_sparse * n] = (IDX)((int64_t)n_sparse * n);
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_criteo_decode(const int32_t *d_records, int64_t n_samples, int32_t n_dense, int32_t n_sparse,
                                  const int64_t *d_table_rows, float *d_dense, int64_t dense_ld, float *d_labels,
                                  void *d_indices, void *d_offsets, int32_t idx_dtype, dlrm_stream_t stream) {
    DLRM_REQUIRE(n_samples >= 0 && n_dense >= 0 && n_sparse >= 1 && n_dense + n_sparse <= 1023, "bad shape");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    if (n_samples == 0) return DLRM_OK;
    DLRM_REQUIRE(d_records && d_labels && d_indices && d_offsets && (n_dense == 0 || d_dense), "NULL argument");
    DLRM_REQUIRE(dense_ld >= n_dense, "dense_ld too small");
    DLRM_REQUIRE(idx_dtype == DLRM_I64 || (int64_t)n_sparse * n_samples < (1ll << 31), "offsets do not fit int32");
    const int fields = 1 + n_dense + n_sparse;
    const size_t lds = sizeof(int32_t) * kDecSamples * (size_t)(fields | 1);
    const unsigned blocks = (unsigned)ceil_div(n_samples, kDecSamples);
    hipStream_t s = as_stream(stream);
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(criteo_decode_kernel<int64_t>, dim3(blocks), dim3(kDecThreads), lds, s, d_records, n_samples,
                           n_dense, n_sparse, d_table_rows, d_dense, dense_ld, d_labels, static_cast<int64_t *>(d_indices),
                           static_cast<int64_t *>(d_offsets));
    else
        hipLaunchKernelGGL(criteo_decode_kernel<int32_t>, dim3(blocks), dim3(kDecThreads), lds, s, d_records, n_samples,
                           n_dense, n_sparse, d_table_rows, d_dense, dense_ld, d_labels, static_cast<int32_t *>(d_indices),
                           static_cast<int32_t *>(d_offsets));
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
