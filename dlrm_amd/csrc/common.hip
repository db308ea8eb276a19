// Error reporting, version query and table initialisation.
#include "common.h"

#include <stdarg.h>

namespace dlrm {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

template <typename T>
__global__ void table_init_uniform_kernel(T *dst, int64_t n, uint64_t seed, int64_t first) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the seed is hashed first: streams of nearby seeds (one per table) are unrelated, not shifted copies
    const uint64_t base = splitmix64(seed);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = (float)(splitmix64(base + (uint64_t)(first + i)) >> 40) * (1.0f / 16777216.0f);
        if constexpr (sizeof(T) == 4)
            dst[i] = v;
        else
            dst[i] = (T)f32_to_bf16_bits(v);
    }
}

// one thread per 16 bytes when both sides allow it, else per float
template <typename V>
__global__ void copy_2d_kernel(V *__restrict__ dst, int64_t dst_ld, const V *__restrict__ src, int64_t src_ld,
                               int64_t rows, int64_t cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

// several strided 2-D copies in one launch: workgroups are dealt to the items in proportion to their size
struct CopyManyArgs {
    dlrm_copy_item item[DLRM_COPY_MAX_ITEMS];
    int first_block[DLRM_COPY_MAX_ITEMS + 1];
    int n;
};

template <typename V, int W>
__global__ __launch_bounds__(256) void copy_2d_many_kernel(CopyManyArgs a) {
    int it = 0;
    while (it + 1 < a.n && (int)blockIdx.x >= a.first_block[it + 1]) it++;
    const dlrm_copy_item m = a.item[it];
    const int64_t cols = m.cols / W, n = m.rows * cols;
    const int64_t dst_ld = m.dst_ld / W, src_ld = m.src_ld / W;
    const int nb = a.first_block[it + 1] - a.first_block[it];
    V *__restrict__ dst = reinterpret_cast<V *>(m.d_dst);
    const V *__restrict__ src = reinterpret_cast<const V *>(m.d_src);
    for (int64_t i = (int64_t)(blockIdx.x - a.first_block[it]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_copy_2d_many(const dlrm_copy_item *items, int32_t n_items, dlrm_stream_t stream) {
    DLRM_REQUIRE(n_items >= 0 && (items != nullptr || n_items == 0), "bad arguments");
    for (int first = 0; first < n_items; first += DLRM_COPY_MAX_ITEMS) {
        CopyManyArgs a{};
        bool v4 = true;
        int64_t total = 0;
        for (int i = first; i < n_items && i < first + DLRM_COPY_MAX_ITEMS; i++) {
            const dlrm_copy_item &m = items[i];
            DLRM_REQUIRE(m.rows >= 0 && m.cols >= 0 && m.dst_ld >= m.cols && m.src_ld >= m.cols, "item %d: bad 2-D copy shape", i);
            if (m.rows == 0 || m.cols == 0) continue;
            DLRM_REQUIRE(m.d_dst != nullptr && m.d_src != nullptr, "item %d: NULL operand", i);
            v4 = v4 && m.cols % 4 == 0 && m.dst_ld % 4 == 0 && m.src_ld % 4 == 0 &&
                 reinterpret_cast<uintptr_t>(m.d_dst) % 16 == 0 && reinterpret_cast<uintptr_t>(m.d_src) % 16 == 0;
            a.item[a.n++] = m;
            total += m.rows * m.cols;
        }
        if (a.n == 0) continue;
        // about 4096 workgroups in all, at least one per item
        int blocks = 0;
        for (int i = 0; i < a.n; i++) {
            const int64_t elems = a.item[i].rows * a.item[i].cols / (v4 ? 4 : 1);
            int64_t want = ceil_div(elems, 256);
            const int64_t share = ceil_div(4096 * (a.item[i].rows * a.item[i].cols), total);
            if (want > share) want = share;
            if (want < 1) want = 1;
            a.first_block[i] = blocks;
            blocks += (int)want;
        }
        a.first_block[a.n] = blocks;
        if (v4) hipLaunchKernelGGL((copy_2d_many_kernel<float4, 4>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
        else hipLaunchKernelGGL((copy_2d_many_kernel<float, 1>), dim3(blocks), dim3(256), 0, as_stream(stream), a);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

extern "C" const char *dlrm_last_error(void) { return err_buf(); }
extern "C" int dlrm_version(void) { return DLRM_ABI_VERSION; }
extern "C" const char *dlrm_target_arch(void) { return "gfx950"; }

extern "C" int dlrm_table_init_uniform(void *dst, int row_dtype, int64_t n, uint64_t seed, int64_t first,
                                       dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 0, "n < 0");
    DLRM_REQUIRE(row_dtype == DLRM_F32 || row_dtype == DLRM_BF16, "row_dtype %d", row_dtype);
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(dst != nullptr, "dst is NULL");
    const int threads = 256;
    const int64_t want = ceil_div(n, threads);
    const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
    if (row_dtype == DLRM_F32)
        hipLaunchKernelGGL(table_init_uniform_kernel<float>, dim3(blocks), dim3(threads), 0, as_stream(stream),
                           static_cast<float *>(dst), n, seed, first);
    else
        hipLaunchKernelGGL(table_init_uniform_kernel<uint16_t>, dim3(blocks), dim3(threads), 0,
                           as_stream(stream), static_cast<uint16_t *>(dst), n, seed, first);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_copy_2d(float *d_dst, int64_t dst_ld, const float *d_src, int64_t src_ld, int64_t rows,
                            int64_t cols, dlrm_stream_t stream) {
    DLRM_REQUIRE(rows >= 0 && cols >= 0 && dst_ld >= cols && src_ld >= cols, "bad 2-D copy shape");
    if (rows == 0 || cols == 0) return DLRM_OK;
    DLRM_REQUIRE(d_dst && d_src, "NULL operand");
    const bool v4 = cols % 4 == 0 && dst_ld % 4 == 0 && src_ld % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(d_dst) % 16 == 0 && reinterpret_cast<uintptr_t>(d_src) % 16 == 0;
    const int64_t n = v4 ? rows * cols / 4 : rows * cols;
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    if (v4)
        hipLaunchKernelGGL(copy_2d_kernel<float4>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<float4 *>(d_dst), dst_ld / 4, reinterpret_cast<const float4 *>(d_src),
                           src_ld / 4, rows, cols / 4);
    else
        hipLaunchKernelGGL(copy_2d_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), d_dst, dst_ld, d_src,
                           src_ld, rows, cols);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
