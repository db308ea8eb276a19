// What the path global memory -> LDS sustains on this part when every compute unit streams 1 KB requests the way
// gemm_f16p_kernel does (buffer_load_dwordx4 ... lds, 64 lanes x 16 B side by side, eight requests in flight per wave,
// two 256-thread workgroups per compute unit), by the size of the region the requests walk through:
//   shared   every workgroup walks the SAME region (like the weight planes: L2 resident when it fits)
//   private  every workgroup walks its own region of that size
// and for comparison the same requests returning into registers.  Prints GB/s over all compute units.
//
//   hipcc -O3 --offload-arch=gfx950 tools/l2_bw_bench.hip -o tools/bin/l2_bw_bench && tools/bin/l2_bw_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// region_bytes: power of two, >= 64 KB.  Each wave issues `iters` groups of eight 1 KB requests; group g of wave w of
// workgroup b starts at ((b * 4 + w) * 37 + g) * 8 KB modulo the region (distinct workgroups are at distinct places).
template <bool TO_LDS, int HALF>
__global__ __launch_bounds__(256, 2) void stream_kernel(const unsigned char *base, size_t region_bytes, size_t wg_stride, int iters,
                                                        unsigned *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned char *mine = base + (size_t)blockIdx.x * wg_stride;
    const size_t mask = region_bytes - 1;
    u32x4 acc = {0, 0, 0, 0};
    // HALF: a request covers sixteen 64-byte pieces 128 bytes apart (half lines) instead of 1 KB side by side
    const unsigned voff = HALF ? (unsigned)((lane >> 2) * 128 + (lane & 3) * 16) : (unsigned)(lane * 16);
    for (int g = 0; g < iters; g++) {
        const size_t start = ((((size_t)blockIdx.x * 4 + wave) * 37 + g) * (HALF ? 16384 : 8192)) & mask;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(mine + start), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if constexpr (TO_LDS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + wave * 16384 + (g & 1) * 8192 + 1024 * j), 16,
                                                         (int)voff + (HALF ? 2048 : 1024) * j, 0, 0, 0);
            else acc += (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff + (HALF ? 2048 : 1024) * j, 0, 0);
        }
        if constexpr (TO_LDS) __builtin_amdgcn_s_waitcnt(0x0F78);     // the group before this one has landed
    }
    if constexpr (TO_LDS) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        acc = *reinterpret_cast<u32x4 *>(smem + threadIdx.x * 16);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[0] = 1;
}

template <bool TO_LDS, int HALF>
static double run(const unsigned char *buf, size_t region, bool shared, int wgs, int iters) {
    unsigned *sink;
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t stride = shared ? 0 : region;
    hipLaunchKernelGGL((stream_kernel<TO_LDS, HALF>), dim3(wgs), dim3(256), 0, 0, buf, region, stride, iters, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((stream_kernel<TO_LDS, HALF>), dim3(wgs), dim3(256), 0, 0, buf, region, stride, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(sink));
    return 5.0 * wgs * 4.0 * iters * 8192.0 / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv) {
    const int wgs = 512, iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t total = (size_t)wgs * (1 << 20);      // private regions of up to 1 MB each: 512 MB
    unsigned char *buf;
    CHECK(hipMalloc(&buf, total + (1 << 20)));
    CHECK(hipMemset(buf, 1, total + (1 << 20)));
    printf("{\"wgs\": %d, \"requests_per_wave\": %d, \"rows\": [\n", wgs, iters * 8);
    const size_t shared_sizes[] = {1u << 20, 4u << 20, 32u << 20, 256u << 20};
    for (size_t r : shared_sizes) {
        printf("  {\"region\": \"shared %zu MB\", \"to_lds_GBps\": %.0f, \"to_lds_half_lines_GBps\": %.0f, \"to_registers_GBps\": %.0f},\n",
               r >> 20, run<true, 0>(buf, r, true, wgs, iters), run<true, 1>(buf, r, true, wgs, iters), run<false, 0>(buf, r, true, wgs, iters));
        fflush(stdout);
    }
    const size_t private_sizes[] = {64u << 10, 1u << 20};
    for (size_t r : private_sizes) {
        printf("  {\"region\": \"private %zu KB each\", \"to_lds_GBps\": %.0f, \"to_lds_half_lines_GBps\": %.0f, \"to_registers_GBps\": %.0f}%s\n",
               r >> 10, run<true, 0>(buf, r, false, wgs, iters), run<true, 1>(buf, r, false, wgs, iters), run<false, 0>(buf, r, false, wgs, iters),
               r == (1u << 20) ? "" : ",");
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}
