// Probe: do packed-float32 VALU results survive when waves of a matrix-core kernel share the SIMDs?
//
// The victim kernel repeats, per lane, the instruction pattern of a divergent accumulate: registers cleared,
// then -- for the lane groups whose turn it is -- v_pk_mul_f32 / v_pk_fma_f32 of freshly loaded values, then the
// end of the divergent region.  Every result is checked against scalar arithmetic on the same inputs; mismatches
// are counted per half (low / high).  It runs alone, then beside an MFMA loop on a second stream.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pk_coexec_probe tools/pk_coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_spin(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) {   // operands with random signs and mantissas: constant inputs draw far less power
        const unsigned h = (threadIdx.x * 8u + i + blockIdx.x * 2048u) * 2654435761u;
        a[i] = (short)(((h >> 8) & 0x807f) | 0x3f80);
        b[i] = (short)(((h >> 17) & 0x807f) | 0x3f00);
    }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 12345.f) out[0] = s;
}

// VALU / LDS instruction loops that the GEMM kernels of this library are made of (besides the MFMAs above)
template <int KIND>
__global__ __launch_bounds__(256) void valu_spin(float *out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    float a = 1.0f + threadIdx.x, b = 0.5f + blockIdx.x;
    f32x2 p = {a, b}, q = {b, a};
    unsigned r = 0;
    f32x2 t = {0.f, 0.f};
    const unsigned addr = (threadIdx.x * 8u) & 0x3ff8u;
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == 1) {
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %0, %2, %1\n\tv_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %0, %2, %1"
                         : "=&v"(r) : "v"(a), "v"(b));
        } else if constexpr (KIND == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %2, %1\n\tv_pk_add_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %2, %1"
                         : "=&v"(t) : "v"(p), "v"(q));
        } else if constexpr (KIND == 3) {
            asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(addr) : "memory");
        } else if constexpr (KIND == 4) {
            asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=&v"(r) : "v"((unsigned)i), "v"(addr), "v"(r));
        }
    }
    if (r == 0x12345u && t[0] == 3.25f) out[0] = a + lds[threadIdx.x];
}

__device__ inline float smul(float a, float b) {   // a product the compiler cannot pack
    float r;
    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// mode 0: packed instructions written in assembly (v_pk_mul_f32 + v_pk_fma_f32, broadcast scale like the compiler emits)
// mode 1: the same arithmetic with scalar instructions
template <int MODE>
__global__ __launch_bounds__(256) void victim(const f32x4 *__restrict__ g, int n_vec, int iters, unsigned *bad) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = threadIdx.x >> 2;
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned h = (unsigned)(grp * 2654435761u + it * 40503u + blockIdx.x * 97u);
        const bool mine = ((h >> 7) & 3) != 0;                 // a group-uniform, wave-divergent condition
        const f32x4 x = g[(unsigned)(tid * 7 + it * 131) % (unsigned)n_vec];
        const float sc = 1.0f + (float)((h >> 3) & 7);
        f32x2 r0 = {0.f, 0.f}, r1 = {0.f, 0.f};
        if (mine) {
            if constexpr (MODE == 0) {
                const f32x2 s2 = {sc, sc};
                const f32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
                asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
                             "v_pk_fma_f32 %1, %2, %4, 0 op_sel_hi:[0,1,0]"
                             : "=&v"(r0), "=&v"(r1)
                             : "v"(s2), "v"(lo), "v"(hi));
            } else {
                r0[0] = smul(x[0], sc); r0[1] = smul(x[1], sc); r1[0] = smul(x[2], sc); r1[1] = smul(x[3], sc);
            }
        }
        const float e0 = mine ? smul(x[0], sc) : 0.f, e1 = mine ? smul(x[1], sc) : 0.f;
        const float e2 = mine ? smul(x[2], sc) : 0.f, e3 = mine ? smul(x[3], sc) : 0.f;
        bad_lo += (r0[0] != e0) + (r1[0] != e2);
        bad_hi += (r0[1] != e1) + (r1[1] != e3);
    }
    if (bad_lo) atomicAdd(&bad[0], bad_lo);
    if (bad_hi) atomicAdd(&bad[1], bad_hi);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int n_vec = 1 << 20;
    std::vector<float> h((size_t)n_vec * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3f * (float)((int)(i * 2654435761u >> 8) % 2001 - 1000) + 0.0005f;
    f32x4 *g;
    unsigned *bad;
    float *sink;
    CK(hipMalloc(&g, h.size() * 4));
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&sink, 4));
    CK(hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *names[] = {"alone", "beside MFMA waves", "beside v_cvt_pk_bf16_f32 waves", "beside v_pk_add_f32 waves",
                           "beside ds_read_b64_tr_b16 waves", "beside v_bitop3_b32 waves"};
    for (int mode = 0; mode < 2; mode++)
        for (int with_mfma = 0; with_mfma < (mode == 0 ? 6 : 2); with_mfma++) {
            unsigned tot[2] = {0, 0};
            for (int r = 0; r < reps; r++) {
                CK(hipMemsetAsync(bad, 0, 8, sa));
                CK(hipStreamSynchronize(sa));
                if (with_mfma == 1) hipLaunchKernelGGL(mfma_spin, dim3(512), dim3(256), 0, sb, sink, 20000);
                if (with_mfma == 2) hipLaunchKernelGGL(valu_spin<1>, dim3(1024), dim3(256), 0, sb, sink, 40000);
                if (with_mfma == 3) hipLaunchKernelGGL(valu_spin<2>, dim3(1024), dim3(256), 0, sb, sink, 40000);
                if (with_mfma == 4) hipLaunchKernelGGL(valu_spin<3>, dim3(1024), dim3(256), 0, sb, sink, 40000);
                if (with_mfma == 5) hipLaunchKernelGGL(valu_spin<4>, dim3(1024), dim3(256), 0, sb, sink, 100000);
                if (mode == 0) hipLaunchKernelGGL(victim<0>, dim3(2048), dim3(256), 0, sa, g, n_vec, 200, bad);
                else hipLaunchKernelGGL(victim<1>, dim3(2048), dim3(256), 0, sa, g, n_vec, 200, bad);
                CK(hipDeviceSynchronize());
                unsigned hb[2];
                CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
                tot[0] += hb[0]; tot[1] += hb[1];
            }
            printf("%s, %s: wrong low halves %u, wrong high halves %u (of %.3g results per half)\n",
                   mode == 0 ? "packed (v_pk_mul_f32 / v_pk_fma_f32)" : "scalar float32", names[with_mfma],
                   tot[0], tot[1], 2.0 * reps * 2048.0 * 256.0 * 200.0);
            fflush(stdout);
        }
    return 0;
}
