// Aggressor kernels for tools/overlap_probe.py: loops of single instruction kinds that the bf16x3 GEMM kernels are made
// of, launched on a caller's stream beside the real sparse row update.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libaggressors.so tools/aggressors.hip
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

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


// kind: 0 MFMA (bf16 32x32x16), 1 v_cvt_pk_bf16_f32, 2 v_pk_add_f32, 3 ds_read_b64_tr_b16, 4 v_bitop3_b32
extern "C" int launch_aggressor(int kind, void *stream, float *sink, int blocks, int iters) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (kind) {
        case 0: hipLaunchKernelGGL(mfma_spin, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case 1: hipLaunchKernelGGL(valu_spin<1>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case 2: hipLaunchKernelGGL(valu_spin<2>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case 3: hipLaunchKernelGGL(valu_spin<3>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        case 4: hipLaunchKernelGGL(valu_spin<4>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
