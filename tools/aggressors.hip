// Aggressor kernels for tools/overlap_probe.py: loops of single instruction kinds that the bf16x3 GEMM kernels are made
// of, launched on a caller's stream beside the real sparse row update.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libaggressors.so tools/aggressors.hip
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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


// Second victim: what the compiled sparse update does around its packed instructions -- two loads into the SAME
// destination registers from the two sides of a divergent branch (disjoint lane groups), both in flight, one
// s_waitcnt, then the packed multiply reads the pair.
__global__ __launch_bounds__(256) void victim_waw(const f32x4 *__restrict__ g, int n_vec, int iters, unsigned *bad) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = threadIdx.x >> 2;
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned h = (unsigned)(grp * 2654435761u + it * 40503u + blockIdx.x * 97u);
        const bool side = ((h >> 9) & 1) != 0;                  // group-uniform, wave-divergent
        const unsigned i1 = (unsigned)(tid * 7 + it * 131) % (unsigned)n_vec, i2 = (unsigned)(tid * 13 + it * 257 + 5) % (unsigned)n_vec;
        const f32x4 *p1 = g + i1, *p2 = g + i2;
        const float sc = 1.0f + (float)((h >> 3) & 7);
        const f32x2 s2 = {sc, sc};
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
        if (side) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(x) : "v"(p1) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "+v"(x), "+v"(w) : "v"(p2), "v"(p1) : "memory");
        }
        f32x2 r0, r1;
        asm volatile("s_waitcnt vmcnt(0)\n\t"
                     "v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %1, %2, %4 op_sel_hi:[0,1]"
                     : "=&v"(r0), "=&v"(r1)
                     : "v"(s2), "v"(__builtin_shufflevector(x, x, 0, 1)), "v"(__builtin_shufflevector(x, x, 2, 3)), "v"(w));
        const f32x4 e = side ? *p1 : *p2;
        bad_lo += (r0[0] != smul(e[0], sc)) + (r1[0] != smul(e[2], sc));
        bad_hi += (r0[1] != smul(e[1], sc)) + (r1[1] != smul(e[3], sc));
    }
    if (bad_lo) atomicAdd(&bad[0], bad_lo);
    if (bad_hi) atomicAdd(&bad[1], bad_hi);
}

// The bare victim pattern (tools/pk_coexec_probe.hip) on a caller's stream: mode 0 packed, 1 scalar; bad[0..1] count wrong
// low / high halves.
extern "C" int launch_victim(int mode, void *stream, const float *g, int n_vec, int iters, unsigned *bad) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 2) hipLaunchKernelGGL(victim_waw, dim3(2048), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(g), n_vec, iters, bad);
    else if (mode == 0) hipLaunchKernelGGL(victim<0>, dim3(2048), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(g), n_vec, iters, bad);
    else hipLaunchKernelGGL(victim<1>, dim3(2048), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(g), n_vec, iters, bad);
    return hipGetLastError() == hipSuccess ? 0 : 2;
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
