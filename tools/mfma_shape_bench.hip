// What the bf16 matrix pipe sustains on this part in the regime of the float32-by-three-bf16-pieces GEMM
// (gemm_bf16x3_kernel): a wave owns a 64 x 64 float32 block, its operand fragments are re-read from LDS for
// every reduction step (three planes per operand: the pieces of a split float32 value), six cross terms per
// product, two waves per SIMD.  No global loads, no split arithmetic: the loop is LDS reads + MFMAs only, so
// the numbers are an upper bound for a kernel with that inner structure.  Two MFMA shapes at equal work:
//   shape 0  v_mfma_f32_32x32x16_bf16: per 16-deep step 12 fragment reads (16 B per lane), 24 MFMAs of 32 cycles
//   shape 1  v_mfma_f32_16x16x32_bf16: per 32-deep step 24 fragment reads,                96 MFMAs of 16 cycles
//   shape 2  v_mfma_f32_16x16x32_bf16 fed from 16-deep steps: the instruction's 32 reduction slots hold the 16
//            elements of TWO cross terms (lanes 0..31 take pieces (x, y) of term one, lanes 32..63 those of term
//            two), so three instructions cover the six terms: per 16-deep step 8 + 12 = 20 fragment reads
//            ([a0|a1], [a0|a2]; [b0|b1], [b1|b0], [b2|b0]), 48 MFMAs of 16 cycles
// Operands: pieces of N(0,1) values ("random"), or zeros (argument 3) -- the clock the chip holds depends on it.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o tools/bin/mfma_shape_bench
//   tools/bin/mfma_shape_bench [iters per launch] [launches] [zero]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// LDS image: 2 operands x 3 planes x 128 rows x 32 reduction elements of bf16 = 48 KB (one 32-deep step of a
// 128 x 128 workgroup tile, or two 16-deep half-steps).  Fragment addressing is conflict-free for both shapes:
// a lane reads 16 bytes; rows are 64 bytes, the 16-byte chunk index is XORed with (row >> 1) & 3.
constexpr int kPlane = 128 * 32 * 2;

__device__ inline int frag_off(int row, int chunk) { return row * 64 + 16 * (chunk ^ ((row >> 1) & 3)); }

constexpr int kTA[6] = {2, 1, 0, 1, 0, 0};
constexpr int kTB[6] = {0, 1, 2, 0, 1, 0};

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const uint4 *__restrict__ image, float *__restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * kPlane];
    for (int i = threadIdx.x; i < 6 * kPlane / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = image[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned char *pa = smem, *pb = smem + 3 * kPlane;
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int mi = 0; mi < 2; mi++)
            for (int ni = 0; ni < 2; ni++)
                for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;
        for (int it = 0; it < iters; it++) {
            int fresh = 0;
            asm volatile("" : "+v"(fresh));   // the reads depend on it: they stay inside the loop
            pa += fresh; pb += fresh;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int p = 0; p < 3; p++) {
                        a[i][p] = *reinterpret_cast<const bf16x8 *>(pa + p * kPlane + frag_off(wr * 64 + i * 32 + (lane & 31), 2 * half + (lane >> 5)));
                        b[i][p] = *reinterpret_cast<const bf16x8 *>(pb + p * kPlane + frag_off(wc * 64 + i * 32 + (lane & 31), 2 * half + (lane >> 5)));
                    }
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int t = 0; t < 6; t++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][kTA[t]], b[ni][kTB[t]], acc[mi][ni], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        float s = 0.f;
        for (int mi = 0; mi < 2; mi++)
            for (int ni = 0; ni < 2; ni++)
                for (int r = 0; r < 16; r++) s += acc[mi][ni][r];
        out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (SHAPE == 1) {
        f32x4 acc[4][4];
        for (int mi = 0; mi < 4; mi++)
            for (int ni = 0; ni < 4; ni++)
                for (int r = 0; r < 4; r++) acc[mi][ni][r] = 0.f;
        for (int it = 0; it < iters; it++) {
            int fresh = 0;
            asm volatile("" : "+v"(fresh));
            pa += fresh; pb += fresh;
            // lane (l & 15, l >> 4) holds reduction elements 8 (l >> 4) .. + 7 of row l & 15
            bf16x8 b[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int p = 0; p < 3; p++)
                    b[i][p] = *reinterpret_cast<const bf16x8 *>(pb + p * kPlane + frag_off(wc * 64 + i * 16 + (lane & 15), lane >> 4));
#pragma unroll
            for (int mi = 0; mi < 4; mi++) {
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; p++)
                    a[p] = *reinterpret_cast<const bf16x8 *>(pa + p * kPlane + frag_off(wr * 64 + mi * 16 + (lane & 15), lane >> 4));
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
#pragma unroll
                    for (int t = 0; t < 6; t++)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kTA[t]], b[ni][kTB[t]], acc[mi][ni], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        float s = 0.f;
        for (int mi = 0; mi < 4; mi++)
            for (int ni = 0; ni < 4; ni++)
                for (int r = 0; r < 4; r++) s += acc[mi][ni][r];
        out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[4][4];
        for (int mi = 0; mi < 4; mi++)
            for (int ni = 0; ni < 4; ni++)
                for (int r = 0; r < 4; r++) acc[mi][ni][r] = 0.f;
        // lane (l & 15, g = l >> 4): groups 0, 1 read reduction elements 8 (g & 1) .. + 7 of the first piece of a
        // pair, groups 2, 3 the same elements of the second piece
        const int g = lane >> 4, hi = g >> 1;
        constexpr int kPairA[2][2] = {{0, 1}, {0, 2}};              // [a0|a1], [a0|a2]
        constexpr int kPairB[3][2] = {{0, 1}, {1, 0}, {2, 0}};      // [b0|b1], [b1|b0], [b2|b0]
        constexpr int kUseA[3] = {0, 0, 1}, kUseB[3] = {0, 1, 2};   // a0b0+a1b1, a0b1+a1b0, a0b2+a2b0
        for (int it = 0; it < iters; it++) {
            int fresh = 0;
            asm volatile("" : "+v"(fresh));
            pa += fresh; pb += fresh;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                bf16x8 b[4][3];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        b[i][q] = *reinterpret_cast<const bf16x8 *>(pb + (hi ? kPairB[q][1] : kPairB[q][0]) * kPlane +
                                                                     frag_off(wc * 64 + i * 16 + (lane & 15), 2 * half + (g & 1)));
#pragma unroll
                for (int mi = 0; mi < 4; mi++) {
                    bf16x8 a[2];
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        a[q] = *reinterpret_cast<const bf16x8 *>(pa + (hi ? kPairA[q][1] : kPairA[q][0]) * kPlane +
                                                                  frag_off(wr * 64 + mi * 16 + (lane & 15), 2 * half + (g & 1)));
#pragma unroll
                    for (int ni = 0; ni < 4; ni++)
#pragma unroll
                        for (int t = 0; t < 3; t++)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kUseA[t]], b[ni][kUseB[t]], acc[mi][ni], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        float s = 0.f;
        for (int mi = 0; mi < 4; mi++)
            for (int ni = 0; ni < 4; ni++)
                for (int r = 0; r < 4; r++) s += acc[mi][ni][r];
        out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    }
}

static uint16_t bf16_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float f32_of(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int launches = argc > 2 ? atoi(argv[2]) : 40;
    const bool zero = argc > 3 && atoi(argv[3]) != 0;
    // pieces of N(0,1) values, as the GEMM's split writes them: plane p of an operand holds piece p
    std::vector<uint16_t> img(6 * kPlane / 2, 0);
    if (!zero) {
        uint64_t s = 0x1234567ull;
        auto uni = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) + 1) / 16777217.0f; };
        for (int op = 0; op < 2; op++)
            for (int e = 0; e < 128 * 32; e++) {
                const float x = sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni());
                const uint16_t h = bf16_of(x);
                const float r1 = x - f32_of(h);
                const uint16_t m = bf16_of(r1);
                const uint16_t l = bf16_of(r1 - f32_of(m));
                img[(op * 3 + 0) * (kPlane / 2) + e] = h;
                img[(op * 3 + 1) * (kPlane / 2) + e] = m;
                img[(op * 3 + 2) * (kPlane / 2) + e] = l;
            }
    }
    uint4 *d_img;
    float *d_out;
    const int grid = 512;   // two workgroups per compute unit
    CHECK(hipMalloc(&d_img, 6 * kPlane));
    CHECK(hipMalloc(&d_out, sizeof(float) * grid * 256));
    CHECK(hipMemcpy(d_img, img.data(), 6 * kPlane, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // flops per iteration and wave: 64 x 64 outputs x 32 reduction elements x 6 terms x 2
    const double flops = (double)grid * 4 * iters * 64.0 * 64.0 * 32.0 * 6.0 * 2.0;
    for (int round = 0; round < 2; round++)
        for (int shape = 0; shape < 3; shape++) {
            auto launch = [&]() {
                if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(256), 0, 0, d_img, d_out, iters);
                else if (shape == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, d_img, d_out, iters);
                else hipLaunchKernelGGL(mfma_loop<2>, dim3(grid), dim3(256), 0, 0, d_img, d_out, iters);
            };
            for (int i = 0; i < 5; i++) launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < launches; i++) launch();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double tf = flops * launches / (ms * 1e-3) / 1e12;
            printf("{\"shape\": \"%s\", \"operands\": \"%s\", \"round\": %d, \"ms_per_launch\": %.4f, \"bf16_mfma_TFLOPs\": %.1f, "
                   "\"f32_equivalent_TFLOPs\": %.1f, \"frac_of_2500\": %.3f}\n",
                   shape == 0 ? "32x32x16" : shape == 1 ? "16x16x32" : "16x16x32 two terms per instruction", zero ? "zeros" : "pieces of N(0,1)", round, ms / launches, tf, tf / 6.0,
                   tf / 2500.0);
        }
    return 0;
}
