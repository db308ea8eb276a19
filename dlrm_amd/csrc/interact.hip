// Pairwise dot feature interaction on the gfx950 matrix cores, forward and
// backward, one sample per wave.
//
// forward : Z = X X^T for X = [F, D] (F <= 32 features padded to 32 rows) with
//           v_mfma_f32_32x32x2_f32; the strict lower triangle of Z plus a copy
//           of feature 0 form the output row.  A and B fragments of X X^T are
//           the same registers (lane l holds X[l&31][k]).
// backward: dX = (S + S^T) X + [d(dense copy) on feature 0], S = pair
//           gradients scattered into a 32x32 LDS tile.
// Memory bound (F*D*4 bytes in, (D + F(F-1)/2)*4 bytes out per sample); the
// sample is staged through LDS with 16-byte coalesced loads.  LDS rows are
// padded by 4 floats so the ds_read_b128 fragment reads are conflict-free.
#include "common.h"

namespace dlrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kIntThreads = 256;
constexpr int kIntWaves = kIntThreads / 64;
constexpr int kFP = 32;  // padded feature count
constexpr int kSLd = 36; // row stride of the 32x32 gradient tile

__device__ inline void stage_sample(const float *__restrict__ xb, int F, int D, int ld, float *__restrict__ sx,
                                    int lane) {
    const int n4 = F * D / 4;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = reinterpret_cast<const float4 *>(xb)[i];
        const int e = i * 4, row = e / D, col = e - row * D;
        *reinterpret_cast<float4 *>(sx + row * ld + col) = v;
    }
}

__global__ __launch_bounds__(kIntThreads) void interact_dot_fwd_kernel(const float *__restrict__ x, int64_t B,
                                                                        int F, int D, float *__restrict__ out,
                                                                        int64_t out_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float *sx = smem + (size_t)wave * kFP * ld;
    // rows F..31 stay zero for the whole kernel
    for (int i = lane; i < (kFP - F) * ld; i += 64) sx[F * ld + i] = 0.f;
    const int n_pairs = F * (F - 1) / 2;
    const int row = lane & 31, h = lane >> 5;

    for (int64_t b = (int64_t)blockIdx.x * kIntWaves + wave; b < B; b += (int64_t)gridDim.x * kIntWaves) {
        stage_sample(x + b * F * D, F, D, ld, sx, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        for (int q = 0; q < D / 8; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(sx + row * ld + q * 8 + h * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, v.w, acc, 0, 0, 0);
        }
        float *o = out + b * out_ld;
        // dense copy (feature 0) and zero padding of the row tail
        for (int d = lane; d < D; d += 64) o[d] = sx[d];
        for (int64_t c = D + n_pairs + lane; c < out_ld; c += 64) o[c] = 0.f;
        // register r of lane l holds Z[i][j], i = (r&3) + 8*(r>>2) + 4*h, j = l&31
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i < F && row < i) o[D + i * (i - 1) / 2 + row] = acc[r];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(kIntThreads) void interact_dot_bwd_kernel(const float *__restrict__ x,
                                                                        const float *__restrict__ dout,
                                                                        int64_t dout_ld, int64_t B, int F, int D,
                                                                        float *__restrict__ dx, int relu_mask0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float *sx = smem + (size_t)wave * (kFP * ld + kFP * kSLd);
    float *ss = sx + kFP * ld;
    for (int i = lane; i < (kFP - F) * ld; i += 64) sx[F * ld + i] = 0.f;
    const int n_pairs = F * (F - 1) / 2;
    const int row = lane & 31, h = lane >> 5;

    for (int64_t b = (int64_t)blockIdx.x * kIntWaves + wave; b < B; b += (int64_t)gridDim.x * kIntWaves) {
        stage_sample(x + b * F * D, F, D, ld, sx, lane);
        const float *go = dout + b * dout_ld;
        for (int i = lane; i < kFP * kSLd; i += 64) ss[i] = 0.f;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int c = lane; c < n_pairs; c += 64) {
            // c = i(i-1)/2 + j, 0 <= j < i
            int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)c)) * 0.5f);
            while (i * (i - 1) / 2 > c) i--;
            while ((i + 1) * i / 2 <= c) i++;
            const int j = c - i * (i - 1) / 2;
            const float g = go[D + c];
            ss[i * kSLd + j] = g;
            ss[j * kSLd + i] = g;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // A fragments of S' = S + S^T: 4 chunks of 8 reduction (feature) indices
        float4 fa[4];
#pragma unroll
        for (int q = 0; q < 4; q++) fa[q] = *reinterpret_cast<const float4 *>(ss + row * kSLd + q * 8 + h * 4);
        float *gx = dx + b * F * D;
        for (int d0 = 0; d0 < D; d0 += 32) {
            const int d = d0 + row;
            const bool d_ok = d < D;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float a[4] = {fa[q].x, fa[q].y, fa[q].z, fa[q].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float bv = d_ok ? sx[(q * 8 + h * 4 + j) * ld + d] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv, acc, 0, 0, 0);
                }
            }
            if (d_ok) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (i < F) {
                        float v = acc[r];
                        if (i == 0) {
                            v += go[d];
                            if (relu_mask0 && !(sx[d] > 0.f)) v = 0.f;
                        }
                        gx[i * D + d] = v;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace dlrm

using namespace dlrm;

static int check_interact(const float *d_x, int64_t B, int32_t F, int32_t D) {
    DLRM_REQUIRE(B >= 0 && F >= 1 && D >= 1, "bad shape B=%lld F=%d D=%d", (long long)B, F, D);
    DLRM_REQUIRE(F <= kFP, "dot interaction supports at most %d features, got %d", kFP, F);
    DLRM_REQUIRE(D % 8 == 0 && D <= 512, "dot interaction needs dim %% 8 == 0 and dim <= 512, got %d", D);
    DLRM_REQUIRE(B == 0 || (d_x != nullptr && reinterpret_cast<uintptr_t>(d_x) % 16 == 0), "x must be 16-byte aligned");
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_forward(const float *d_x, int64_t B, int32_t F, int32_t D, float *d_out,
                                         int64_t out_ld, dlrm_stream_t stream) {
    if (int rc = check_interact(d_x, B, F, D)) return rc;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_out != nullptr && out_ld >= D + F * (F - 1) / 2, "out is NULL or out_ld too small");
    const size_t lds = sizeof(float) * kIntWaves * kFP * (D + 4);
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_fwd_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(interact_dot_fwd_kernel, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), d_x, B, F, D,
                       d_out, out_ld);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_backward(const float *d_x, const float *d_dout, int64_t dout_ld, int64_t B,
                                          int32_t F, int32_t D, float *d_dx, int32_t relu_mask_feature0,
                                          dlrm_stream_t stream) {
    if (int rc = check_interact(d_x, B, F, D)) return rc;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_dout != nullptr && d_dx != nullptr && dout_ld >= D + F * (F - 1) / 2,
                 "dout/dx is NULL or dout_ld too small");
    const size_t lds = sizeof(float) * kIntWaves * (kFP * (D + 4) + kFP * kSLd);
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_bwd_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(interact_dot_bwd_kernel, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), d_x, d_dout,
                       dout_ld, B, F, D, d_dx, relu_mask_feature0);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
