// Loss forward + backward in one pass (mean reduction), single workgroup:
// the prediction vector is one float per sample, so even the MLPerf batch
// (65536) is 256 KiB -- far below anything worth spreading over the chip, and
// a single workgroup gives a fixed summation order.
#include "common.h"

namespace dlrm {

constexpr int kLossThreads = 1024;

__device__ inline float block_sum(float v, float *s) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < kLossThreads / 64; w++) t += s[w];
    return t;  // valid on thread 0
}

template <bool BCE>
__global__ __launch_bounds__(kLossThreads) void loss_kernel(const float *__restrict__ pred,
                                                             const float *__restrict__ target, int64_t n,
                                                             int64_t mean_over, float *__restrict__ loss, float *__restrict__ dpred,
                                                             int relu_mask) {
    __shared__ float s[kLossThreads / 64];
    const float inv_n = 1.0f / (float)mean_over;
    float a = 0.f;
    // kU elements per thread at a time: their loads are independent and in flight together (a plain
    // strided loop pays one memory round trip per element and per thread)
    constexpr int kU = 8;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)kLossThreads * kU) {
        float x[kU], t[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int64_t i = i0 + (int64_t)u * kLossThreads;
            const int64_t ic = i < n ? i : n - 1;
            x[u] = pred[ic];
            t[u] = target[ic];
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int64_t i = i0 + (int64_t)u * kLossThreads;
            if (i >= n) break;
            float g;
            if constexpr (BCE) {
                // sigmoid(x) - t without cancellation when the prediction saturates towards its label
                // (1/(1+e^-x) rounds to exactly 1 from x = 17 on, yet the gradient there is e^-x)
                const float e = expf(-fabsf(x[u]));
                const float r = e / (1.0f + e);                       // sigmoid(-|x|)
                a += fmaxf(x[u], 0.f) - x[u] * t[u] + log1pf(e);
                g = (x[u] >= 0.f ? (1.0f - t[u]) - r : r - t[u]) * inv_n;
            } else {
                const float d = x[u] - t[u];
                a += d * d;
                g = 2.0f * d * inv_n;
                if (relu_mask && !(x[u] > 0.f)) g = 0.f;
            }
            if (dpred) dpred[i] = g;
        }
    }
    const float tot = block_sum(a, s);
    if (threadIdx.x == 0 && loss) *loss = tot * inv_n;
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_loss_mse(const float *d_pred, const float *d_target, int64_t n, int64_t mean_over, float *d_loss,
                             float *d_dpred, int32_t relu_mask_pred, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 1 && mean_over >= 1 && d_pred && d_target, "bad arguments");
    hipLaunchKernelGGL(loss_kernel<false>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), d_pred, d_target, n,
                       mean_over, d_loss, d_dpred, relu_mask_pred);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_loss_bce_logits(const float *d_logit, const float *d_target, int64_t n, int64_t mean_over,
                                    float *d_loss, float *d_dlogit, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 1 && mean_over >= 1 && d_logit && d_target, "bad arguments");
    hipLaunchKernelGGL(loss_kernel<true>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), d_logit, d_target, n,
                       mean_over, d_loss, d_dlogit, 0);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
