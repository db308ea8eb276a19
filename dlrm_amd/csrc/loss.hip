// Loss value and gradient (mean reduction).  The gradient is what the backward pass waits for: one
// thread per four samples over as many workgroups as that takes (a few microseconds at the MLPerf batch).
// The loss VALUE is a report, nothing downstream reads it: a single workgroup sums it in a fixed order
// (65536 samples are 64 per thread, about 40 microseconds of exp / log arithmetic on one compute unit),
// so a host that cares launches it on a side stream (d_dpred = NULL) beside the backward pass and the
// gradient alone (d_loss = NULL) on the critical path.  With both pointers given the two kernels run
// one after the other on the caller's stream.
#include "common.h"
#include "loss.h"

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

// gradient only: thread i owns samples 4i .. 4i+3
template <bool BCE>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                         int64_t n, int64_t mean_over, float *__restrict__ dpred, int relu_mask) {
    const float inv_n = 1.0f / (float)mean_over;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    float x[4], t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int64_t ic = i0 + u < n ? i0 + u : n - 1;
        x[u] = pred[ic];
        t[u] = target[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float term;
        const float g = loss_grad<BCE>(x[u], t[u], inv_n, relu_mask, term);
        if (i0 + u < n) dpred[i0 + u] = g;
    }
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
            float term;
            const float g = loss_grad<BCE>(x[u], t[u], inv_n, relu_mask, term);
            a += term;
            if (dpred) dpred[i] = g;
        }
    }
    const float tot = block_sum(a, s);
    if (threadIdx.x == 0 && loss) *loss = tot * inv_n;
}

}  // namespace dlrm

using namespace dlrm;

template <bool BCE>
static int launch_loss(const float *pred, const float *target, int64_t n, int64_t mean_over, float *loss, float *dpred,
                       int relu_mask, hipStream_t s) {
    if (dpred) {
        hipLaunchKernelGGL(loss_grad_kernel<BCE>, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, s, pred, target, n, mean_over,
                           dpred, relu_mask);
        DLRM_LAUNCH_CHECK();
    }
    if (loss) {
        hipLaunchKernelGGL(loss_kernel<BCE>, dim3(1), dim3(kLossThreads), 0, s, pred, target, n, mean_over, loss,
                           static_cast<float *>(nullptr), relu_mask);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

extern "C" int dlrm_loss_mse(const float *d_pred, const float *d_target, int64_t n, int64_t mean_over, float *d_loss,
                             float *d_dpred, int32_t relu_mask_pred, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 1 && mean_over >= 1 && d_pred && d_target, "bad arguments");
    return launch_loss<false>(d_pred, d_target, n, mean_over, d_loss, d_dpred, relu_mask_pred, as_stream(stream));
}

extern "C" int dlrm_loss_bce_logits(const float *d_logit, const float *d_target, int64_t n, int64_t mean_over,
                                    float *d_loss, float *d_dlogit, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 1 && mean_over >= 1 && d_logit && d_target, "bad arguments");
    return launch_loss<true>(d_logit, d_target, n, mean_over, d_loss, d_dlogit, 0, as_stream(stream));
}
