// The per-sample loss arithmetic, shared by the loss kernels and by the kernel that evaluates the last,
// one-column layer of the top MLP, the loss gradient and that layer's backward-data pass in one sweep.
#pragma once
#include "common.h"

namespace dlrm {

// d(loss)/d(pred) of one sample (already divided by the number of samples the mean runs over) and, in
// `term`, the sample's loss term
template <bool BCE>
__device__ inline float loss_grad(float x, float t, float inv_n, int relu_mask, float &term) {
    if constexpr (BCE) {
        // sigmoid(x) - t without cancellation when the prediction saturates towards its label
        // (1/(1+e^-x) rounds to exactly 1 from x = 17 on, yet the gradient there is e^-x)
        const float e = expf(-fabsf(x));
        const float r = e / (1.0f + e);                       // sigmoid(-|x|)
        term = fmaxf(x, 0.f) - x * t + log1pf(e);
        return (x >= 0.f ? (1.0f - t) - r : r - t) * inv_n;
    } else {
        const float d = x - t;
        term = d * d;
        float g = 2.0f * d * inv_n;
        if (relu_mask && !(x > 0.f)) g = 0.f;
        return g;
    }
}

}  // namespace dlrm
