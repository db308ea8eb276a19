// Kernels for the two layer shapes of a DLRM that are no matrix product (see skinny.h): plain float32
// fma chains, one pass over the large operand.
#include "skinny.h"
#include "loss.h"

#include <stdlib.h>

namespace dlrm {

constexpr int kSkThreads = 256;
constexpr int kSkRows = 32;   // rows of x staged in LDS per trip

// rows [m0, m0 + kSkRows) of x, zero beyond K and beyond m_end, as [kSkRows][kSkinnyK] in LDS
__device__ inline void stage_rows(float (*sx)[kSkinnyK], const float *__restrict__ x, int64_t x_ld, int m0, int m_end, int K) {
    for (int i = threadIdx.x; i < kSkRows * kSkinnyK; i += kSkThreads) {
        const int r = i / kSkinnyK, k = i % kSkinnyK;
        const int m = m0 + r;
        sx[r][k] = (m < m_end && k < K) ? x[(int64_t)m * x_ld + k] : 0.f;
    }
}

// A thread owns one output column (its weights sit in registers); a workgroup walks a range of rows whose x
// values every lane reads from the same LDS address (broadcast).  Output rows leave as coalesced 4-byte stores.
__global__ __launch_bounds__(kSkThreads) void skinny_k_fwd_kernel(const float *__restrict__ x, int64_t x_ld,
                                                                   const float *__restrict__ w, int64_t w_ld,
                                                                   const float *__restrict__ bias, float *__restrict__ y,
                                                                   int64_t y_ld, int M, int N, int K, int relu,
                                                                   int rows_per_block, float *__restrict__ y_max) {
    __shared__ float sx[kSkRows][kSkinnyK];
    float vmax = 0.f;
    const int n = blockIdx.x * kSkThreads + threadIdx.x;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float wr[kSkinnyK];
#pragma unroll
    for (int k = 0; k < kSkinnyK; k++) wr[k] = (n < N && k < K) ? w[(int64_t)n * w_ld + k] : 0.f;
    const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += kSkRows) {
        __syncthreads();
        stage_rows(sx, x, x_ld, m0, m_end, K);
        __syncthreads();
        if (n >= N) continue;
        const int rows = min(kSkRows, m_end - m0);
#pragma unroll 4
        for (int r = 0; r < rows; r++) {
            float a = b;
#pragma unroll
            for (int k = 0; k < kSkinnyK; k++) a = fmaf(sx[r][k], wr[k], a);
            a = relu ? fmaxf(a, 0.f) : a;
            y[(int64_t)(m0 + r) * y_ld + n] = a;
            vmax = fmaxf(vmax, fabsf(a));
        }
    }
    if (y_max != nullptr) fold_absmax_block(vmax, y_max);
}

// the same walk with the roles swapped: dz[m][n] streams in (coalesced), x[m][:] comes from LDS, the thread keeps
// dW[n][0..K) and db[n] of its row range
__global__ __launch_bounds__(kSkThreads) void skinny_k_bwd_weight_kernel(const float *__restrict__ dz, int64_t dz_ld,
                                                                          const float *__restrict__ x, int64_t x_ld,
                                                                          float *__restrict__ slabs,
                                                                          float *__restrict__ db_slabs, int M, int N, int K,
                                                                          int rows_per_block) {
    __shared__ float sx[kSkRows][kSkinnyK];
    const int n = blockIdx.x * kSkThreads + threadIdx.x;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float acc[kSkinnyK];
#pragma unroll
    for (int k = 0; k < kSkinnyK; k++) acc[k] = 0.f;
    float accb = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += kSkRows) {
        __syncthreads();
        stage_rows(sx, x, x_ld, m0, m_end, K);
        __syncthreads();
        if (n >= N) continue;
        const int rows = min(kSkRows, m_end - m0);
        if (rows == kSkRows) {
#pragma unroll
            for (int r0 = 0; r0 < kSkRows; r0 += 8) {
                float g[8];
#pragma unroll
                for (int u = 0; u < 8; u++) g[u] = dz[(int64_t)(m0 + r0 + u) * dz_ld + n];   // eight loads in flight
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    accb += g[u];
#pragma unroll
                    for (int k = 0; k < kSkinnyK; k++) acc[k] = fmaf(g[u], sx[r0 + u][k], acc[k]);
                }
            }
        } else {
            for (int r = 0; r < rows; r++) {
                const float g = dz[(int64_t)(m0 + r) * dz_ld + n];
                accb += g;
#pragma unroll
                for (int k = 0; k < kSkinnyK; k++) acc[k] = fmaf(g, sx[r][k], acc[k]);
            }
        }
    }
    if (n < N) {
        float *o = slabs + ((int64_t)blockIdx.y * N + n) * K;
#pragma unroll
        for (int k = 0; k < kSkinnyK; k++)
            if (k < K) o[k] = acc[k];
        if (db_slabs) db_slabs[(int64_t)blockIdx.y * N + n] = accb;
    }
}


// ---- one output column -----------------------------------------------------

// 16 lanes per row, 16 bytes each per trip
__global__ __launch_bounds__(kSkThreads) void gemv_fwd_kernel(const float *__restrict__ x, int64_t x_ld,
                                                               const float *__restrict__ w, const float *__restrict__ bias,
                                                               float *__restrict__ y, int64_t y_ld, int M, int K, int relu,
                                                               float *__restrict__ y_max) {
    const int lig = threadIdx.x & 15;
    const float b = bias ? bias[0] : 0.f;
    float vmax = 0.f;
    for (int m = blockIdx.x * (kSkThreads / 16) + (threadIdx.x >> 4); m < M; m += gridDim.x * (kSkThreads / 16)) {
        const float *xr = x + (int64_t)m * x_ld;
        float a = 0.f;
        for (int k = lig * 4; k < K; k += 64) {
            const float4 xv = *reinterpret_cast<const float4 *>(xr + k);
            const float4 wv = *reinterpret_cast<const float4 *>(w + k);
            a = fmaf(xv.x, wv.x, a); a = fmaf(xv.y, wv.y, a); a = fmaf(xv.z, wv.z, a); a = fmaf(xv.w, wv.w, a);
        }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
        if (lig == 0) {
            a += b;
            a = relu ? fmaxf(a, 0.f) : a;
            y[(int64_t)m * y_ld] = a;
            vmax = fmaxf(vmax, fabsf(a));
        }
    }
    if (y_max != nullptr) fold_absmax_block(vmax, y_max);
}

__global__ __launch_bounds__(kSkThreads) void gemv_bwd_data_kernel(const float *__restrict__ dz, int64_t dz_ld,
                                                                    const float *__restrict__ w, float *__restrict__ dx,
                                                                    int64_t dx_ld, const float *__restrict__ mask,
                                                                    int64_t mask_ld, int M, int K,
                                                                    float *__restrict__ dx_max) {
    const int kq = K >> 2;
    const int64_t total = (int64_t)M * kq;
    float vmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kSkThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kSkThreads) {
        const int64_t m = i / kq;
        const int k = (int)(i - m * kq) * 4;
        const float g = dz[m * dz_ld];
        const float4 wv = *reinterpret_cast<const float4 *>(w + k);
        float4 r = make_float4(g * wv.x, g * wv.y, g * wv.z, g * wv.w);
        if (mask) {
            const float4 q = *reinterpret_cast<const float4 *>(mask + m * mask_ld + k);
            r.x = q.x > 0.f ? r.x : 0.f; r.y = q.y > 0.f ? r.y : 0.f; r.z = q.z > 0.f ? r.z : 0.f; r.w = q.w > 0.f ? r.w : 0.f;
        }
        *reinterpret_cast<float4 *>(dx + m * dx_ld + k) = r;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
    if (dx_max != nullptr) fold_absmax_block(vmax, dx_max);
}

// The one-column last layer of a training step in one sweep over its input: y = act(x . w + b) (the prediction, summed
// in gemv_fwd_kernel's order), d(loss)/dy of the sample, and dx = dy w where x > 0 (x is a ReLU output).  A group of 16
// lanes owns a row and keeps its KI float4 pieces in registers between the dot product and the gradient; the
// separate kernels read x twice and are three launches on the step's critical path.
template <bool BCE, int KI>
__global__ __launch_bounds__(kSkThreads) void gemv_fwd_loss_bwd_kernel(const float *__restrict__ x, int64_t x_ld,
                                                                        const float *__restrict__ w, const float *__restrict__ bias,
                                                                        const float *__restrict__ target, int64_t mean_over,
                                                                        float *__restrict__ y, float *__restrict__ dy,
                                                                        float *__restrict__ dx, int64_t dx_ld, int M, int K,
                                                                        int relu, int relu_mask, float *__restrict__ dx_max) {
    const int lig = threadIdx.x & 15;
    const float b = bias ? bias[0] : 0.f;
    const float inv_n = 1.0f / (float)mean_over;
    float4 wv[KI];
#pragma unroll
    for (int i = 0; i < KI; i++) {
        const int k = lig * 4 + 64 * i;
        wv[i] = k < K ? *reinterpret_cast<const float4 *>(w + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float vmax = 0.f;
    for (int m = blockIdx.x * (kSkThreads / 16) + (threadIdx.x >> 4); m < M; m += gridDim.x * (kSkThreads / 16)) {
        const float *xr = x + (int64_t)m * x_ld;
        float4 xv[KI];
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int k = lig * 4 + 64 * i;
            xv[i] = k < K ? *reinterpret_cast<const float4 *>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float t = target[m];
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < KI; i++)
            if (lig * 4 + 64 * i < K) {
                a = fmaf(xv[i].x, wv[i].x, a); a = fmaf(xv[i].y, wv[i].y, a);
                a = fmaf(xv[i].z, wv[i].z, a); a = fmaf(xv[i].w, wv[i].w, a);
            }
        // (a butterfly: both lanes of a pair add the same two numbers, so all 16 end with the same bits)
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
        a += b;
        a = relu ? fmaxf(a, 0.f) : a;
        float term;
        const float g = loss_grad<BCE>(a, t, inv_n, relu_mask, term);
        if (lig == 0) {
            y[m] = a;
            dy[m] = g;
        }
        float *dr = dx + (int64_t)m * dx_ld;
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int k = lig * 4 + 64 * i;
            if (k < K) {
                float4 r = make_float4(g * wv[i].x, g * wv[i].y, g * wv[i].z, g * wv[i].w);
                r.x = xv[i].x > 0.f ? r.x : 0.f; r.y = xv[i].y > 0.f ? r.y : 0.f;
                r.z = xv[i].z > 0.f ? r.z : 0.f; r.w = xv[i].w > 0.f ? r.w : 0.f;
                *reinterpret_cast<float4 *>(dr + k) = r;
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            }
        }
    }
    if (dx_max != nullptr) fold_absmax_block(vmax, dx_max);
}

// thread (row lane, column quad): K/4 quads side by side, 256 / (K/4) row lanes; partial sums meet in LDS
__global__ __launch_bounds__(kSkThreads) void gemv_bwd_weight_kernel(const float *__restrict__ dz, int64_t dz_ld,
                                                                      const float *__restrict__ x, int64_t x_ld,
                                                                      float *__restrict__ slabs, float *__restrict__ db_slabs,
                                                                      int M, int K, int rows_per_block) {
    __shared__ float4 part[kSkThreads];
    __shared__ float partb[kSkThreads];
    const int kq = K >> 2, rl_n = kSkThreads / kq;
    const int c = threadIdx.x % kq, rl = threadIdx.x / kq;
    const int m_begin = blockIdx.x * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accb = 0.f;
    int m = m_begin + rl;
    for (; m + 3 * rl_n < m_end; m += 4 * rl_n) {          // four rows in flight
        float g[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            g[u] = dz[(int64_t)(m + u * rl_n) * dz_ld];
            v[u] = *reinterpret_cast<const float4 *>(x + (int64_t)(m + u * rl_n) * x_ld + c * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc.x = fmaf(g[u], v[u].x, acc.x); acc.y = fmaf(g[u], v[u].y, acc.y);
            acc.z = fmaf(g[u], v[u].z, acc.z); acc.w = fmaf(g[u], v[u].w, acc.w);
            accb += g[u];
        }
    }
    for (; m < m_end; m += rl_n) {
        const float g = dz[(int64_t)m * dz_ld];
        const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)m * x_ld + c * 4);
        acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
        accb += g;
    }
    part[threadIdx.x] = acc;
    partb[threadIdx.x] = accb;
    __syncthreads();
    if (rl == 0) {
        for (int q = 1; q < rl_n; q++) {                    // fixed order: deterministic
            const float4 o = part[q * kq + c];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *reinterpret_cast<float4 *>(slabs + (int64_t)blockIdx.x * K + c * 4) = acc;
        if (c == 0 && db_slabs) {
            for (int q = 1; q < rl_n; q++) accb += partb[q * kq];
            db_slabs[blockIdx.x] = accb;
        }
    }
}

static bool aligned16p(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

// DLRM_SKINNY=0 sends every layer through the tile kernels (for A/B timing)
static bool enabled() {
    static const bool on = [] { const char *e = getenv("DLRM_SKINNY"); return !(e && e[0] == '0'); }();
    return on;
}

bool skinny_k_forward_ok(int N, int K) { return enabled() && K <= kSkinnyK && N >= 64; }

int skinny_k_forward(const float *x, int64_t x_ld, const float *w, int64_t w_ld, const float *bias, float *y, int64_t y_ld,
                     int M, int N, int K, int relu, float *y_max, hipStream_t s) {
    int rows_per_block = 64;
    while (ceil_div(M, rows_per_block) > 32768) rows_per_block *= 2;     // grid.y stays below the launch limit
    dim3 grid((unsigned)ceil_div(N, kSkThreads), (unsigned)ceil_div(M, rows_per_block));
    hipLaunchKernelGGL(skinny_k_fwd_kernel, grid, dim3(kSkThreads), 0, s, x, x_ld, w, w_ld, bias, y, y_ld, M, N, K, relu,
                       rows_per_block, y_max);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

int skinny_k_backward_weight(const float *dz, int64_t dz_ld, const float *x, int64_t x_ld, float *slabs, float *db_slabs,
                             int M, int N, int K, int max_chunks, int *chunks, hipStream_t s) {
    // four LDS trips per workgroup while that still gives every compute unit four workgroups (a workgroup's trips
    // are a chain of dependent round trips to memory: small batches want more, shorter chains), else two, else one
    const int64_t gx = ceil_div(N, kSkThreads);
    int trips = 4;
    while (trips > 1 && ceil_div(M, trips * kSkRows) * gx < 1024) trips /= 2;
    int want = (int)ceil_div(M, trips * kSkRows);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    const int rows_per_block = (int)(ceil_div(ceil_div(M, want), kSkRows) * kSkRows);
    const int n_chunks = (int)ceil_div(M, rows_per_block);
    dim3 grid((unsigned)ceil_div(N, kSkThreads), (unsigned)n_chunks);
    hipLaunchKernelGGL(skinny_k_bwd_weight_kernel, grid, dim3(kSkThreads), 0, s, dz, dz_ld, x, x_ld, slabs, db_slabs, M, N, K,
                       rows_per_block);
    DLRM_LAUNCH_CHECK();
    *chunks = n_chunks;
    return DLRM_OK;
}

bool gemv_ok(const void *x, int64_t x_ld, const void *w, int K) {
    return enabled() && K % 4 == 0 && x_ld % 4 == 0 && aligned16p(x) && aligned16p(w);
}

int gemv_forward(const float *x, int64_t x_ld, const float *w, const float *bias, float *y, int64_t y_ld, int M, int K, int relu,
                 float *y_max, hipStream_t s) {
    const int64_t blocks = ceil_div(M, kSkThreads / 16);
    hipLaunchKernelGGL(gemv_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kSkThreads), 0, s, x, x_ld, w, bias,
                       y, y_ld, M, K, relu, y_max);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

int gemv_backward_data(const float *dz, int64_t dz_ld, const float *w, float *dx, int64_t dx_ld, const float *mask,
                       int64_t mask_ld, int M, int K, float *dx_max, hipStream_t s) {
    const int64_t blocks = ceil_div((int64_t)M * (K / 4), kSkThreads);
    const int64_t cap = dx_max ? 2048 : 8192;     // (fewer, longer workgroups when each ends with an atomic)
    hipLaunchKernelGGL(gemv_bwd_data_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(kSkThreads), 0, s, dz, dz_ld, w,
                       dx, dx_ld, mask, mask_ld, M, K, dx_max);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

bool gemv_forward_loss_backward_ok(const void *x, int64_t x_ld, const void *w, const void *dx, int64_t dx_ld, int K) {
    return gemv_ok(x, x_ld, w, K) && gemv_ok(dx, dx_ld, w, K) && K <= 1024;
}

template <bool BCE, int KI>
static void launch_fwd_loss_bwd(unsigned grid, hipStream_t s, const float *x, int64_t x_ld, const float *w, const float *bias,
                                const float *target, int64_t mean_over, float *y, float *dy, float *dx, int64_t dx_ld, int M, int K,
                                int relu, int relu_mask, float *dx_max) {
    hipLaunchKernelGGL((gemv_fwd_loss_bwd_kernel<BCE, KI>), dim3(grid), dim3(kSkThreads), 0, s, x, x_ld, w, bias, target, mean_over,
                       y, dy, dx, dx_ld, M, K, relu, relu_mask, dx_max);
}

int gemv_forward_loss_backward(const float *x, int64_t x_ld, const float *w, const float *bias, const float *target,
                               int64_t mean_over, bool bce, float *y, float *dy, float *dx, int64_t dx_ld, int M, int K, int relu,
                               int relu_mask, float *dx_max, hipStream_t s) {
    const int64_t blocks = ceil_div(M, kSkThreads / 16);
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
#define DLRM_FLB(B, KI) launch_fwd_loss_bwd<B, KI>(grid, s, x, x_ld, w, bias, target, mean_over, y, dy, dx, dx_ld, M, K, relu, relu_mask, dx_max)
    if (K <= 256) { if (bce) DLRM_FLB(true, 4); else DLRM_FLB(false, 4); }
    else if (K <= 512) { if (bce) DLRM_FLB(true, 8); else DLRM_FLB(false, 8); }
    else { if (bce) DLRM_FLB(true, 16); else DLRM_FLB(false, 16); }
#undef DLRM_FLB
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

bool gemv_backward_weight_ok(const void *x, int64_t x_ld, int K) {
    return enabled() && K % 4 == 0 && K >= 4 && K <= 4 * kSkThreads && kSkThreads % (K / 4) == 0 && x_ld % 4 == 0 && aligned16p(x);
}

int gemv_backward_weight(const float *dz, int64_t dz_ld, const float *x, int64_t x_ld, float *slabs, float *db_slabs, int M,
                         int K, int max_chunks, int *chunks, hipStream_t s) {
    const int rl_n = kSkThreads / (K / 4);
    int rows_per_lane = 16;                                 // sixteen rows per lane, fewer when that leaves most of the chip idle
    while (rows_per_lane > 2 && ceil_div(M, rows_per_lane * rl_n) < 256) rows_per_lane /= 2;
    int want = (int)ceil_div(M, rows_per_lane * rl_n);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    const int rows_per_block = (int)ceil_div(M, want);
    const int n_chunks = (int)ceil_div(M, rows_per_block);
    hipLaunchKernelGGL(gemv_bwd_weight_kernel, dim3((unsigned)n_chunks), dim3(kSkThreads), 0, s, dz, dz_ld, x, x_ld, slabs,
                       db_slabs, M, K, rows_per_block);
    DLRM_LAUNCH_CHECK();
    *chunks = n_chunks;
    return DLRM_OK;
}

}  // namespace dlrm
