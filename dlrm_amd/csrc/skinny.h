// Layers that are no matrix product: a 13-wide input (Criteo's dense features) or a single output (the
// click logit).  They are bound by reading or writing their one large operand; the MFMA tile kernels
// spend a whole 128 x 128 x 32 step on them.
#pragma once
#include "common.h"

namespace dlrm {

constexpr int kSkinnyK = 16;   // widest reduction the column-per-thread kernels take

// y = act(x W^T + b) with K <= kSkinnyK; y_max (or NULL): max |y| is folded into it
bool skinny_k_forward_ok(int N, int K);
int skinny_k_forward(const float *x, int64_t x_ld, const float *w, int64_t w_ld, const float *bias, float *y, int64_t y_ld,
                     int M, int N, int K, int relu, float *y_max, hipStream_t s);
// partial sums of dW = dz^T x and db = column sums of dz over `chunks` row ranges (chunks <= max_chunks):
// slabs [chunks][N][K], db_slabs [chunks][N] (or NULL).  The caller adds the chunks up.
int skinny_k_backward_weight(const float *dz, int64_t dz_ld, const float *x, int64_t x_ld, float *slabs, float *db_slabs,
                             int M, int N, int K, int max_chunks, int *chunks, hipStream_t s);

// N == 1: y[m] = act(x[m,:] . w + b); y_max (or NULL): max |y| is folded into it
bool gemv_ok(const void *x, int64_t x_ld, const void *w, int K);
int gemv_forward(const float *x, int64_t x_ld, const float *w, const float *bias, float *y, int64_t y_ld, int M, int K, int relu,
                 float *y_max, hipStream_t s);
// dx[m,k] = dz[m] w[k], zero where mask[m,k] <= 0; dx_max (or NULL): max |dx| is folded into it
int gemv_backward_data(const float *dz, int64_t dz_ld, const float *w, float *dx, int64_t dx_ld, const float *mask,
                       int64_t mask_ld, int M, int K, float *dx_max, hipStream_t s);
// the last layer of a training step in one launch: y[m] = act(x[m,:] . w + b), dy[m] = d(mean loss)/dy[m] against
// target[m] (the mean runs over mean_over samples; relu_mask: the squared error's gradient is zero where y <= 0),
// dx[m,k] = dy[m] w[k] where x[m,k] > 0, else zero; K <= 1024.  The three results carry the bits of gemv_forward,
// the loss gradient kernel and gemv_backward_data with x as the mask.
bool gemv_forward_loss_backward_ok(const void *x, int64_t x_ld, const void *w, const void *dx, int64_t dx_ld, int K);
int gemv_forward_loss_backward(const float *x, int64_t x_ld, const float *w, const float *bias, const float *target,
                               int64_t mean_over, bool bce, float *y, float *dy, float *dx, int64_t dx_ld, int M, int K, int relu,
                               int relu_mask, float *dx_max, hipStream_t s);
// partial sums of dw[k] = sum_m dz[m] x[m,k] and db = sum_m dz[m]: slabs [chunks][K], db_slabs [chunks]
bool gemv_backward_weight_ok(const void *x, int64_t x_ld, int K);
int gemv_backward_weight(const float *dz, int64_t dz_ld, const float *x, int64_t x_ld, float *slabs, float *db_slabs, int M,
                         int K, int max_chunks, int *chunks, hipStream_t s);

}  // namespace dlrm
