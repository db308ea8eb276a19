// Shared host/device helpers of libdlrm_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dlrm_amd.h"

namespace dlrm {

// thread-local message returned by dlrm_last_error()
char *err_buf();
int set_error(int code, const char *fmt, ...);

#define DLRM_REQUIRE(cond, ...)                                          \
    do {                                                                 \
        if (!(cond)) return ::dlrm::set_error(DLRM_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define DLRM_HIP(call)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return ::dlrm::set_error(DLRM_ERR_HIP, "%s failed: %s (%s:%d)", #call,        \
                                     hipGetErrorString(e_), __FILE__, __LINE__);          \
    } while (0)

#define DLRM_LAUNCH_CHECK() DLRM_HIP(hipGetLastError())

static inline hipStream_t as_stream(dlrm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// bf16 <-> f32 on raw bits (round to nearest even; NaN stays NaN)
__device__ inline float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ inline uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Exact split of float32 values into three bf16 pieces, x = h + m + l up to 2^-27 |x| (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m)): products of such pieces are exact in float32, which is how float32 products run on the
// bf16 matrix pipe (six leading cross terms of the pieces, float32 accumulation).
// a pair of float32 -> the pair's three bf16 pieces (each packed in one dword), 9 VALU operations
__device__ inline void split3_pair(f32x2 x, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hb = __builtin_convertvector(x, bf16x2);
    h = __builtin_bit_cast(unsigned, hb);
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    const f32x2 r1 = x - hf;
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    m = __builtin_bit_cast(unsigned, mb);
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    l = __builtin_bit_cast(unsigned, lb);
}

// Two-piece split on the f16 matrix pipe: xs = x * scale with scale a power of two that puts the operand's largest
// magnitude into [2^13, 2^14) (so h cannot overflow f16), h = f16(xs), l = f16((xs - h) * 2^11): the low piece is kept
// 2^11 times too large, so that it stays a normal f16 number for every |xs| >= 2^-14.  h + l * 2^-11 then carries 22
// mantissa bits of every element within 2^27 of the operand's largest (without the factor: within 2^15), and an absolute
// error of at most 2^-49 of the largest magnitude for the rest (f16 subnormals).  The product is h_a h_b in one
// accumulator and h_a l_b + l_a h_b in a second one that enters the result times 2^-11; |l| <= 2^13, nothing overflows.
// Representation error near one float32 ulp (7.5e-8 relative rms against float64, below the accumulation error of any
// float32 GEMM).
constexpr float kSplit2LowScale = 2048.f, kSplit2LowInv = 1.f / 2048.f;
// a pair of float32 -> the pair's two f16 pieces (each packed in one dword)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ inline void split2_pair(f32x2 x, float scale, unsigned &h, unsigned &l) {
    const f32x2 xs = x * scale;
    const f16x2 hh = __builtin_convertvector(xs, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    // x * scale - h in one mixed-precision fma per element (v_fma_mix_f32 reads the f16 piece directly; the product
    // with a power of two is exact and so is the difference: h holds the leading bits of xs)
    const f32x2 r = {__builtin_fmaf(x[0], scale, -(float)hh[0]) * kSplit2LowScale,
                     __builtin_fmaf(x[1], scale, -(float)hh[1]) * kSplit2LowScale};
    const f16x2 lh = __builtin_convertvector(r, f16x2);
    l = __builtin_bit_cast(unsigned, lh);
}

// max |x| of several [rows, cols] matrices (row stride ld) folded into their slots by one launch per kAbsmaxMany
// matrices (gemm_f32.hip; the hosted model measures its weight matrices with it)
constexpr int kAbsmaxMany = 16;
struct AbsmaxItem {
    const float *x;
    int64_t rows;
    int cols;
    int64_t ld;
    float *slot;
};
int absmax_many(const AbsmaxItem *items, int n, hipStream_t s);

// Weight gradients whose partial sums are added up LATER, several layers by one launch (gemm_f32.hip): the products of
// a stack's layers leave their slabs in workspaces of their own, reduce_slabs_many adds all of them in the fixed order
// dlrm_linear_backward_weight uses (same results, bit for bit).
struct SlabSum {          // out[r * ld_out + c] = sum_s slabs[s * stride + r * cols + c]
    const float *slabs;
    int64_t stride;
    int rows, cols;
    float *out;
    int64_t ld_out;
};
struct SlabReduce {       // what one layer leaves to do: its weight gradient and (has_b) its bias gradient; splits 0 = nothing
    SlabSum w, b;
    int splits;
    int has_b;
};
int linear_backward_weight_deferred(const dlrm_linear_bwd *l, void *d_workspace, size_t workspace_bytes, SlabReduce *pending,
                                    dlrm_stream_t stream);
int reduce_slabs_many(const SlabReduce *items, int n, hipStream_t s);

// Stable sort of (key, value) pairs by the low `bits` bits of the key (pair_sort.hip: least-significant-digit radix sort,
// digits of up to 10 bits).  The pairs start in (keys_a, vals_a) and end there when pair_sort_passes(bits) is even, in
// (keys_b, vals_b) when it is odd; both pairs of buffers are overwritten.  tmp: pair_sort_tmp_bytes(n) bytes.
int pair_sort_passes(int bits);
size_t pair_sort_tmp_bytes(int64_t n);
int pair_sort(uint32_t *keys_a, uint32_t *keys_b, int32_t *vals_a, int32_t *vals_b, int64_t n, int bits, void *tmp,
              size_t tmp_bytes, hipStream_t s);
int pair_sort(uint64_t *keys_a, uint64_t *keys_b, int32_t *vals_a, int32_t *vals_b, int64_t n, int bits, void *tmp,
              size_t tmp_bytes, hipStream_t s);

// dlrm_interact_dot_forward_gather / dlrm_interact_dot_backward with one more output (interact.hip): the largest
// magnitude of the interaction's output / of feature 0's gradient, folded into a slot like the GEMM epilogues do
int interact_dot_forward_gather_max(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, float *d_out, int64_t out_ld,
                                    float *d_x_copy, float *d_out_max, dlrm_stream_t stream);
int interact_dot_backward_gather_max(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, const float *d_dout, int64_t dout_ld,
                                     float *d_dx, int32_t relu_mask_feature0, float *d_dx0_max, dlrm_stream_t stream);
int interact_dot_backward_max(const float *d_x, const float *d_dout, int64_t dout_ld, int64_t B, int32_t F, int32_t D,
                              float *d_dx, int32_t relu_mask_feature0, float *d_dx0_max, dlrm_stream_t stream);
// ... and the same for the kernels that read / write the features through views
int interact_dot_forward_views_max(const dlrm_feature_view *x_views, int64_t B, int32_t F, int32_t D, float *d_out, int64_t out_ld,
                                   float *d_out_max, dlrm_stream_t stream);
int interact_dot_backward_views_max(const dlrm_feature_view *x_views, const float *d_dout, int64_t dout_ld, int64_t B, int32_t F,
                                    int32_t D, const dlrm_feature_view *dx_views, int32_t relu_mask_feature0, float *d_dx0_max,
                                    dlrm_stream_t stream);

// Folds the largest magnitude a wave has seen (per-lane `vmax` >= 0) into *slot: one wave reduction, at most one atomic
// per wave and only if it can still raise the slot (thousands of atomics on one address queue up behind each other;
// a stale read costs one atomic too many, never a wrong result).  Every lane of the wave must call it.
__device__ inline void fold_absmax(float vmax, float *slot) {
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned *>(slot))
        atomicMax(reinterpret_cast<unsigned *>(slot), __float_as_uint(vmax));
}

// The same for a whole 256-thread workgroup: one atomic per workgroup (kernels whose workgroups all finish at the same
// moment would otherwise all find the slot still empty).  Every thread of the workgroup must call it.
__device__ inline void fold_absmax_block(float vmax, float *slot) {
    __shared__ float part[4];
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        vmax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (__float_as_uint(vmax) > *reinterpret_cast<volatile unsigned *>(slot))
            atomicMax(reinterpret_cast<unsigned *>(slot), __float_as_uint(vmax));
    }
}

// power-of-two scale for an operand whose largest magnitude has the float32 bits `absmax_bits`: the scaled maximum
// lands in [2^13, 2^14); zero / denormal maxima scale by 2^127, infinities and NaNs by 2^-115 (the product is then
// whatever float32 arithmetic makes of them)
__device__ inline float split2_scale(unsigned absmax_bits) {
    const int e = (int)((absmax_bits >> 23) & 0xffu);
    int se = 267 - e;
    se = se > 254 ? 254 : (se < 1 ? 1 : se);
    return __uint_as_float((unsigned)se << 23);
}

}  // namespace dlrm
