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

}  // namespace dlrm
