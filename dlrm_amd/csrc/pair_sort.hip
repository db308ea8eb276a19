// Stable sort of (key, value) pairs by key for the sparse update (embbag_bwd.hip): the occurrences of a batch's rows,
// key = global row, value = bag.  Takes the place of the reference's std::sort of a table's ids in create_emb /
// save_emb (dlrm_libtorch/src/dlrmtest.cpp:62-63, 93-94); stability keeps the occurrences of one row in position
// order, which fixes the summation order of their gradients.
//
// The job is small and latency-bound -- 1.7 M pairs at the MLPerf shape, on a stream beside the forward pass -- so the
// sort is shaped for few, short launches rather than for peak bandwidth: least-significant-digit radix sort with
// digits of up to 10 bits (three passes cover 28-bit keys, where 8-bit digits need four), workgroups of 1024 threads
// that own 8192 consecutive positions (208 workgroups at that shape: the chip's other compute units stay with the
// GEMMs), three kernels per pass:
//   digit_hist    per workgroup, the histogram of its positions' digits (LDS atomics) -> H[block][digit], and the
//                 digit totals G[digit] (integer atomics: the result does not depend on their order)
//   digit_offsets H[block][digit] <- where block `block` starts writing digit `digit`: (number of smaller digits
//                 anywhere) + (same digit in earlier blocks); 16 workgroups, each 64 digits
//   digit_scatter every pair to H[block][digit] + (pairs of the same digit earlier in the block).  The rank inside the
//                 block comes from 8 rounds of 1024 positions: lanes of a wave with equal digits find each other with
//                 one ballot per digit bit, the wave's first lane of a digit posts the count, a per-digit pass over the
//                 16 waves turns the counts into starting ranks.
#include "common.h"

namespace dlrm {

constexpr int kSortThreads = 1024, kSortRounds = 8, kSortBlock = kSortThreads * kSortRounds, kDigits = 1024;
constexpr int kSortWaves = kSortThreads / 64;

int pair_sort_passes(int bits) { return bits <= 0 ? 1 : (bits + 9) / 10; }

// digit width of pass `pass` (the bits are spread evenly over the passes)
static int digit_bits(int bits, int pass) {
    const int P = pair_sort_passes(bits);
    const int lo = bits / P, extra = bits % P;        // the first `extra` passes take one bit more
    return lo + (pass < extra ? 1 : 0);
}

size_t pair_sort_tmp_bytes(int64_t n) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, kSortBlock);
    return (size_t)nb * kDigits * sizeof(unsigned) + 3 * kDigits * sizeof(unsigned) + 256;
}

template <typename KEY>
__global__ __launch_bounds__(kSortThreads) void digit_hist_kernel(const KEY *__restrict__ keys, int64_t n, int shift,
                                                                  unsigned mask, unsigned *__restrict__ H,
                                                                  unsigned *__restrict__ G) {
    __shared__ unsigned h[kDigits];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortBlock;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int64_t p = base + r * kSortThreads + tid;
        if (p < n) atomicAdd(&h[(unsigned)(keys[p] >> shift) & mask], 1u);
    }
    __syncthreads();
    const unsigned c = h[tid];
    H[(int64_t)blockIdx.x * kDigits + tid] = c;
    if (c) atomicAdd(&G[tid], c);
}

// grid: kDigits / 64 workgroups of 256 threads; thread (digit dl of the workgroup's 64, quarter q of the blocks)
__global__ __launch_bounds__(256) void digit_offsets_kernel(unsigned *__restrict__ H, const unsigned *__restrict__ G, int nb) {
    __shared__ unsigned sg[kDigits], wsum[4], part[4][64];
    const int tid = threadIdx.x;
    // exclusive prefix of the digit totals (every workgroup computes all of it: 1024 values)
    unsigned v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = G[4 * tid + i]; s += v[i]; }
    unsigned incl = s;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const unsigned o = __shfl_up(incl, k, 64);
        if ((tid & 63) >= k) incl += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned before = incl - s;
    for (int w = 0; w < (tid >> 6); w++) before += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; i++) { sg[4 * tid + i] = before; before += v[i]; }
    __syncthreads();
    const int dl = tid & 63, q = tid >> 6, d = blockIdx.x * 64 + dl;
    const int per = (nb + 3) / 4, b0 = q * per, b1 = min(nb, b0 + per);
    unsigned sum = 0;
    for (int b = b0; b < b1; b++) sum += H[(int64_t)b * kDigits + d];
    part[q][dl] = sum;
    __syncthreads();
    unsigned start = sg[d];
    for (int w = 0; w < q; w++) start += part[w][dl];
    for (int b = b0; b < b1; b++) {
        const unsigned c = H[(int64_t)b * kDigits + d];
        H[(int64_t)b * kDigits + d] = start;
        start += c;
    }
}

template <typename KEY>
__global__ __launch_bounds__(kSortThreads) void digit_scatter_kernel(const KEY *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                                                                     KEY *__restrict__ keys_out, int32_t *__restrict__ vals_out,
                                                                     int64_t n, int shift, int dbits,
                                                                     const unsigned *__restrict__ offsets) {
    __shared__ unsigned run[kDigits];                     // pairs of each digit in the rounds done so far
    __shared__ unsigned short cw[kSortWaves][kDigits];    // this round: per wave the count, then the starting rank
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned mask = (1u << dbits) - 1u;
    const int64_t base = (int64_t)blockIdx.x * kSortBlock;
    KEY key[kSortRounds];
    int32_t val[kSortRounds];
    unsigned dig[kSortRounds], rank[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int64_t p = base + r * kSortThreads + tid;
        const bool ok = p < n;
        key[r] = ok ? keys_in[p] : (KEY)0;
        val[r] = ok ? vals_in[p] : 0;
        dig[r] = (unsigned)(key[r] >> shift) & mask;
    }
    run[tid] = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const bool ok = base + r * kSortThreads + tid < n;
        unsigned *cwz = reinterpret_cast<unsigned *>(&cw[0][0]);
#pragma unroll
        for (int i = 0; i < kSortWaves * kDigits / 2 / kSortThreads; i++) cwz[tid + kSortThreads * i] = 0u;
        // lanes of this wave that hold the same digit
        unsigned long long peers = __ballot(ok);
        for (int b = 0; b < dbits; b++) {
            const bool one = (dig[r] >> b) & 1u;
            const unsigned long long vote = __ballot(ok && one);
            peers &= one ? vote : ~vote;
        }
        const unsigned rw = (unsigned)__popcll(peers & lt);
        __syncthreads();                                   // cw is zero (and `run` of round 0 is)
        if (ok && rw == 0) cw[wave][dig[r]] = (unsigned short)__popcll(peers);
        __syncthreads();
        {
            unsigned p = run[tid];
#pragma unroll
            for (int w = 0; w < kSortWaves; w++) {
                const unsigned c = cw[w][tid];
                cw[w][tid] = (unsigned short)p;
                p += c;
            }
            run[tid] = p;
        }
        __syncthreads();
        rank[r] = ok ? cw[wave][dig[r]] + rw : 0u;
        __syncthreads();                                   // everyone has read its rank before the next round clears cw
    }
    const unsigned *off = offsets + (int64_t)blockIdx.x * kDigits;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        if (base + r * kSortThreads + tid < n) {
            const int64_t dst = (int64_t)off[dig[r]] + rank[r];
            keys_out[dst] = key[r];
            vals_out[dst] = val[r];
        }
    }
}

// Pairs start in (keys_a, vals_a); passes alternate between the two buffer pairs, so after pair_sort_passes(bits)
// passes the sorted pairs are in (keys_b, vals_b) when that number is odd, in (keys_a, vals_a) when it is even.
template <typename KEY>
static int pair_sort_impl(KEY *keys_a, KEY *keys_b, int32_t *vals_a, int32_t *vals_b, int64_t n, int bits, void *tmp,
                          size_t tmp_bytes, hipStream_t s) {
    DLRM_REQUIRE(n >= 0 && bits >= 1 && bits <= (int)sizeof(KEY) * 8, "bad sort request");
    DLRM_REQUIRE(n < (1ll << 32), "more than 2^32 pairs");
    DLRM_REQUIRE(tmp != nullptr && tmp_bytes >= pair_sort_tmp_bytes(n), "sort scratch too small");
    if (n == 0) return DLRM_OK;
    const int P = pair_sort_passes(bits);
    const int64_t nb = ceil_div(n, kSortBlock);
    unsigned *H = static_cast<unsigned *>(tmp);
    unsigned *G = H + nb * kDigits;
    DLRM_HIP(hipMemsetAsync(G, 0, 3 * kDigits * sizeof(unsigned), s));
    int shift = 0;
    for (int pass = 0; pass < P; pass++) {
        const int db = digit_bits(bits, pass);
        const KEY *ki = pass & 1 ? keys_b : keys_a;
        const int32_t *vi = pass & 1 ? vals_b : vals_a;
        KEY *ko = pass & 1 ? keys_a : keys_b;
        int32_t *vo = pass & 1 ? vals_a : vals_b;
        unsigned *Gp = G + (pass % 3) * kDigits;
        if (pass >= 3) DLRM_HIP(hipMemsetAsync(Gp, 0, kDigits * sizeof(unsigned), s));
        hipLaunchKernelGGL((digit_hist_kernel<KEY>), dim3((unsigned)nb), dim3(kSortThreads), 0, s, ki, n, shift,
                           (1u << db) - 1u, H, Gp);
        hipLaunchKernelGGL(digit_offsets_kernel, dim3(kDigits / 64), dim3(256), 0, s, H, Gp, (int)nb);
        hipLaunchKernelGGL((digit_scatter_kernel<KEY>), dim3((unsigned)nb), dim3(kSortThreads), 0, s, ki, vi, ko, vo, n, shift,
                           db, H);
        DLRM_LAUNCH_CHECK();
        shift += db;
    }
    return DLRM_OK;
}

int pair_sort(uint32_t *keys_a, uint32_t *keys_b, int32_t *vals_a, int32_t *vals_b, int64_t n, int bits, void *tmp,
              size_t tmp_bytes, hipStream_t s) {
    return pair_sort_impl<uint32_t>(keys_a, keys_b, vals_a, vals_b, n, bits, tmp, tmp_bytes, s);
}
int pair_sort(uint64_t *keys_a, uint64_t *keys_b, int32_t *vals_a, int32_t *vals_b, int64_t n, int bits, void *tmp,
              size_t tmp_bytes, hipStream_t s) {
    return pair_sort_impl<uint64_t>(keys_a, keys_b, vals_a, vals_b, n, bits, tmp, tmp_bytes, s);
}

}  // namespace dlrm
