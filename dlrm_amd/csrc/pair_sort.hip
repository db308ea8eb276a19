// Stable sort of (key, value) pairs by key for the sparse update (embbag_bwd.hip): the occurrences of a batch's rows,
// key = global row, value = bag.  Takes the place of the reference's std::sort of a table's ids in create_emb /
// save_emb (dlrm_libtorch/src/dlrmtest.cpp:62-63, 93-94); stability keeps the occurrences of one row in position
// order, which fixes the summation order of their gradients.
//
// The job is small and latency-bound -- 1.7 M pairs at the MLPerf shape -- and it runs on a stream of its own beside
// the forward pass, whose GEMM workgroups leave a compute unit 64 registers per lane and SIMD, 32 KB of LDS and most of
// its wave slots.  So the sort is shaped to FIT BESIDE them rather than for peak bandwidth: workgroups of 256 threads
// (one wave per SIMD, under 64 registers, under 32 KB of LDS with 32-bit keys) that own 2048 consecutive positions,
// least-significant-digit radix sort with digits of up to 10 bits (three passes cover 28-bit keys, where 8-bit digits
// need four), three kernels per pass:
//   digit_hist    per workgroup, the histogram of its positions' digits (LDS atomics) -> H[block][digit], and the
//                 digit totals G[digit] (integer atomics: the result does not depend on their order)
//   digit_offsets H[block][digit] <- where block `block` starts writing digit `digit`: (number of smaller digits
//                 anywhere) + (same digit in earlier blocks); 64 workgroups, each 16 digits
//   digit_scatter every pair to H[block][digit] + (pairs of the same digit earlier in the block).  Wave w owns positions
//                 512 w .. 512 w + 511 of the block, 64 at a time: lanes with equal digits find each other with one
//                 ballot per digit bit, the first of them advances the wave's own counter of that digit in LDS -- no
//                 barrier while ranking.  One pass over the four waves' counters and a scan over the digits turn the
//                 ranks into positions inside the block; the pairs go to LDS in that order and leave it with
//                 consecutive threads writing consecutive addresses of a digit's run.
#include "common.h"

namespace dlrm {

constexpr int kSortThreads = 256, kSortRounds = 8, kSortBlock = kSortThreads * kSortRounds, kDigits = 1024;
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
#pragma unroll
    for (int i = 0; i < kDigits / kSortThreads; i++) h[tid + kSortThreads * i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortBlock;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int64_t p = base + r * kSortThreads + tid;
        if (p < n) atomicAdd(&h[(unsigned)(keys[p] >> shift) & mask], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kDigits / kSortThreads; i++) {
        const int d = tid + kSortThreads * i;
        const unsigned c = h[d];
        H[(int64_t)blockIdx.x * kDigits + d] = c;
        if (c) atomicAdd(&G[d], c);
    }
}

// grid: kDigits / 16 workgroups of 256 threads; thread (digit dl of the workgroup's 16, part q of 16 of the blocks)
__global__ __launch_bounds__(256) void digit_offsets_kernel(unsigned *__restrict__ H, const unsigned *__restrict__ G, int nb) {
    __shared__ unsigned sg[kDigits], wsum[4], part[16][16];
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
    const int dl = tid & 15, q = tid >> 4, d = blockIdx.x * 16 + dl;
    const int per = (nb + 15) / 16, b0 = q * per, b1 = min(nb, b0 + per);
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
    __shared__ unsigned short cnt[kSortWaves][kDigits];   // per wave: pairs of each digit so far, later their first rank
    __shared__ unsigned short dstart[kDigits];            // first position of each digit inside the sorted block
    __shared__ unsigned wtot[kSortWaves];
    __shared__ KEY skey[kSortBlock];
    __shared__ int32_t sval[kSortBlock];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned mask = (1u << dbits) - 1u;
    const int64_t base = (int64_t)blockIdx.x * kSortBlock;
    const int n_blk = (int)min((int64_t)kSortBlock, n - base);
    unsigned dig[kSortRounds];
    unsigned short rank[kSortRounds];
    // wave w owns positions [512 w, 512 w + 512) of the block, 64 consecutive ones per round.  Only the digits stay in
    // registers while the ranks are worked out (the kernel is to fit into 64 registers, beside the GEMM workgroups of
    // another stream); keys and values are read again when they move to LDS, out of the cache by then.
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = wave * (kSortBlock / kSortWaves) + r * 64 + lane;
        dig[r] = i < n_blk ? (unsigned)(keys_in[base + i] >> shift) & mask : 0u;
    }
    {
        unsigned *z = reinterpret_cast<unsigned *>(&cnt[0][0]);
#pragma unroll
        for (int i = 0; i < kSortWaves * kDigits / 2 / kSortThreads; i++) z[tid + kSortThreads * i] = 0u;
    }
    __syncthreads();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const bool ok = wave * (kSortBlock / kSortWaves) + r * 64 + lane < n_blk;
        const unsigned dg = dig[r];
        unsigned long long peers = __ballot(ok);       // lanes of this wave that hold the same digit
        for (int b = 0; b < dbits; b++) {
            const bool one = (dg >> b) & 1u;
            const unsigned long long vote = __ballot(ok && one);
            peers &= one ? vote : ~vote;
        }
        const int leader = ok ? __ffsll((long long)peers) - 1 : lane;
        unsigned before = 0;
        if (ok && lane == leader) {                    // (a wave's LDS operations keep their order: no barrier)
            before = cnt[wave][dg];
            cnt[wave][dg] = (unsigned short)(before + __popcll(peers));
        }
        before = __shfl(before, leader, 64);
        rank[r] = (unsigned short)(before + __popcll(peers & lt));
    }
    __syncthreads();
    // per digit: the waves' counts become the rank of each wave's first pair of that digit; the digit's total is scanned
    // over the digits (thread t: digits 4 t .. 4 t + 3)
    unsigned tot[4], s4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int d = 4 * tid + i;
        unsigned s = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) {
            const unsigned c = cnt[w][d];
            cnt[w][d] = (unsigned short)s;
            s += c;
        }
        tot[i] = s;
        s4 += s;
    }
    unsigned incl = s4;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const unsigned o = __shfl_up(incl, k, 64);
        if (lane >= k) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned before4 = incl - s4;
    for (int w = 0; w < wave; w++) before4 += wtot[w];
#pragma unroll
    for (int i = 0; i < 4; i++) { dstart[4 * tid + i] = (unsigned short)before4; before4 += tot[i]; }
    __syncthreads();
    // the pairs into LDS in sorted order ...
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = wave * (kSortBlock / kSortWaves) + r * 64 + lane;
        if (i < n_blk) {
            const unsigned dg = dig[r];
            const int pos = dstart[dg] + cnt[wave][dg] + rank[r];
            skey[pos] = keys_in[base + i];
            sval[pos] = vals_in[base + i];
        }
    }
    __syncthreads();
    // ... and out of it: consecutive threads, consecutive addresses of a digit's run
    const unsigned *off = offsets + (int64_t)blockIdx.x * kDigits;
    for (int i = tid; i < n_blk; i += kSortThreads) {
        const KEY k = skey[i];
        const unsigned dg = (unsigned)(k >> shift) & mask;
        const int64_t dst = (int64_t)off[dg] + (i - dstart[dg]);
        keys_out[dst] = k;
        vals_out[dst] = sval[i];
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
        hipLaunchKernelGGL(digit_offsets_kernel, dim3(kDigits / 16), dim3(256), 0, s, H, Gp, (int)nb);
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
