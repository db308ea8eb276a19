// EmbeddingBag sparse backward fused with the row-wise optimizer step (gfx950).
//
//  1. expand : every occurrence p of a row in the batch becomes a pair
//              (key = global row, value = bag).
//  2. sort   : stable LSD radix sort of the pairs by key (pair_sort.hip, only as
//              many key bits as total_rows needs; DLRM_SORT=rocprim in the
//              environment: rocPRIM's device primitive instead, for comparison).
//              Stability keeps the occurrences of one row in position order, so
//              the reduction order -- and the result -- is reproducible.
//  3. update : segmented reduction over the sorted occurrences fused with the optimizer step
//              (sparse_update_kernel).  A workgroup owns one or two stretches of kSegLen sorted
//              positions, each G-lane group a few consecutive positions of it, fetched kChunk gradient
//              rows (and the table rows of the runs ending there) at a time.  A run of equal keys inside
//              one group is summed in position order and applied there (SGD or row-wise Adagrad); pieces
//              of a run that crosses groups meet in LDS and the group where the run ends applies it;
//              pieces of a run that crosses a stretch boundary go to the `partial` slabs of the workspace.
//  4. combine: combine_runs_kernel adds the slabs of each run that crossed a stretch boundary, in
//              position order, and applies the step.  Every touched row is read once and written once.
//
// HBM-bound: algorithmic bytes per call = n_idx*dim*4 (gradient rows) +
// 2*U*dim*row_bytes (read-modify-write of U unique rows) + n_idx*idx_bytes.
#include <stdlib.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace dlrm {

int validate_embbag(const dlrm_embbag *e);

constexpr int kBwdThreads = 256;

#ifdef DLRM_PK_PROBE
// Diagnostic build only (tools/build_pk_probe.sh; never part of dlrm_amd/lib): every accumulation of the update is
// recomputed with one scalar v_fma_f32 on the same registers; where the two disagree the operands, both results and
// the gradient element read from memory once more are recorded.  Tells a packed instruction that returned a wrong
// half from an operand register that did not hold the loaded value yet.
constexpr int kProbeRecords = 256;
__device__ unsigned g_pk_probe_count;
__device__ unsigned g_pk_probe[kProbeRecords][12];
#endif

template <typename KEY>
struct WsLayout {
    KEY *keys_in, *keys_out;
    int32_t *vals_in, *vals_out;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    float *partial;   // [2 * ceil(n / kSegLen)][dpad] partial row gradients of runs cut at segment boundaries
    size_t total;
};

static inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

static inline int key_bits(int64_t total_rows) {
    int b = 1;
    while ((total_rows >> b) != 0) b++;
    return b;
}

static inline int pad_dim(int dim) { return (dim + 3) / 4 * 4; }

template <typename KEY>
static int ws_layout(int64_t n, int bits, int dim, void *base, WsLayout<KEY> *L) {
    size_t tmp = 0;
    hipError_t err = rocprim::radix_sort_pairs(nullptr, tmp, (KEY *)nullptr, (KEY *)nullptr, (int32_t *)nullptr,
                                               (int32_t *)nullptr, (size_t)n, 0u, (unsigned)bits, (hipStream_t)0,
                                               false);
    if (err != hipSuccess) return set_error(DLRM_ERR_HIP, "radix_sort_pairs size query: %s", hipGetErrorString(err));
    char *p = static_cast<char *>(base);
    size_t o = 0;
    L->keys_in = reinterpret_cast<KEY *>(p + o); o += align_up(sizeof(KEY) * (size_t)n);
    L->keys_out = reinterpret_cast<KEY *>(p + o); o += align_up(sizeof(KEY) * (size_t)n);
    L->vals_in = reinterpret_cast<int32_t *>(p + o); o += align_up(sizeof(int32_t) * (size_t)n);
    L->vals_out = reinterpret_cast<int32_t *>(p + o); o += align_up(sizeof(int32_t) * (size_t)n);
    const size_t own = pair_sort_tmp_bytes(n);
    tmp = tmp > own ? tmp : own;                      // (room for either sort)
    L->sort_tmp = p + o; o += align_up(tmp);
    L->sort_tmp_bytes = tmp;
    L->partial = reinterpret_cast<float *>(p + o);
    o += align_up(sizeof(float) * 2 * (size_t)((n + 127) / 128) * (size_t)pad_dim(dim));
    L->total = o;
    return DLRM_OK;
}

struct BwdParams {
    void *w;
    const int64_t *tro;
    int64_t total_rows;
    int T;
    int D;
    const void *idx;
    const void *off;
    int64_t n_idx;
    int64_t B;
    int64_t n_bags;
    int include_last;
    int mean;
    const float *grad;
    int64_t grad_ld;
    int *err;
    int last_wins;
    int adagrad;
    float lr;
    float eps;
    float *state;
    const int64_t *shard;  // first global row held per table (row-wise shard), or NULL
};

constexpr int kTailBlocks = 16;

// lanes_per_bag lanes walk the positions of one bag (workgroups [0, bag_blocks)).  The index array may
// be longer than the range the offsets address (a descriptor for some of the tables of a batch):
// the kTailBlocks workgroups behind bag_blocks give every position outside that range the key that
// the update skips, so the sort never sees what an earlier call left in the workspace.
template <typename IDX, typename KEY>
__global__ __launch_bounds__(kBwdThreads) void expand_pairs_kernel(BwdParams p, int lanes_per_bag, unsigned bag_blocks,
                                                                    KEY *keys, int32_t *vals) {
    const IDX *off = static_cast<const IDX *>(p.off);
    const IDX *idx = static_cast<const IDX *>(p.idx);
    if (blockIdx.x >= bag_blocks) {
        const int64_t lo = (int64_t)off[0];
        const int64_t hi = p.include_last ? (int64_t)off[p.n_bags] : p.n_idx;
        const int64_t first = (int64_t)(blockIdx.x - bag_blocks) * kBwdThreads + threadIdx.x;
        const int64_t stride = (int64_t)kTailBlocks * kBwdThreads;
        for (int64_t q = first; q < lo && q < p.n_idx; q += stride) { keys[q] = (KEY)p.total_rows; vals[q] = 0; }
        for (int64_t q = (hi > 0 ? hi : 0) + first; q < p.n_idx; q += stride) { keys[q] = (KEY)p.total_rows; vals[q] = 0; }
        return;
    }
    const int64_t tid = (int64_t)blockIdx.x * kBwdThreads + threadIdx.x;
    const int64_t g = tid / lanes_per_bag;
    const int lane = (int)(tid % lanes_per_bag);
    if (g >= p.n_bags) return;
    const int64_t start = (int64_t)off[g];
    const int64_t end = (g + 1 < p.n_bags || p.include_last) ? (int64_t)off[g + 1] : p.n_idx;
    const int t = (int)((uint32_t)g / (uint32_t)p.B);
    const int64_t tbase = p.tro ? p.tro[t] : 0;
    const int64_t trows = p.tro ? p.tro[t + 1] - tbase : p.total_rows;
    const int64_t sbase = p.shard ? p.shard[t] : 0;
    bool bad = false;
    for (int64_t q = start + lane; q < end; q += lanes_per_bag) {
        const int64_t r = (int64_t)idx[q] - sbase;
        KEY k;
        if (r < 0 || r >= trows) {
            k = (KEY)p.total_rows;  // sorts behind every real row and is skipped by the update
            bad = bad || p.shard == nullptr;   // rows of another shard are not an error
        } else {
            k = (KEY)(tbase + r);
        }
        keys[q] = k;
        vals[q] = (int32_t)g;
    }
    if (bad && p.err) *p.err = 1;
}

template <typename ROW, int VEC16>
struct RowRW;

template <>
struct RowRW<float, 1> {
    static constexpr int NF = 4;
    typedef float4 Raw;
    __device__ static Raw load_raw(const void *w, int64_t e) {
        return *reinterpret_cast<const float4 *>(static_cast<const float *>(w) + e);
    }
    __device__ static void unpack(const Raw &x, float (&v)[4]) { v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
    __device__ static void store(void *w, int64_t e, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(static_cast<float *>(w) + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct RowRW<uint16_t, 1> {
    static constexpr int NF = 8;
    typedef uint4 Raw;
    __device__ static Raw load_raw(const void *w, int64_t e) {
        return *reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(w) + e);
    }
    __device__ static void unpack(const Raw &x, float (&v)[8]) {
        v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xffff0000u);
        v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xffff0000u);
        v[4] = __uint_as_float(x.z << 16); v[5] = __uint_as_float(x.z & 0xffff0000u);
        v[6] = __uint_as_float(x.w << 16); v[7] = __uint_as_float(x.w & 0xffff0000u);
    }
    __device__ static void store(void *w, int64_t e, const float (&v)[8]) {
        uint4 x;
        x.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        x.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        x.z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
        x.w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
        *reinterpret_cast<uint4 *>(static_cast<uint16_t *>(w) + e) = x;
    }
};
template <>
struct RowRW<float, 0> {
    static constexpr int NF = 1;
    typedef float Raw;
    __device__ static Raw load_raw(const void *w, int64_t e) { return static_cast<const float *>(w)[e]; }
    __device__ static void unpack(const Raw &x, float (&v)[1]) { v[0] = x; }
    __device__ static void store(void *w, int64_t e, const float (&v)[1]) { static_cast<float *>(w)[e] = v[0]; }
};
template <>
struct RowRW<uint16_t, 0> {
    static constexpr int NF = 1;
    typedef uint16_t Raw;
    __device__ static Raw load_raw(const void *w, int64_t e) { return static_cast<const uint16_t *>(w)[e]; }
    __device__ static void unpack(const Raw &x, float (&v)[1]) { v[0] = __uint_as_float(static_cast<uint32_t>(x) << 16); }
    __device__ static void store(void *w, int64_t e, const float (&v)[1]) {
        static_cast<uint16_t *>(w)[e] = (uint16_t)f32_to_bf16_bits(v[0]);
    }
};

// gradient vector of NF floats (grad rows are float32; 16-byte loads when VEC16)
template <int NF>
__device__ inline void load_grad(const float *g, float (&v)[NF]) {
    if constexpr (NF == 1) {
        v[0] = g[0];
    } else {
#pragma unroll
        for (int q = 0; q < NF; q += 4) {
            const float4 x = *reinterpret_cast<const float4 *>(g + q);
            v[q] = x.x; v[q + 1] = x.y; v[q + 2] = x.z; v[q + 3] = x.w;
        }
    }
}

// Runs of equal keys are cut at multiples of kSegLen sorted positions, so no group ever sums
// more than kSegLen gradient rows: a table of 3 rows hit by a 65536-sample batch is 20000-long
// runs, which one group would otherwise walk alone.
#ifndef DLRM_BWD_SEGLEN
#define DLRM_BWD_SEGLEN 128      // (256: measured equal at the MLPerf shape)
#endif
constexpr int kSegLen = DLRM_BWD_SEGLEN;

template <int NF>
__device__ inline void load_vec(const float *g, float (&v)[NF]) { load_grad<NF>(g, v); }

template <int NF>
__device__ inline void store_vec(float *g, const float (&v)[NF]) {
    if constexpr (NF == 1) {
        g[0] = v[0];
    } else {
#pragma unroll
        for (int q = 0; q < NF; q += 4) *reinterpret_cast<float4 *>(g + q) = make_float4(v[q], v[q + 1], v[q + 2], v[q + 3]);
    }
}

// optimizer step on row k given its current values w and its summed gradient acc
template <typename ROW, int VEC16, int G, int NC>
__device__ inline void apply_loaded(const BwdParams &p, int64_t k, float (&acc)[NC][RowRW<ROW, VEC16>::NF],
                                    const typename RowRW<ROW, VEC16>::Raw (&wraw)[NC], const int (&col)[NC],
                                    const bool (&col_ok)[NC], int lig) {
    using RW = RowRW<ROW, VEC16>;
    constexpr int NF = RW::NF;
    float w[NC][NF];   // rows are converted only now: their loads were issued together, earlier
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (col_ok[c]) RW::unpack(wraw[c], w[c]);
    const int64_t e0 = k * p.D;
    float mult = p.lr;
    if (p.adagrad) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (col_ok[c]) {
#pragma unroll
                for (int q = 0; q < NF; q++) ss += acc[c][q] * acc[c][q];
            }
#pragma unroll
        for (int m = 1; m < G; m <<= 1) ss += __shfl_xor(ss, m, kWave);
        const float st = p.state[k] + ss / (float)p.D;
        if (lig == 0) p.state[k] = st;
        mult = p.lr / (sqrtf(st) + p.eps);
    }
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (col_ok[c]) {
#pragma unroll
            for (int q = 0; q < NF; q++) w[c][q] -= mult * acc[c][q];
            RW::store(p.w, e0 + col[c], w[c]);
        }
}

// (each touched row is read once and written once)
template <typename ROW, int VEC16, int G, int NC>
__device__ inline void apply_row_update(const BwdParams &p, int64_t k, float (&acc)[NC][RowRW<ROW, VEC16>::NF],
                                        const int (&col)[NC], const bool (&col_ok)[NC], int lig) {
    using RW = RowRW<ROW, VEC16>;
    typename RW::Raw w[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (col_ok[c]) w[c] = RW::load_raw(p.w, k * p.D + col[c]);
    apply_loaded<ROW, VEC16, G, NC>(p, k, acc, w, col, col_ok, lig);
}

// gradient row pointer and scale of an occurrence in bag g
template <typename IDX>
__device__ inline const float *occurrence(const BwdParams &p, uint32_t g, float &sc) {
    const IDX *off = static_cast<const IDX *>(p.off);
    const uint32_t t = g / (uint32_t)p.B;
    const uint32_t b = g - t * (uint32_t)p.B;
    sc = 1.0f;
    if (p.mean) {
        const int64_t s0 = (int64_t)off[g];
        const int64_t e0 = ((int64_t)g + 1 < p.n_bags || p.include_last) ? (int64_t)off[g + 1] : p.n_idx;
        sc = 1.0f / (float)(e0 - s0);
    }
    return p.grad + (int64_t)b * p.grad_ld + (int64_t)t * p.D;
}

// Segmented reduction by row over the sorted occurrences, fused with the optimizer step.
//
// A workgroup owns BLOCK_POS consecutive sorted positions (one or two stretches of kSegLen), each
// G-lane group PPG consecutive positions of it, walked kChunk at a time: the chunk's gradient rows --
// and the table rows of the runs that end in it -- are fetched together (up to 2*kChunk independent
// 16-byte loads per lane), then added up in position order.
//  * A run that lies inside one group is summed and applied there: every touched row is read once
//    and written once.  On large tables almost every run is a single position.
//  * Pieces of runs that cross group boundaries meet in LDS: per group a "head" piece (the run came in
//    from the previous group; flag 1 = it ends inside this group, 2 = it runs through the whole group)
//    and a "tail" piece (a run that starts in this group and leaves it).  The group in which a run
//    ends adds its pieces in position order and applies the update.
//  * Pieces of runs that cross a kSegLen boundary go to `partial` (2 slots per stretch: [0] the piece
//    that starts on the boundary, [1] the piece that starts inside and runs to the end);
//    combine_runs_kernel adds them up.  A table of 3 rows hit by a 65536-sample batch is 20000-long
//    runs: every group of every workgroup sums its share.
// DLRM_DUP_LAST_WINS: only the last occurrence of a row counts, so every position is its own run and
// only run ends are applied.
#ifndef DLRM_BWD_CHUNK
#define DLRM_BWD_CHUNK 4         // (8: 110 registers, four waves per SIMD instead of six: measured 15% slower)
#endif
constexpr int kChunk = DLRM_BWD_CHUNK;

template <typename IDX, typename KEY, typename ROW, int VEC16, int G, int NC>
__global__ __launch_bounds__(kBwdThreads) void sparse_update_kernel(BwdParams p, const KEY *__restrict__ keys,
                                                                     const int32_t *__restrict__ vals,
                                                                     float *__restrict__ partial, int dpad) {
    using RW = RowRW<ROW, VEC16>;
    constexpr int NF = RW::NF;
    constexpr int GROUPS = kBwdThreads / G;
    constexpr int PPG = (kSegLen / GROUPS) > kChunk ? (kSegLen / GROUPS) : kChunk;   // positions per group
    constexpr int GPS = kSegLen / PPG;                                                // groups per stretch
    constexpr int BLOCK_POS = GROUPS * PPG;
    constexpr int DP = NC * G * NF;                                                   // floats per piece
    static_assert(GROUPS % GPS == 0 && PPG % kChunk == 0 && BLOCK_POS % kSegLen == 0, "tiling");
    static_assert(GROUPS * 2 * DP * 4 <= 48 * 1024, "pieces must fit LDS");

    __shared__ KEY s_key[BLOCK_POS + 2];         // keys of positions b0-1 .. b0+BLOCK_POS
    __shared__ int32_t s_val[BLOCK_POS];
    __shared__ int s_head[GROUPS], s_tail[GROUPS];
    __shared__ __attribute__((aligned(16))) float s_piece[GROUPS][2][DP];

    const int tid = threadIdx.x, grp = tid / G, lig = tid % G;
    const int64_t b0 = (int64_t)blockIdx.x * BLOCK_POS;
    const KEY none = ~(KEY)0;
    for (int i = tid; i < BLOCK_POS + 2; i += kBwdThreads) {
        const int64_t q = b0 - 1 + i;
        s_key[i] = (q >= 0 && q < p.n_idx) ? keys[q] : none;
    }
    for (int i = tid; i < BLOCK_POS; i += kBwdThreads) s_val[i] = b0 + i < p.n_idx ? vals[b0 + i] : 0;
    if (tid < GROUPS) { s_head[tid] = 0; s_tail[tid] = 0; }
    __syncthreads();

    auto K = [&](int pos) { return s_key[pos + 1]; };                       // key of block-local position
    auto valid = [&](KEY k) { return (int64_t)k < p.total_rows && k != none; };   // bad indices carry total_rows

    int col[NC];
    bool col_ok[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        col[c] = (c * G + lig) * NF;
        col_ok[c] = col[c] < p.D;
    }
    auto zero = [&](float (&a)[NC][NF]) {
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int q = 0; q < NF; q++) a[c][q] = 0.f;
    };
    auto put_piece = [&](int g, int slot, const float (&a)[NC][NF]) {
#pragma unroll
        for (int c = 0; c < NC; c++) store_vec<NF>(&s_piece[g][slot][(c * G + lig) * NF], a[c]);
    };
    auto add_piece = [&](int g, int slot, float (&a)[NC][NF]) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float v[NF];
            load_vec<NF>(&s_piece[g][slot][(c * G + lig) * NF], v);
#pragma unroll
            for (int q = 0; q < NF; q++) a[c][q] += v[q];
        }
    };

    const int gp0 = grp * PPG;
    const bool accumulate = !p.last_wins;
    // the run at the group's first position came in from the previous group
    bool ext = accumulate && valid(K(gp0)) && K(gp0 - 1) == K(gp0);
    float acc[NC][NF];
    zero(acc);

    for (int c0 = 0; c0 < PPG; c0 += kChunk) {
        const int q0 = gp0 + c0;
        bool v[kChunk], endr[kChunk], need_w[kChunk];
        bool any = false, any_w = false, e = ext;
        int fu = 0, fw = -1;
#pragma unroll
        for (int u = kChunk - 1; u >= 0; u--) {
            v[u] = valid(K(q0 + u));
            endr[u] = K(q0 + u + 1) != K(q0 + u);
            if (v[u]) fu = u;
            any |= v[u];
        }
        if (!any) continue;
#pragma unroll
        for (int u = 0; u < kChunk; u++) {
            need_w[u] = v[u] && endr[u] && !e;      // a run ends here and this group holds all of it
            if (v[u] && endr[u]) e = false;
            if (need_w[u] && fw < 0) fw = u;
            any_w |= need_w[u];
        }
        // slots without work re-read the first slot that has some: the loads need no branches
        const float *gp[kChunk];
        float sc[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; u++) gp[u] = occurrence<IDX>(p, (uint32_t)s_val[q0 + (v[u] ? u : fu)], sc[u]);
        float g[kChunk][NC][NF];
        typename RW::Raw w[kChunk][NC];
        if (any_w) {
            int64_t row[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; u++) row[u] = (int64_t)K(q0 + (need_w[u] ? u : fw));
#pragma unroll
            for (int u = 0; u < kChunk; u++)
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if (col_ok[c]) {
                        load_grad<NF>(gp[u] + col[c], g[u][c]);
                        w[u][c] = RW::load_raw(p.w, row[u] * p.D + col[c]);
                    }
        } else {
#pragma unroll
            for (int u = 0; u < kChunk; u++)
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if (col_ok[c]) load_grad<NF>(gp[u] + col[c], g[u][c]);
        }
#pragma unroll
        for (int u = 0; u < kChunk; u++) {
            if (!v[u]) continue;
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < NF; q++) {
#ifdef DLRM_PK_PROBE
                    const float before = acc[c][q];
#endif
                    acc[c][q] = accumulate ? acc[c][q] + g[u][c][q] * sc[u] : g[u][c][q] * sc[u];
#ifdef DLRM_PK_PROBE
                    if (accumulate && col_ok[c]) {
                        float ref;
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ref) : "v"(g[u][c][q]), "v"(sc[u]), "v"(before));
                        if (__float_as_uint(ref) != __float_as_uint(acc[c][q])) {
                            const unsigned slot = atomicAdd(&g_pk_probe_count, 1u);
                            if (slot < (unsigned)kProbeRecords) {
                                const float again = __builtin_nontemporal_load(gp[u] + col[c] + q);
                                unsigned *r = g_pk_probe[slot];
                                r[0] = blockIdx.x; r[1] = threadIdx.x; r[2] = (unsigned)(c0 + u); r[3] = (unsigned)q;
                                r[4] = __float_as_uint(g[u][c][q]); r[5] = __float_as_uint(again);
                                r[6] = __float_as_uint(sc[u]); r[7] = __float_as_uint(before);
                                r[8] = __float_as_uint(acc[c][q]); r[9] = __float_as_uint(ref);
                                r[10] = (unsigned)any_w; r[11] = (unsigned)ext;
                            }
                        }
                    }
#endif
                }
            if (!endr[u]) continue;
            if (ext) {                       // the run started in an earlier group: hand the piece over
                put_piece(grp, 0, acc);
                if (lig == 0) s_head[grp] = 1;
                ext = false;
            } else {
                apply_loaded<ROW, VEC16, G, NC>(p, (int64_t)K(q0 + u), acc, w[u], col, col_ok, lig);
            }
            zero(acc);
        }
    }
    if (accumulate && valid(K(gp0 + PPG - 1)) && K(gp0 + PPG) == K(gp0 + PPG - 1)) {   // the last run leaves the group
        put_piece(grp, ext ? 0 : 1, acc);
        if (lig == 0) {
            if (ext) s_head[grp] = 2;
            else s_tail[grp] = 1;
        }
    }
    if (!accumulate) return;
    __syncthreads();

    const int seg = grp / GPS, seg_first = seg * GPS;
    const int64_t seg_global = b0 / kSegLen + seg;
    // sum of the pieces of the run that reaches into group `last` (whose piece there is in the head
    // slot); returns the group in which the run started, or seg_first - 1 if it came in from before
    // this stretch
    auto gather = [&](int last, float (&a)[NC][NF]) {
        int j = last - 1;
        while (j >= seg_first && s_head[j] == 2) j--;
        zero(a);
        if (j >= seg_first) add_piece(j, 1, a);
        for (int q = j + 1; q <= last; q++) add_piece(q, 0, a);
        return j;
    };
    if (s_head[grp] == 1) {                  // a run that came in from earlier groups ends here
        const KEY k = K(gp0);
        float a[NC][NF];
        const bool from_inside = gather(grp, a) >= seg_first;
        if (from_inside || K(seg_first * PPG - 1) != k) {
            apply_row_update<ROW, VEC16, G, NC>(p, (int64_t)k, a, col, col_ok, lig);
        } else {                             // it started before this stretch
            float *dst = partial + (2 * seg_global + 0) * (int64_t)dpad;
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (col_ok[c]) store_vec<NF>(dst + col[c], a[c]);
        }
    }
    if (grp == seg_first + GPS - 1 && (s_head[grp] == 2 || s_tail[grp] == 1)) {   // a run leaves the stretch
        float a[NC][NF];
        bool on_boundary = false;            // does the piece start at the first position of the stretch?
        if (s_tail[grp] == 1) {
            zero(a);
            add_piece(grp, 1, a);
        } else {
            const int j = gather(grp, a);
            // from before the stretch, or the first group is one single run that leaves it
            on_boundary = j < seg_first || (j == seg_first && K(seg_first * PPG) == K(seg_first * PPG + PPG - 1));
        }
        float *dst = partial + (2 * seg_global + (on_boundary ? 0 : 1)) * (int64_t)dpad;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (col_ok[c]) store_vec<NF>(dst + col[c], a[c]);
    }
}

// One workgroup per stretch of kSegLen positions: if a run starts inside this stretch and goes on beyond it, add up
// its pieces and apply the update.  The run's later stretches are dealt round-robin, kU at a time, to the groups
// of the workgroup (a 3-row table hit by a 65536-sample batch is a run of 170 stretches: one group walking them
// alone is a chain of 22 dependent round trips to memory); every group adds its share in position order and
// group 0 adds the groups' sums in group order, so the result does not depend on timing.
template <typename KEY, typename ROW, int VEC16, int G, int NC>
__global__ __launch_bounds__(kBwdThreads) void combine_runs_kernel(BwdParams p, const KEY *__restrict__ keys,
                                                                    const float *__restrict__ partial, int dpad) {
    using RW = RowRW<ROW, VEC16>;
    constexpr int NF = RW::NF;
    constexpr int GROUPS = kBwdThreads / G;
    constexpr int DP = NC * G * NF;
    __shared__ __attribute__((aligned(16))) float s_sum[GROUPS][DP];
    const int grp = threadIdx.x / G, lig = threadIdx.x % G;
    const int64_t c0 = blockIdx.x;
    const int64_t s0 = c0 * kSegLen;
    // (every exit below depends on the stretch only: the whole workgroup takes it together)
    if (s0 >= p.n_idx) return;
    const int64_t e = min(p.n_idx, s0 + kSegLen) - 1;
    if (e + 1 >= p.n_idx) return;
    const KEY k = keys[e];
    if ((int64_t)k >= p.total_rows || keys[e + 1] != k) return;   // nothing continues past this stretch
    const bool whole = keys[s0] == k;                              // the run covers the stretch from its first position
    if (whole && s0 > 0 && keys[s0 - 1] == k) return;              // the run started earlier: not ours

    int col[NC];
    bool col_ok[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        col[c] = (c * G + lig) * NF;
        col_ok[c] = col[c] < p.D;
    }
    float acc[NC][NF];
    const float *src = partial + (2 * c0 + (whole ? 0 : 1)) * (int64_t)dpad;
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int q = 0; q < NF; q++) acc[c][q] = 0.f;
        if (grp == 0 && col_ok[c]) load_vec<NF>(src + col[c], acc[c]);
    }
    constexpr int kU = 8;
    const int64_t n_seg = (p.n_idx + kSegLen - 1) / kSegLen;
    for (int64_t cc = c0 + 1 + (int64_t)grp * kU; cc < n_seg && keys[cc * kSegLen] == k; cc += (int64_t)GROUPS * kU) {
        const float *pp[kU];
        float m[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int64_t cu = cc + u;
            const bool ok = cu < n_seg && keys[(cu < n_seg ? cu : cc) * kSegLen] == k;
            pp[u] = partial + 2 * (ok ? cu : cc) * (int64_t)dpad;
            m[u] = ok ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (col_ok[c]) {
                float v[kU][NF];
#pragma unroll
                for (int u = 0; u < kU; u++) load_vec<NF>(pp[u] + col[c], v[u]);
#pragma unroll
                for (int u = 0; u < kU; u++)
#pragma unroll
                    for (int q = 0; q < NF; q++) acc[c][q] += v[u][q] * m[u];
            }
    }
    if (grp > 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) store_vec<NF>(&s_sum[grp][(c * G + lig) * NF], acc[c]);
    }
    __syncthreads();
    if (grp > 0) return;
    for (int g = 1; g < GROUPS; g++) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float v[NF];
            load_vec<NF>(&s_sum[g][(c * G + lig) * NF], v);
#pragma unroll
            for (int q = 0; q < NF; q++) acc[c][q] += v[q];
        }
    }
    apply_row_update<ROW, VEC16, G, NC>(p, (int64_t)k, acc, col, col_ok, lig);
}

template <typename IDX, typename KEY, typename ROW, int VEC16, int G, int NC>
static int launch_update(const BwdParams &p, const KEY *keys, const int32_t *vals, float *partial, int dpad,
                         hipStream_t s) {
    constexpr int kGroups = kBwdThreads / G;
    constexpr int kPpg = (kSegLen / kGroups) > kChunk ? (kSegLen / kGroups) : kChunk;
    const int64_t blocks = ceil_div(p.n_idx, (int64_t)kGroups * kPpg);
    hipLaunchKernelGGL((sparse_update_kernel<IDX, KEY, ROW, VEC16, G, NC>), dim3((unsigned)blocks),
                       dim3(kBwdThreads), 0, s, p, keys, vals, partial, dpad);
    DLRM_LAUNCH_CHECK();
    if (!p.last_wins && p.n_idx > kSegLen) {
        const int64_t segs = ceil_div(p.n_idx, kSegLen);
        hipLaunchKernelGGL((combine_runs_kernel<KEY, ROW, VEC16, G, NC>), dim3((unsigned)segs), dim3(kBwdThreads), 0, s, p,
                           keys, partial, dpad);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

template <typename IDX, typename KEY, typename ROW>
static int dispatch_update(const BwdParams &p, const KEY *keys, const int32_t *vals, float *partial, int dpad,
                           bool vec16, int lanes, hipStream_t s) {
    if (!vec16) {
        if (lanes <= 64) return launch_update<IDX, KEY, ROW, 0, 64, 1>(p, keys, vals, partial, dpad, s);
        if (lanes <= 128) return launch_update<IDX, KEY, ROW, 0, 64, 2>(p, keys, vals, partial, dpad, s);
        return launch_update<IDX, KEY, ROW, 0, 64, 4>(p, keys, vals, partial, dpad, s);
    }
    if (lanes <= 4) return launch_update<IDX, KEY, ROW, 1, 4, 1>(p, keys, vals, partial, dpad, s);
    if (lanes <= 8) return launch_update<IDX, KEY, ROW, 1, 8, 1>(p, keys, vals, partial, dpad, s);
    if (lanes <= 16) return launch_update<IDX, KEY, ROW, 1, 16, 1>(p, keys, vals, partial, dpad, s);
    if (lanes <= 32) return launch_update<IDX, KEY, ROW, 1, 32, 1>(p, keys, vals, partial, dpad, s);
    if (lanes <= 64) return launch_update<IDX, KEY, ROW, 1, 64, 1>(p, keys, vals, partial, dpad, s);
    if (lanes <= 128) return launch_update<IDX, KEY, ROW, 1, 64, 2>(p, keys, vals, partial, dpad, s);
    if constexpr (sizeof(ROW) == 4) return launch_update<IDX, KEY, ROW, 1, 64, 4>(p, keys, vals, partial, dpad, s);
    return set_error(DLRM_ERR_INVALID, "dim=%d is too wide for the sparse update (max 1024)", p.D);
}

// phases of the backward: 1 = expand + sort (needs the ids only), 2 = reduce + update (needs the
// gradients and a workspace that phase 1 filled for the same ids), 3 = both
template <typename IDX, typename KEY>
static int run_backward(const dlrm_embbag *e, BwdParams &p, void *ws, size_t ws_bytes, bool vec16, int lanes,
                        int phases, hipStream_t s) {
    WsLayout<KEY> L;
    const int bits = key_bits(e->total_rows);
    if (int rc = ws_layout<KEY>(e->n_indices, bits, e->dim, ws, &L)) return rc;
    DLRM_REQUIRE(ws_bytes >= L.total, "workspace too small: %zu < %zu bytes", ws_bytes, L.total);

    if (phases & 1) {
        const double avg_len = (double)e->n_indices / (double)p.n_bags;
        int lpb = 1;
        while (lpb < 64 && lpb * 2 <= avg_len) lpb <<= 1;
        const int64_t blocks = ceil_div(p.n_bags * lpb, kBwdThreads);
        static const bool use_rocprim = [] { const char *v = getenv("DLRM_SORT"); return v && strcmp(v, "rocprim") == 0; }();
        // the sorted pairs must end in (keys_out, vals_out): the library's own sort alternates between the two pairs of
        // buffers once per pass, so with an even number of passes the pairs start out in the "out" buffers
        const bool start_in_out = !use_rocprim && pair_sort_passes(bits) % 2 == 0;
        KEY *k0 = start_in_out ? L.keys_out : L.keys_in, *k1 = start_in_out ? L.keys_in : L.keys_out;
        int32_t *v0 = start_in_out ? L.vals_out : L.vals_in, *v1 = start_in_out ? L.vals_in : L.vals_out;
        hipLaunchKernelGGL((expand_pairs_kernel<IDX, KEY>), dim3((unsigned)blocks + kTailBlocks), dim3(kBwdThreads), 0, s,
                           p, lpb, (unsigned)blocks, k0, v0);
        DLRM_LAUNCH_CHECK();
        if (use_rocprim) {
            size_t tmp = L.sort_tmp_bytes;
            DLRM_HIP(rocprim::radix_sort_pairs(L.sort_tmp, tmp, L.keys_in, L.keys_out, L.vals_in, L.vals_out,
                                               (size_t)e->n_indices, 0u, (unsigned)bits, s, false));
        } else {
            if (int rc = pair_sort(k0, k1, v0, v1, e->n_indices, bits, L.sort_tmp, L.sort_tmp_bytes, s)) return rc;
        }
    }
    if (!(phases & 2)) return DLRM_OK;
    if (e->row_dtype == DLRM_BF16)
        return dispatch_update<IDX, KEY, uint16_t>(p, L.keys_out, L.vals_out, L.partial, pad_dim(e->dim), vec16, lanes, s);
    return dispatch_update<IDX, KEY, float>(p, L.keys_out, L.vals_out, L.partial, pad_dim(e->dim), vec16, lanes, s);
}

static int backward_phases(const dlrm_embbag *e, const float *d_grad, int64_t grad_ld, const dlrm_sparse_opt *opt,
                           void *d_workspace, size_t workspace_bytes, int phases, dlrm_stream_t stream) {
    if (int rc = validate_embbag(e)) return rc;
    const bool apply = (phases & 2) != 0;
    if (apply) {
        DLRM_REQUIRE(opt != nullptr, "opt is NULL");
        DLRM_REQUIRE(opt->optimizer == DLRM_OPT_SGD || opt->optimizer == DLRM_OPT_ADAGRAD_ROWWISE, "optimizer %d",
                     opt->optimizer);
        DLRM_REQUIRE(opt->dup_mode == DLRM_DUP_ACCUMULATE || opt->dup_mode == DLRM_DUP_LAST_WINS, "dup_mode %d",
                     opt->dup_mode);
        DLRM_REQUIRE(opt->optimizer != DLRM_OPT_ADAGRAD_ROWWISE || opt->d_opt_state != nullptr,
                     "row-wise Adagrad needs d_opt_state");
    }
    const int64_t n_bags = (int64_t)e->num_tables * e->batch;
    if (n_bags == 0 || e->n_indices == 0) return DLRM_OK;
    DLRM_REQUIRE(d_workspace != nullptr && (!apply || d_grad != nullptr), "d_grad / d_workspace is NULL");
    DLRM_REQUIRE(!apply || grad_ld >= (int64_t)e->num_tables * e->dim, "grad_ld too small");
    DLRM_REQUIRE(n_bags < (1ll << 31) && e->n_indices < (1ll << 40), "batch too large for 32-bit bag ids");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(d_workspace) % 256 == 0, "workspace must be 256-byte aligned");

    BwdParams p;
    p.w = e->d_weight; p.tro = e->d_table_row_offsets; p.total_rows = e->total_rows;
    p.T = e->num_tables; p.D = e->dim; p.idx = e->d_indices; p.off = e->d_offsets;
    p.n_idx = e->n_indices; p.B = e->batch; p.n_bags = n_bags;
    p.include_last = e->include_last_offset; p.mean = e->pooling == DLRM_POOL_MEAN;
    p.grad = d_grad; p.grad_ld = grad_ld; p.err = e->d_error_flag; p.shard = e->d_shard_row_begin;
    p.last_wins = apply && opt->dup_mode == DLRM_DUP_LAST_WINS;
    p.adagrad = apply && opt->optimizer == DLRM_OPT_ADAGRAD_ROWWISE;
    p.lr = apply ? opt->lr : 0.f; p.eps = apply ? opt->eps : 0.f; p.state = apply ? opt->d_opt_state : nullptr;

    const int esz = e->row_dtype == DLRM_F32 ? 4 : 2;
    const int nf = 16 / esz;
    const bool vec16 = (e->dim % nf == 0) && (reinterpret_cast<uintptr_t>(e->d_weight) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(d_grad) % 16 == 0) && (grad_ld % 4 == 0) && (e->dim % 4 == 0);
    const int lanes = vec16 ? e->dim / nf : e->dim;
    DLRM_REQUIRE(lanes <= 256 && e->dim <= 1024, "dim=%d is too wide for this kernel (max 1024)", e->dim);

    hipStream_t s = as_stream(stream);
    const bool k64 = key_bits(e->total_rows) > 32;
    const bool i64 = e->idx_dtype == DLRM_I64;
    if (k64) {
        if (i64) return run_backward<int64_t, uint64_t>(e, p, d_workspace, workspace_bytes, vec16, lanes, phases, s);
        return run_backward<int32_t, uint64_t>(e, p, d_workspace, workspace_bytes, vec16, lanes, phases, s);
    }
    if (i64) return run_backward<int64_t, uint32_t>(e, p, d_workspace, workspace_bytes, vec16, lanes, phases, s);
    return run_backward<int32_t, uint32_t>(e, p, d_workspace, workspace_bytes, vec16, lanes, phases, s);
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_embbag_backward_workspace_bytes(const dlrm_embbag *e, size_t *bytes) {
    if (int rc = validate_embbag(e)) return rc;
    DLRM_REQUIRE(bytes != nullptr, "bytes is NULL");
    const int bits = key_bits(e->total_rows);
    const int64_t n = e->n_indices > 0 ? e->n_indices : 1;
    if (bits <= 32) {
        WsLayout<uint32_t> L;
        if (int rc = ws_layout<uint32_t>(n, bits, e->dim, nullptr, &L)) return rc;
        *bytes = L.total;
    } else {
        WsLayout<uint64_t> L;
        if (int rc = ws_layout<uint64_t>(n, bits, e->dim, nullptr, &L)) return rc;
        *bytes = L.total;
    }
    return DLRM_OK;
}

extern "C" int dlrm_embbag_backward_update(const dlrm_embbag *e, const float *d_grad, int64_t grad_ld,
                                           const dlrm_sparse_opt *opt, void *d_workspace, size_t workspace_bytes,
                                           dlrm_stream_t stream) {
    return backward_phases(e, d_grad, grad_ld, opt, d_workspace, workspace_bytes, 3, stream);
}

extern "C" int dlrm_embbag_backward_prepare(const dlrm_embbag *e, void *d_workspace, size_t workspace_bytes,
                                            dlrm_stream_t stream) {
    return backward_phases(e, nullptr, 0, nullptr, d_workspace, workspace_bytes, 1, stream);
}

extern "C" int dlrm_embbag_backward_apply(const dlrm_embbag *e, const float *d_grad, int64_t grad_ld,
                                          const dlrm_sparse_opt *opt, void *d_workspace, size_t workspace_bytes,
                                          dlrm_stream_t stream) {
    return backward_phases(e, d_grad, grad_ld, opt, d_workspace, workspace_bytes, 2, stream);
}

#ifdef DLRM_PK_PROBE
// diagnostic build only: number of disagreements since the last call and the first records (12 words each)
extern "C" int dlrm_pk_probe_read(unsigned *h_count, unsigned *h_records, int max_records) {
    DLRM_HIP(hipDeviceSynchronize());
    DLRM_HIP(hipMemcpyFromSymbol(h_count, HIP_SYMBOL(dlrm::g_pk_probe_count), sizeof(unsigned)));
    const int n = max_records < dlrm::kProbeRecords ? max_records : dlrm::kProbeRecords;
    DLRM_HIP(hipMemcpyFromSymbol(h_records, HIP_SYMBOL(dlrm::g_pk_probe), sizeof(unsigned) * 12 * n));
    const unsigned zero = 0;
    DLRM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dlrm::g_pk_probe_count), &zero, sizeof(unsigned)));
    return DLRM_OK;
}
#endif
