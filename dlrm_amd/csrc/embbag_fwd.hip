// EmbeddingBag forward (gather + sum/mean) for gfx950.
//
// HBM-bound: per bag it reads len*dim row elements and writes dim floats, so
// the design goal is many independent 16-byte row loads in flight per lane
// and full-line coalescing of every row read and every pooled-vector write.
//
//  * A row is covered by G lanes x 16 bytes (G = 4..64, a power of two), so a
//    wave64 reads 64/G rows per load instruction; rows wider than 1 KiB take
//    NC column chunks per lane.
//  * "group" kernel: each G-lane group owns a run of consecutive bags and
//    walks their (contiguous) index range four rows at a time -- four
//    independent row loads in flight per lane; short bags (L = 1 in the
//    Criteo shapes) are batched `bags_per_group` to a group so the same depth
//    is reached.
//  * "wave" kernel: one wave per bag for long bags when there are too few
//    bags to fill the chip: the 64/G groups of the wave take interleaved rows
//    and the partial sums are combined with cross-lane shuffles.
//  * The CSR offsets of a workgroup's bags are staged through LDS with one
//    coalesced read.
//  * Pooled vectors are written with non-temporal 16-byte stores straight into
//    the caller's concatenated layout (no per-table tensors, no cat).
#include "common.h"

namespace dlrm {

struct FwdParams {
    const void *w;
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
    float *out;
    int64_t out_ld;
    int *err;
    int bpg;  // bags per group (group kernel)
    const int64_t *shard;  // first global row held per table (row-wise shard), or NULL
};

// 16 bytes of a row as NF floats.  `load` only fetches (the raw register image): conversions happen
// in `add`, so that all of a step's loads are issued before anything waits for one of them.
template <typename ROW, int VEC16>
struct RowLoad;

template <>
struct RowLoad<float, 1> {
    static constexpr int NF = 4;
    typedef float4 Raw;
    __device__ static Raw load(const void *w, int64_t elem) {
        return *reinterpret_cast<const float4 *>(static_cast<const float *>(w) + elem);
    }
    __device__ static void add(float (&a)[4], const Raw &x) { a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w; }
};
template <>
struct RowLoad<uint16_t, 1> {
    static constexpr int NF = 8;
    typedef uint4 Raw;
    __device__ static Raw load(const void *w, int64_t elem) {
        return *reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(w) + elem);
    }
    __device__ static void add(float (&a)[8], const Raw &x) {
        a[0] += __uint_as_float(x.x << 16); a[1] += __uint_as_float(x.x & 0xffff0000u);
        a[2] += __uint_as_float(x.y << 16); a[3] += __uint_as_float(x.y & 0xffff0000u);
        a[4] += __uint_as_float(x.z << 16); a[5] += __uint_as_float(x.z & 0xffff0000u);
        a[6] += __uint_as_float(x.w << 16); a[7] += __uint_as_float(x.w & 0xffff0000u);
    }
};
// element-per-lane fallback for dims / alignments the 16-byte path cannot take
template <>
struct RowLoad<float, 0> {
    static constexpr int NF = 1;
    typedef float Raw;
    __device__ static Raw load(const void *w, int64_t elem) { return static_cast<const float *>(w)[elem]; }
    __device__ static void add(float (&a)[1], const Raw &x) { a[0] += x; }
};
template <>
struct RowLoad<uint16_t, 0> {
    static constexpr int NF = 1;
    typedef uint16_t Raw;
    __device__ static Raw load(const void *w, int64_t elem) { return static_cast<const uint16_t *>(w)[elem]; }
    __device__ static void add(float (&a)[1], const Raw &x) { a[0] += __uint_as_float(static_cast<uint32_t>(x) << 16); }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NF>
__device__ inline void store_pooled(float *dst, const float (&a)[NF], float sc) {
    if constexpr (NF == 1) {
        __builtin_nontemporal_store(a[0] * sc, dst);
    } else {
#pragma unroll
        for (int q = 0; q < NF; q += 4) {
            f32x4 o = {a[q] * sc, a[q + 1] * sc, a[q + 2] * sc, a[q + 3] * sc};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dst + q));
        }
    }
}

constexpr int kFwdThreads = 256;
constexpr int kMaxBpg = 8;
constexpr int kUnroll = 4;

// Each G-lane group pools `bpg` consecutive bags.
template <typename IDX, typename ROW, int VEC16, int G, int NC>
__global__ __launch_bounds__(kFwdThreads) void embbag_fwd_group_kernel(FwdParams p) {
    using RL = RowLoad<ROW, VEC16>;
    constexpr int NF = RL::NF;
    constexpr int GROUPS = kFwdThreads / G;
    __shared__ int64_t s_off[GROUPS * kMaxBpg + 1];

    const int tid = threadIdx.x;
    const int grp = tid / G, lig = tid % G;
    const int64_t bags_per_block = (int64_t)GROUPS * p.bpg;
    const int64_t g_block = (int64_t)blockIdx.x * bags_per_block;
    const int64_t n_here = min(bags_per_block, p.n_bags - g_block);
    const IDX *off = static_cast<const IDX *>(p.off);
    const IDX *idx = static_cast<const IDX *>(p.idx);

    for (int i = tid; i <= n_here; i += kFwdThreads) {
        const int64_t g = g_block + i;
        s_off[i] = (g < p.n_bags || p.include_last) ? (int64_t)off[g] : p.n_idx;
    }
    __syncthreads();

    const int64_t gl0 = (int64_t)grp * p.bpg;
    if (gl0 >= n_here) return;
    const int nb = (int)min((int64_t)p.bpg, n_here - gl0);
    const int64_t g0 = g_block + gl0;
    const int64_t *so = s_off + gl0;

    // (table, batch row) of the bag being flushed, and table of the bag being loaded
    int t_cur = (int)(g0 / p.B);
    int64_t b_cur = g0 - (int64_t)t_cur * p.B;
    int t_look = t_cur;
    int64_t look_next_table = p.B - b_cur;  // local bag index at which the next table starts
    int64_t tbase = p.tro ? p.tro[t_look] : 0;
    int64_t trows = p.tro ? p.tro[t_look + 1] - tbase : p.total_rows;
    int64_t sbase = p.shard ? p.shard[t_look] : 0;

    bool col_ok[NC];
    int col[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        col[c] = (c * G + lig) * NF;
        col_ok[c] = col[c] < p.D;
    }

    float acc[NC][NF];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int q = 0; q < NF; q++) acc[c][q] = 0.f;

    int cur = 0, look = 0;
    bool bad = false;

    auto flush = [&]() {
        const int64_t len = so[cur + 1] - so[cur];
        const float sc = p.mean ? (len > 0 ? 1.0f / (float)len : 0.f) : 1.0f;
        float *o = p.out + b_cur * p.out_ld + (int64_t)t_cur * p.D;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (col_ok[c]) store_pooled<NF>(o + col[c], acc[c], sc);
#pragma unroll
            for (int q = 0; q < NF; q++) acc[c][q] = 0.f;
        }
        cur++;
        if (++b_cur == p.B) { b_cur = 0; t_cur++; }
    };

    const int64_t start = so[0], end = so[nb];
    for (int64_t pos = start; pos < end; pos += kUnroll) {
        typename RL::Raw v[kUnroll][NC];
        int bagj[kUnroll];
        bool ok[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; j++) {
            const int64_t pj = pos + j;
            ok[j] = pj < end;
            bagj[j] = look;
            if (ok[j]) {
                while (pj >= so[look + 1]) look++;
                while (look >= look_next_table) {
                    t_look++;
                    look_next_table += p.B;
                    if (p.tro) {
                        tbase = p.tro[t_look];
                        trows = p.tro[t_look + 1] - tbase;
                    }
                    if (p.shard) sbase = p.shard[t_look];
                }
                bagj[j] = look;
                const int64_t r = (int64_t)idx[pj] - sbase;
                if (r < 0 || r >= trows) {
                    ok[j] = false;
                    bad = bad || p.shard == nullptr;
                } else {
                    const int64_t e0 = (tbase + r) * p.D;
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (col_ok[c]) v[j][c] = RL::load(p.w, e0 + col[c]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; j++) {
            if (pos + j < end) {
                while (cur < bagj[j]) flush();
                if (ok[j]) {
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (col_ok[c]) RL::add(acc[c], v[j][c]);
                }
            }
        }
    }
    while (cur < nb) flush();
    if (bad && p.err && lig == 0) *p.err = 1;
}

// One wave per bag; the 64/G groups of the wave take rows start+grp, +NG, ...
template <typename IDX, typename ROW, int VEC16, int G, int NC>
__global__ __launch_bounds__(kFwdThreads) void embbag_fwd_wave_kernel(FwdParams p) {
    using RL = RowLoad<ROW, VEC16>;
    constexpr int NF = RL::NF;
    constexpr int NG = kWave / G;  // groups per wave
    constexpr int WAVES = kFwdThreads / kWave;
    __shared__ int64_t s_off[WAVES + 1];

    const int tid = threadIdx.x;
    const int wave = tid / kWave, lane = tid % kWave;
    const int grp = lane / G, lig = lane % G;
    const int64_t g_block = (int64_t)blockIdx.x * WAVES;
    const int64_t n_here = min((int64_t)WAVES, p.n_bags - g_block);
    const IDX *off = static_cast<const IDX *>(p.off);
    const IDX *idx = static_cast<const IDX *>(p.idx);
    if (tid <= n_here) {
        const int64_t g = g_block + tid;
        s_off[tid] = (g < p.n_bags || p.include_last) ? (int64_t)off[g] : p.n_idx;
    }
    __syncthreads();
    if (wave >= n_here) return;
    const int64_t g = g_block + wave;
    const int t = (int)(g / p.B);
    const int64_t b = g - (int64_t)t * p.B;
    const int64_t tbase = p.tro ? p.tro[t] : 0;
    const int64_t trows = p.tro ? p.tro[t + 1] - tbase : p.total_rows;
    const int64_t sbase = p.shard ? p.shard[t] : 0;
    const int64_t start = s_off[wave], end = s_off[wave + 1];

    bool col_ok[NC];
    int col[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        col[c] = (c * G + lig) * NF;
        col_ok[c] = col[c] < p.D;
    }
    float acc[NC][NF];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int q = 0; q < NF; q++) acc[c][q] = 0.f;
    bool bad = false;

    for (int64_t pos = start + grp; pos < end; pos += (int64_t)NG * kUnroll) {
        typename RL::Raw v[kUnroll][NC];
        bool ok[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; j++) {
            const int64_t pj = pos + (int64_t)j * NG;
            ok[j] = pj < end;
            if (ok[j]) {
                const int64_t r = (int64_t)idx[pj] - sbase;
                if (r < 0 || r >= trows) {
                    ok[j] = false;
                    bad = bad || p.shard == nullptr;
                } else {
                    const int64_t e0 = (tbase + r) * p.D;
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (col_ok[c]) v[j][c] = RL::load(p.w, e0 + col[c]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; j++)
            if (ok[j]) {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if (col_ok[c]) RL::add(acc[c], v[j][c]);
            }
    }
    // combine the NG partial sums: butterfly over the lane bits above log2(G)
#pragma unroll
    for (int m = G; m < kWave; m <<= 1)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int q = 0; q < NF; q++) acc[c][q] += __shfl_xor(acc[c][q], m, kWave);
    if (grp == 0) {
        const int64_t len = end - start;
        const float sc = p.mean ? (len > 0 ? 1.0f / (float)len : 0.f) : 1.0f;
        float *o = p.out + b * p.out_ld + (int64_t)t * p.D;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (col_ok[c]) store_pooled<NF>(o + col[c], acc[c], sc);
    }
    if (bad && p.err && lig == 0) *p.err = 1;
}

template <typename IDX, typename ROW, int VEC16, int G, int NC>
static int launch_fwd(const FwdParams &p, bool wave_per_bag, hipStream_t s) {
    if (wave_per_bag) {
        const int64_t blocks = ceil_div(p.n_bags, kFwdThreads / kWave);
        hipLaunchKernelGGL((embbag_fwd_wave_kernel<IDX, ROW, VEC16, G, NC>), dim3((unsigned)blocks),
                           dim3(kFwdThreads), 0, s, p);
    } else {
        const int64_t blocks = ceil_div(p.n_bags, (int64_t)(kFwdThreads / G) * p.bpg);
        hipLaunchKernelGGL((embbag_fwd_group_kernel<IDX, ROW, VEC16, G, NC>), dim3((unsigned)blocks),
                           dim3(kFwdThreads), 0, s, p);
    }
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

template <typename IDX, typename ROW>
static int dispatch_fwd(const FwdParams &p, bool vec16, int lanes, bool wave_per_bag, hipStream_t s) {
    if (!vec16) {
        if (lanes <= 64) return launch_fwd<IDX, ROW, 0, 64, 1>(p, wave_per_bag, s);
        if (lanes <= 128) return launch_fwd<IDX, ROW, 0, 64, 2>(p, wave_per_bag, s);
        return launch_fwd<IDX, ROW, 0, 64, 4>(p, wave_per_bag, s);
    }
    if (lanes <= 4) return launch_fwd<IDX, ROW, 1, 4, 1>(p, wave_per_bag, s);
    if (lanes <= 8) return launch_fwd<IDX, ROW, 1, 8, 1>(p, wave_per_bag, s);
    if (lanes <= 16) return launch_fwd<IDX, ROW, 1, 16, 1>(p, wave_per_bag, s);
    if (lanes <= 32) return launch_fwd<IDX, ROW, 1, 32, 1>(p, wave_per_bag, s);
    if (lanes <= 64) return launch_fwd<IDX, ROW, 1, 64, 1>(p, wave_per_bag, s);
    if (lanes <= 128) return launch_fwd<IDX, ROW, 1, 64, 2>(p, wave_per_bag, s);
    return launch_fwd<IDX, ROW, 1, 64, 4>(p, wave_per_bag, s);
}

int validate_embbag(const dlrm_embbag *e) {
    DLRM_REQUIRE(e != nullptr, "embbag descriptor is NULL");
    DLRM_REQUIRE(e->num_tables >= 1 && e->dim >= 1, "num_tables=%d dim=%d must be >= 1", e->num_tables, e->dim);
    DLRM_REQUIRE(e->batch >= 0 && e->n_indices >= 0 && e->total_rows >= 0, "negative size");
    DLRM_REQUIRE(e->row_dtype == DLRM_F32 || e->row_dtype == DLRM_BF16, "row_dtype %d", e->row_dtype);
    DLRM_REQUIRE(e->idx_dtype == DLRM_I32 || e->idx_dtype == DLRM_I64, "idx_dtype %d", e->idx_dtype);
    DLRM_REQUIRE(e->pooling == DLRM_POOL_SUM || e->pooling == DLRM_POOL_MEAN, "pooling %d", e->pooling);
    DLRM_REQUIRE(e->d_table_row_offsets != nullptr || e->d_shard_row_begin == nullptr,
                 "row-wise shards need d_table_row_offsets");
    DLRM_REQUIRE(e->d_weight != nullptr || e->total_rows == 0, "d_weight is NULL");
    DLRM_REQUIRE(e->d_offsets != nullptr || e->batch == 0, "d_offsets is NULL");
    DLRM_REQUIRE(e->d_indices != nullptr || e->n_indices == 0, "d_indices is NULL");
    return DLRM_OK;
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_embbag_forward(const dlrm_embbag *e, float *d_out, int64_t out_ld, dlrm_stream_t stream) {
    if (int rc = validate_embbag(e)) return rc;
    const int64_t n_bags = (int64_t)e->num_tables * e->batch;
    if (n_bags == 0) return DLRM_OK;
    DLRM_REQUIRE(d_out != nullptr, "d_out is NULL");
    DLRM_REQUIRE(out_ld >= (int64_t)e->num_tables * e->dim, "out_ld=%lld < num_tables*dim=%lld",
                 (long long)out_ld, (long long)e->num_tables * e->dim);
    DLRM_REQUIRE(n_bags < (1ll << 40), "too many bags");

    FwdParams p;
    p.w = e->d_weight; p.tro = e->d_table_row_offsets; p.total_rows = e->total_rows;
    p.T = e->num_tables; p.D = e->dim; p.idx = e->d_indices; p.off = e->d_offsets;
    p.n_idx = e->n_indices; p.B = e->batch; p.n_bags = n_bags;
    p.include_last = e->include_last_offset; p.mean = e->pooling == DLRM_POOL_MEAN;
    p.out = d_out; p.out_ld = out_ld; p.err = e->d_error_flag; p.shard = e->d_shard_row_begin;

    const int esz = e->row_dtype == DLRM_F32 ? 4 : 2;
    const int nf = 16 / esz;
    // the 16-byte path needs whole vectors per row and 16-byte aligned rows and outputs
    const bool vec16 = (e->dim % nf == 0) && (reinterpret_cast<uintptr_t>(e->d_weight) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(d_out) % 16 == 0) && (out_ld % 4 == 0) &&
                       (e->dim % 4 == 0);
    const int lanes = vec16 ? e->dim / nf : e->dim;
    DLRM_REQUIRE(lanes <= 256, "dim=%d is too wide for this kernel (max %d)", e->dim, vec16 ? 256 * nf : 256);

    // Work decomposition: groups of G lanes.  Short bags are batched so that a
    // group has about kUnroll rows to fetch; when the bags cannot fill the chip
    // with groups but are long, give each bag a whole wave.
    int G = 4;
    while (G < lanes && G < 64) G <<= 1;
    const double avg_len = (double)e->n_indices / (double)n_bags;
    const int64_t groups_per_wave = 64 / G;
    const bool wave_per_bag = groups_per_wave > 1 && avg_len >= 4.0 * groups_per_wave &&
                              n_bags / groups_per_wave < 256 * 16;
    int bpg = 1;
    if (avg_len < kUnroll) bpg = (int)(kUnroll / (avg_len < 0.5 ? 0.5 : avg_len) + 0.5);
    if (bpg < 1) bpg = 1;
    if (bpg > kMaxBpg) bpg = kMaxBpg;
    // a batch too small to give every compute unit eight workgroups at that depth spreads over more groups instead
    // (narrow rows: a wave holds 64 / G groups, so 65536 one-row bags of 256-byte rows are only 1024 workgroups at
    // four bags per group)
    const int64_t fill = n_bags / ((int64_t)(kFwdThreads / G) * 2048);
    if (bpg > fill) bpg = fill < 1 ? 1 : (int)fill;
    p.bpg = bpg;

    hipStream_t s = as_stream(stream);
    const bool i64 = e->idx_dtype == DLRM_I64, bf = e->row_dtype == DLRM_BF16;
    if (i64 && bf) return dispatch_fwd<int64_t, uint16_t>(p, vec16, lanes, wave_per_bag, s);
    if (i64) return dispatch_fwd<int64_t, float>(p, vec16, lanes, wave_per_bag, s);
    if (bf) return dispatch_fwd<int32_t, uint16_t>(p, vec16, lanes, wave_per_bag, s);
    return dispatch_fwd<int32_t, float>(p, vec16, lanes, wave_per_bag, s);
}
