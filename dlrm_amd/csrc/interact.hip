// Pairwise dot feature interaction on the gfx950 matrix cores, forward and
// backward, one sample per wave.
//
// forward : Z = X X^T for X = [F, D] (F <= 32 features padded to 32 rows) with
//           v_mfma_f32_32x32x2_f32; the strict lower triangle of Z plus a copy
//           of feature 0 form the output row.  A and B fragments of X X^T are
//           the same registers (lane l holds X[l&31][k]).
// backward: dX = (S + S^T) X + [d(dense copy) on feature 0], S = pair
//           gradients scattered into a 32x32 LDS tile.
// Memory bound (F*D*4 bytes in, (D + F(F-1)/2)*4 bytes out per sample); the
// sample is staged through LDS with 16-byte coalesced loads.  LDS rows are
// padded by 4 floats so the ds_read_b128 fragment reads are conflict-free.
#include "common.h"

namespace dlrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// eight consecutive float32 values -> their three bf16 pieces as MFMA operands (8 reduction elements per lane)
struct Pieces8 {
    bf16x8 h, m, l;
};
__device__ inline Pieces8 split3x8(const float4 &a, const float4 &b) {
    unsigned h[4], m[4], l[4];
    split3_pair(f32x2{a.x, a.y}, h[0], m[0], l[0]);
    split3_pair(f32x2{a.z, a.w}, h[1], m[1], l[1]);
    split3_pair(f32x2{b.x, b.y}, h[2], m[2], l[2]);
    split3_pair(f32x2{b.z, b.w}, h[3], m[3], l[3]);
    return Pieces8{__builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]}),
                   __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]}),
                   __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]})};
}

// acc += A B with float32 accuracy from the pieces of A and B: the six leading cross terms, smallest first
__device__ inline f32x16 mfma_bf16x3(const Pieces8 &a, const Pieces8 &b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
    return acc;
}

constexpr int kIntThreads = 256;
constexpr int kIntWaves = kIntThreads / 64;
constexpr int kFP = 32;  // padded feature count
constexpr int kSLd = 36; // row stride of the 32x32 gradient tile

// A sample's [F, D] block travels global -> registers -> LDS so that the NEXT sample's loads are in
// flight while the current one is being multiplied.  NV = ceil(F*D/4 / 64) 16-byte vectors per lane
// is a template parameter: fully unrolled, only the last vector is partial, and the LDS offset of
// every vector is computed once (no division in the loop, no branches around the loads).
template <int NV>
__device__ inline void sample_offsets(int n4, int D, int ld, int lane, int (&lds_off)[NV]) {
#pragma unroll
    for (int u = 0; u < NV; u++) {
        int i = lane + 64 * u;
        if (i >= n4) i = n4 - 1;   // lanes past the end re-read the final vector and park it in the same place
        const int e = i * 4, row = e / D, col = e - row * D;
        lds_off[u] = row * ld + col;
    }
}

template <int NV>
__device__ inline void sample_fetch(const float *__restrict__ xb, int last_idx, int lane, f32x4 (&v)[NV]) {
#pragma unroll
    for (int u = 0; u < NV - 1; u++) v[u] = reinterpret_cast<const f32x4 *>(xb)[lane + 64 * u];
    v[NV - 1] = reinterpret_cast<const f32x4 *>(xb)[last_idx];
}

template <int NV>
__device__ inline void sample_park(float *__restrict__ sx, const int (&lds_off)[NV], const f32x4 (&v)[NV]) {
#pragma unroll
    for (int u = 0; u < NV; u++) *reinterpret_cast<f32x4 *>(sx + lds_off[u]) = v[u];
}

// Z = X X^T of the sample staged at sx ([32][ld], rows F.. zero) and the sample's output row: the dense copy of
// feature 0, the strict lower triangle of Z, zeros up to out_ld
// (vmax: running maximum of the magnitudes this lane has stored into output rows, for callers that want max |out|)
__device__ inline void fwd_sample(const float *__restrict__ sx, int F, int D, int ld, float *__restrict__ o, int64_t out_ld,
                                  int lane, float &vmax) {
    const int row = lane & 31, h = lane >> 5;
    const int n_pairs = F * (F - 1) / 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    if (D % 16 == 0) {
        // float32 products on the bf16 matrix pipe (three exact pieces per value, six cross terms): 6 x 32 cycles
        // per 16 reduction elements instead of 8 x 64 on the float32 MFMA, so that the matrix work hides
        // behind the sample's memory traffic.  Lane (row, h) holds elements 16q + 8h .. + 7 of its row.
        for (int q = 0; q < D / 16; q++) {
            const float *src = sx + row * ld + q * 16 + h * 8;
            const Pieces8 x = split3x8(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4));
            acc = mfma_bf16x3(x, x, acc);
        }
    } else {
        for (int q = 0; q < D / 8; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(sx + row * ld + q * 8 + h * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, v.w, acc, 0, 0, 0);
        }
    }
    // dense copy (feature 0) and zero padding of the row tail
    for (int d = lane; d < D; d += 64) {
        const float v = sx[d];
        o[d] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    for (int64_t c = D + n_pairs + lane; c < out_ld; c += 64) o[c] = 0.f;
    // register r of lane l holds Z[i][j], i = (r&3) + 8*(r>>2) + 4*h, j = l&31
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (i < F && row < i) {
            o[D + i * (i - 1) / 2 + row] = acc[r];
            vmax = fmaxf(vmax, fabsf(acc[r]));
        }
    }
}

template <int NV>
__global__ __launch_bounds__(kIntThreads) void interact_dot_fwd_kernel(const float *__restrict__ x, int64_t B,
                                                                        int F, int D, float *__restrict__ out,
                                                                        int64_t out_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float *sx = smem + (size_t)wave * kFP * ld;
    // rows F..31 stay zero for the whole kernel
    for (int i = lane; i < (kFP - F) * ld; i += 64) sx[F * ld + i] = 0.f;
    const int n4 = F * D / 4;
    const int64_t stride = (int64_t)gridDim.x * kIntWaves;

    int64_t b = (int64_t)blockIdx.x * kIntWaves + wave;
    int lds_off[NV];
    sample_offsets<NV>(n4, D, ld, lane, lds_off);
    const int last_idx = min(lane + 64 * (NV - 1), n4 - 1);
    f32x4 pre[NV];
    sample_fetch<NV>(x + (b < B ? b : 0) * F * D, last_idx, lane, pre);
    for (; b < B; b += stride) {
        sample_park<NV>(sx, lds_off, pre);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // next sample's loads fly behind the MFMAs (the last iteration re-reads its own sample)
        sample_fetch<NV>(x + (b + stride < B ? b + stride : b) * F * D, last_idx, lane, pre);
        __builtin_amdgcn_sched_barrier(0);
        float unused = 0.f;
        fwd_sample(sx, F, D, ld, out + b * out_ld, out_ld, lane, unused);
        __builtin_amdgcn_wave_barrier();
    }
}

// dX of the sample staged at sx (pair gradients scattered into the tile ss): dX = (S + S^T) X plus the dense copy's
// gradient on feature 0, written over the staged sample and streamed to gx with 16-byte stores
template <int NV>
__device__ inline void bwd_sample(float *__restrict__ sx, const float *__restrict__ ss, int F, int D, int ld,
                                  const float *__restrict__ go, float *__restrict__ gx, const int (&lds_off)[NV], int last_idx,
                                  int relu_mask0, int lane, float &vmax0) {
    // (vmax0: running maximum of the magnitudes of feature 0's gradient, the bottom MLP's incoming gradient)
    const int row = lane & 31, h = lane >> 5;
    // A operand S' = S + S^T (rows = output feature, reduction over the other feature, 32 deep = two steps of 16):
    // lane (row, h) holds elements 16s + 8h .. + 7 of its row, as three bf16 pieces (float32 products on the bf16
    // matrix pipe: 12 MFMAs of 32 cycles per block of 32 columns instead of 16 of 64)
    Pieces8 fa[2];
#pragma unroll
    for (int st = 0; st < 2; st++) {   // rows >= F only feed output rows that are never stored
        const float *src = ss + (row < F ? row : F - 1) * kSLd + st * 16 + h * 8;
        fa[st] = split3x8(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4));
    }
    for (int d0 = 0; d0 < D; d0 += 32) {
        const int d = d0 + row;
        const bool d_ok = d < D;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; st++) {
            // B operand X: lane (column d, h) holds features 16s + 8h .. + 7 of its column
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int k = st * 16 + h * 8 + e;
                const float v = sx[(k < F ? k : F - 1) * ld + (d_ok ? d : 0)];
                xv[e] = (d_ok && k < F) ? v : 0.f;
            }
            const Pieces8 fb = split3x8(float4{xv[0], xv[1], xv[2], xv[3]}, float4{xv[4], xv[5], xv[6], xv[7]});
            acc = mfma_bf16x3(fa[st], fb, acc);
        }
        // The 32 x 32 block of dX goes back into the columns of the staged sample it was computed from (every
        // operand read of these columns was issued above, by this same wave).  The whole sample then leaves
        // with 16-byte stores (a row per register would be 4-byte stores).
        if (d_ok) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i >= F) continue;
                float v = acc[r];
                if (i == 0) {
                    v += go[d];
                    if (relu_mask0 && !(sx[d] > 0.f)) v = 0.f;
                    vmax0 = fmaxf(vmax0, fabsf(v));
                }
                sx[i * ld + d] = v;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (gx == nullptr) return;     // the caller sends the staged gradient where it belongs
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int idx = u < NV - 1 ? lane + 64 * u : last_idx;
        reinterpret_cast<f32x4 *>(gx)[idx] = *reinterpret_cast<const f32x4 *>(sx + lds_off[u]);
    }
}

template <int NV>
__global__ __launch_bounds__(kIntThreads) void interact_dot_bwd_kernel(const float *__restrict__ x,
                                                                        const float *__restrict__ dout,
                                                                        int64_t dout_ld, int64_t B, int F, int D,
                                                                        float *__restrict__ dx, int relu_mask0,
                                                                        float *__restrict__ dx0_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float vmax0 = 0.f;
    // per wave: the sample [F][ld] and the gradient tile [F][kSLd] (+ one spare cell); only F rows
    // are kept (operand rows F..31 are masked at the reads), which lets two workgroups share a CU
    float *sx = smem + (size_t)wave * (F * ld + F * kSLd + 4);
    float *ss = sx + F * ld;
    // the gradient tile only ever has entries (i,j), (j,i) with j < i < F written: zero it once
    for (int i = lane; i < F * kSLd + 4; i += 64) ss[i] = 0.f;
    const int n_pairs = F * (F - 1) / 2;
    const int n4 = F * D / 4;
    const int64_t stride = (int64_t)gridDim.x * kIntWaves;
    // this lane's pair indices c = lane, lane+64, ... and their (i, j): c = i(i-1)/2 + j, 0 <= j < i
    constexpr int kMaxPairs = (kFP * (kFP - 1) / 2 + 63) / 64;   // 8
    int pij[kMaxPairs];
#pragma unroll
    for (int u = 0; u < kMaxPairs; u++) {
        const int c = lane + 64 * u;
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)c)) * 0.5f);
        while (i * (i - 1) / 2 > c) i--;
        while ((i + 1) * i / 2 <= c) i++;
        const int j = c - i * (i - 1) / 2;
        pij[u] = c < n_pairs ? (i << 8) | j : -1;
    }

    // LDS offsets of this lane's pair gradients (lanes past the last pair write a spare cell)
    int sij[kMaxPairs], sji[kMaxPairs];
#pragma unroll
    for (int u = 0; u < kMaxPairs; u++) {
        const bool real = pij[u] >= 0;
        const int i = real ? pij[u] >> 8 : 0, j = real ? pij[u] & 255 : 0;
        sij[u] = real ? i * kSLd + j : F * kSLd;     // lanes past the last pair write the spare cell
        sji[u] = real ? j * kSLd + i : F * kSLd;
    }
    int64_t b = (int64_t)blockIdx.x * kIntWaves + wave;
    int lds_off[NV];
    sample_offsets<NV>(n4, D, ld, lane, lds_off);
    const int last_idx = min(lane + 64 * (NV - 1), n4 - 1);
    f32x4 pre[NV];
    float pg[kMaxPairs];
    auto fetch_grad = [&](int64_t bb) {
        const float *go = dout + bb * dout_ld + D;
#pragma unroll
        for (int u = 0; u < kMaxPairs; u++) pg[u] = go[pij[u] >= 0 ? lane + 64 * u : 0];
    };
    sample_fetch<NV>(x + (b < B ? b : 0) * F * D, last_idx, lane, pre);
    fetch_grad(b < B ? b : 0);
    for (; b < B; b += stride) {
        sample_park<NV>(sx, lds_off, pre);
#pragma unroll
        for (int u = 0; u < kMaxPairs; u++) {
            const float g = pij[u] >= 0 ? pg[u] : 0.f;
            ss[sij[u]] = g;
            ss[sji[u]] = g;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        {
            const int64_t nb = b + stride < B ? b + stride : b;
            sample_fetch<NV>(x + nb * F * D, last_idx, lane, pre);
            fetch_grad(nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<NV>(sx, ss, F, D, ld, dout + b * dout_ld, dx + b * F * D, lds_off, last_idx, relu_mask0, lane, vmax0);
        __builtin_amdgcn_wave_barrier();
    }
    if (dx0_max != nullptr) fold_absmax_block(vmax0, dx0_max);
}

// ---------------------------------------------------------------------------------------------------------
// The forward kernel with the EmbeddingBag lookup folded in, for batches whose bags hold exactly one id
// (Criteo's one-hot categorical features): feature f >= 1 of sample b IS row indices[(f-1) B + b] of table
// f-1, so the sample's block is gathered straight from the tables instead of being written by the EmbeddingBag
// kernel and read back (inference saves a write and a read of [B, T, D]; a training step has the kernel leave a
// copy of the block for the backward pass and still saves the read).  Feature 0 (the bottom MLP's output) comes
// from x0 + b x0_ld.
// A workgroup owns chunks of 32 consecutive samples: the chunk's ids -- T runs of 32 consecutive ids in the
// table-major index array, one 128-byte line each -- are staged in LDS first (checked: offsets[g] == g, id inside
// its table; a violation raises the error flag and the id counts as a zero row), then wave w takes samples
// w, w+4, ... of the chunk.
constexpr int kChunk = 32;

struct GatherArgs {
    const float *weight;
    const int64_t *tro;       // [T+1] first row of every table, or NULL (all tables address rows [0, total_rows))
    int64_t total_rows;
    const void *indices, *offsets;
    int T;
    int32_t *err;
    const float *x0;
    int64_t x0_ld;
};

template <typename IdxT>
__device__ inline void stage_ids(const GatherArgs &g, int64_t B, int64_t b0, int *__restrict__ sidx) {
    const IdxT *__restrict__ idx = static_cast<const IdxT *>(g.indices);
    const IdxT *__restrict__ off = static_cast<const IdxT *>(g.offsets);
    for (int i = threadIdx.x; i < g.T * kChunk; i += kIntThreads) {
        const int t = i / kChunk, j = i % kChunk;
        const int64_t b = b0 + j;
        int v = -1;
        if (b < B) {
            const int64_t pos = (int64_t)t * B + b;
            const int64_t id = (int64_t)idx[pos];
            const int64_t rows = g.tro ? g.tro[t + 1] - g.tro[t] : g.total_rows;
            const bool one = (int64_t)off[pos] == pos;
            const bool in = id >= 0 && id < rows;
            if (one && in) v = (int)((g.tro ? g.tro[t] : 0) + id);   // row of the whole pool
            else if (g.err) *g.err = one ? 1 : 2;
        }
        sidx[i] = v;
    }
}

// Where the 16-byte vectors of a lane sit -- vector u of lane l is vector l + 64 u of the sample's [F, D] block --
// is recomputed from the lane number wherever it is needed instead of being kept in registers (14 vectors would
// hold 28 of them across the whole loop): `lane` goes
// through an empty asm statement first, so the compiler cannot hoist the arithmetic out of the sample loop.
struct VecPlace {
    int f, col, lds;     // feature, first column, float offset in the staged sample
};
__device__ inline VecPlace vec_place(int lane_now, int u, int n4, int D, int ld) {   // D: a power of two
    int i = lane_now + 64 * u;
    if (i >= n4) i = n4 - 1;       // lanes past the end re-read the final vector and park it in the same place
    const int e = i * 4;
    const int f = e >> (__ffs(D) - 1);
    VecPlace p;
    p.f = f; p.col = e & (D - 1); p.lds = f * ld + p.col;
    return p;
}
__device__ inline int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// sidx holds rows of the whole pool (table's first row + id), or -1 for a lookup that counts as a zero row
template <int NV>
__device__ inline void gather_fetch(const GatherArgs &g, const int *__restrict__ sidx, int j, int64_t b, int n4, int D, int ld,
                                    int lane, f32x4 (&v)[NV]) {
    const int lv = opaque(lane);
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const VecPlace p = vec_place(lv, u, n4, D, ld);
        const int id = p.f == 0 ? 0 : sidx[(p.f - 1) * kChunk + j];
        const float *src = p.f == 0 ? g.x0 + b * g.x0_ld + p.col : g.weight + (int64_t)(id < 0 ? 0 : id) * D + p.col;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(src);
        v[u] = id < 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : x;
    }
}

template <int NV>
__device__ inline void gather_park(float *__restrict__ sx, int n4, int D, int ld, int lane, const f32x4 (&v)[NV]) {
    const int lv = opaque(lane);
#pragma unroll
    for (int u = 0; u < NV; u++) *reinterpret_cast<f32x4 *>(sx + vec_place(lv, u, n4, D, ld).lds) = v[u];
}

template <int NV, typename IdxT>
__global__ __launch_bounds__(kIntThreads, 2) void interact_dot_fwd_gather_kernel(GatherArgs g, int64_t B, int F, int D,
                                                                               float *__restrict__ out, int64_t out_ld,
                                                                               float *__restrict__ x_copy,
                                                                               float *__restrict__ out_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float vmax = 0.f;
    float *sx = smem + (size_t)wave * kFP * ld;
    int *sidx = reinterpret_cast<int *>(smem + (size_t)kIntWaves * kFP * ld);
    for (int i = lane; i < (kFP - F) * ld; i += 64) sx[F * ld + i] = 0.f;   // rows F..31 stay zero
    const int n4 = F * D / 4;
    const int64_t n_chunks = (B + kChunk - 1) / kChunk;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const int64_t b0 = c * kChunk;
        __syncthreads();     // the previous chunk's ids are no longer needed
        stage_ids<IdxT>(g, B, b0, sidx);
        __syncthreads();
        f32x4 pre[NV];
        int j = wave;
        if (b0 + j < B) gather_fetch<NV>(g, sidx, j, b0 + j, n4, D, ld, lane, pre);
        for (; j < kChunk && b0 + j < B; j += kIntWaves) {
            gather_park<NV>(sx, n4, D, ld, lane, pre);
            if (x_copy) {   // the gathered block, for a backward pass that reads it instead of gathering again
                float *xc = x_copy + (b0 + j) * (int64_t)(F * D);
#pragma unroll
                for (int u = 0; u < NV; u++) reinterpret_cast<f32x4 *>(xc)[min(lane + 64 * u, n4 - 1)] = pre[u];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            {   // the next sample's loads fly behind the MFMAs (the chunk's last one re-reads itself)
                const int jn = (j + kIntWaves < kChunk && b0 + j + kIntWaves < B) ? j + kIntWaves : j;
                gather_fetch<NV>(g, sidx, jn, b0 + jn, n4, D, ld, lane, pre);
            }
            __builtin_amdgcn_sched_barrier(0);
            fwd_sample(sx, F, D, ld, out + (b0 + j) * out_ld, out_ld, lane, vmax);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (out_max != nullptr) fold_absmax_block(vmax, out_max);
}

// The backward kernel with the same lookup folded in: a training step whose forward pass gathered the rows itself
// gathers them again here (the tables are not touched between the two) instead of having the forward kernel write a
// copy of the [B, F, D] block and reading that: the forward pass writes F*D*4 bytes per sample less, this kernel reads
// the same bytes from the tables that it would read from the copy.  Same chunks of 32 samples, same id checks (a
// lookup that counted as a zero row in the forward pass counts as one here).
template <int NV, typename IdxT>
__global__ __launch_bounds__(kIntThreads, 2) void interact_dot_bwd_gather_kernel(GatherArgs g, const float *__restrict__ dout,
                                                                               int64_t dout_ld, int64_t B, int F, int D,
                                                                               float *__restrict__ dx, int relu_mask0,
                                                                               float *__restrict__ dx0_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    const int per_wave = F * ld + F * kSLd + 4;
    float vmax0 = 0.f;
    float *sx = smem + (size_t)wave * per_wave;
    float *ss = sx + F * ld;
    int *sidx = reinterpret_cast<int *>(smem + (size_t)kIntWaves * per_wave);
    for (int i = lane; i < F * kSLd + 4; i += 64) ss[i] = 0.f;
    const int n_pairs = F * (F - 1) / 2;
    const int n4 = F * D / 4;
    constexpr int kMaxPairs = (kFP * (kFP - 1) / 2 + 63) / 64;   // 8
    int pij[kMaxPairs];   // (i << 8) | j of this lane's pair gradients, -1 past the last pair
#pragma unroll
    for (int u = 0; u < kMaxPairs; u++) {
        const int c = lane + 64 * u;
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)c)) * 0.5f);
        while (i * (i - 1) / 2 > c) i--;
        while ((i + 1) * i / 2 <= c) i++;
        pij[u] = c < n_pairs ? (i << 8) | (c - i * (i - 1) / 2) : -1;
    }
    const int last_idx = min(lane + 64 * (NV - 1), n4 - 1);
    f32x4 pre[NV];
    float pg[kMaxPairs];
    auto fetch_grad = [&](int64_t bb) {
        const float *go = dout + bb * dout_ld + D;
#pragma unroll
        for (int u = 0; u < kMaxPairs; u++) pg[u] = go[pij[u] >= 0 ? lane + 64 * u : 0];
    };
    const int64_t n_chunks = (B + kChunk - 1) / kChunk;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const int64_t b0 = c * kChunk;
        __syncthreads();     // the previous chunk's ids are no longer needed
        stage_ids<IdxT>(g, B, b0, sidx);
        __syncthreads();
        int j = wave;
        if (b0 + j < B) {
            gather_fetch<NV>(g, sidx, j, b0 + j, n4, D, ld, lane, pre);
            fetch_grad(b0 + j);
        }
        for (; j < kChunk && b0 + j < B; j += kIntWaves) {
            const int64_t b = b0 + j;
            gather_park<NV>(sx, n4, D, ld, lane, pre);
#pragma unroll
            for (int u = 0; u < kMaxPairs; u++) {
                const bool real = pij[u] >= 0;
                const int i = pij[u] >> 8, jj = pij[u] & 255;
                const float gv = real ? pg[u] : 0.f;
                ss[real ? i * kSLd + jj : F * kSLd] = gv;     // lanes past the last pair write the spare cell
                ss[real ? jj * kSLd + i : F * kSLd] = gv;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            {   // the next sample's loads fly behind the MFMAs (the chunk's last one re-reads itself)
                const int jn = (j + kIntWaves < kChunk && b + kIntWaves < B) ? j + kIntWaves : j;
                gather_fetch<NV>(g, sidx, jn, b0 + jn, n4, D, ld, lane, pre);
                fetch_grad(b0 + jn);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                int lds_off[NV];   // (unused by a bwd_sample that leaves the store to its caller)
#pragma unroll
                for (int u = 0; u < NV; u++) lds_off[u] = 0;
                bwd_sample<NV>(sx, ss, F, D, ld, dout + b * dout_ld, nullptr, lds_off, last_idx, relu_mask0, lane, vmax0);
            }
            {
                float *gx = dx + b * (int64_t)(F * D);
                const int lv = opaque(lane);
#pragma unroll
                for (int u = 0; u < NV; u++)
                    reinterpret_cast<f32x4 *>(gx)[min(lane + 64 * u, n4 - 1)] =
                        *reinterpret_cast<const f32x4 *>(sx + vec_place(lv, u, n4, D, ld).lds);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (dx0_max != nullptr) fold_absmax_block(vmax0, dx0_max);
}

// ---------------------------------------------------------------------------------------------------------
// Both kernels on features that do not sit in one [B, F, D] block: feature f of sample b is D floats at
// base[f] + b * stride[f].  A sharded rank receives the pooled vectors as one block per source rank and sends the
// gradients back the same way; with views the interaction reads and writes those exchange buffers directly (no
// unpack / pack pass over [B, T, D] in between).  The table of views travels as a kernel argument and is parked in
// LDS; a lane recomputes the place of its vectors per use (vec_place: D is a power of two).
struct Views {
    float *base[kFP];
    int64_t stride[kFP];
};

template <int NV>
__device__ inline void views_fetch(float *const *__restrict__ sb, const int64_t *__restrict__ sst, int64_t b, int n4, int D, int ld,
                                   int lane, f32x4 (&v)[NV]) {
    const int lv = opaque(lane);
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const VecPlace p = vec_place(lv, u, n4, D, ld);
        v[u] = *reinterpret_cast<const f32x4 *>(sb[p.f] + b * sst[p.f] + p.col);
    }
}

template <int NV>
__global__ __launch_bounds__(kIntThreads, 2) void interact_dot_fwd_views_kernel(Views xv, int64_t B, int F, int D,
                                                                              float *__restrict__ out, int64_t out_ld,
                                                                              float *__restrict__ out_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    float *sx = smem + (size_t)wave * kFP * ld;
    float **sb = reinterpret_cast<float **>(smem + (size_t)kIntWaves * kFP * ld);
    int64_t *sst = reinterpret_cast<int64_t *>(sb + kFP);
    if (threadIdx.x < F) { sb[threadIdx.x] = xv.base[threadIdx.x]; sst[threadIdx.x] = xv.stride[threadIdx.x]; }
    for (int i = lane; i < (kFP - F) * ld; i += 64) sx[F * ld + i] = 0.f;   // rows F..31 stay zero
    __syncthreads();
    const int n4 = F * D / 4;
    const int64_t stride = (int64_t)gridDim.x * kIntWaves;
    int64_t b = (int64_t)blockIdx.x * kIntWaves + wave;
    f32x4 pre[NV];
    float vmax = 0.f;
    views_fetch<NV>(sb, sst, b < B ? b : 0, n4, D, ld, lane, pre);
    for (; b < B; b += stride) {
        gather_park<NV>(sx, n4, D, ld, lane, pre);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        views_fetch<NV>(sb, sst, b + stride < B ? b + stride : b, n4, D, ld, lane, pre);
        __builtin_amdgcn_sched_barrier(0);
        fwd_sample(sx, F, D, ld, out + b * out_ld, out_ld, lane, vmax);
        __builtin_amdgcn_wave_barrier();
    }
    if (out_max != nullptr) fold_absmax_block(vmax, out_max);
}

template <int NV>
__global__ __launch_bounds__(kIntThreads, 2) void interact_dot_bwd_views_kernel(Views xv, Views dv, const float *__restrict__ dout,
                                                                              int64_t dout_ld, int64_t B, int F, int D,
                                                                              int relu_mask0, float *__restrict__ dx0_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = D + 4;
    const int per_wave = F * ld + F * kSLd + 4;
    float vmax0 = 0.f;
    float *sx = smem + (size_t)wave * per_wave;
    float *ss = sx + F * ld;
    float **sb = reinterpret_cast<float **>(smem + (size_t)kIntWaves * per_wave);
    int64_t *sst = reinterpret_cast<int64_t *>(sb + kFP);
    float **db = reinterpret_cast<float **>(sst + kFP);
    int64_t *dst = reinterpret_cast<int64_t *>(db + kFP);
    if (threadIdx.x < F) {
        sb[threadIdx.x] = xv.base[threadIdx.x]; sst[threadIdx.x] = xv.stride[threadIdx.x];
        db[threadIdx.x] = dv.base[threadIdx.x]; dst[threadIdx.x] = dv.stride[threadIdx.x];
    }
    for (int i = lane; i < F * kSLd + 4; i += 64) ss[i] = 0.f;
    __syncthreads();
    const int n_pairs = F * (F - 1) / 2;
    const int n4 = F * D / 4;
    constexpr int kMaxPairs = (kFP * (kFP - 1) / 2 + 63) / 64;   // 8
    int pij[kMaxPairs];   // (i << 8) | j of this lane's pair gradients, -1 past the last pair
#pragma unroll
    for (int u = 0; u < kMaxPairs; u++) {
        const int c = lane + 64 * u;
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)c)) * 0.5f);
        while (i * (i - 1) / 2 > c) i--;
        while ((i + 1) * i / 2 <= c) i++;
        pij[u] = c < n_pairs ? (i << 8) | (c - i * (i - 1) / 2) : -1;
    }
    const int last_idx = min(lane + 64 * (NV - 1), n4 - 1);
    const int64_t stride = (int64_t)gridDim.x * kIntWaves;
    int64_t b = (int64_t)blockIdx.x * kIntWaves + wave;
    f32x4 pre[NV];
    float pg[kMaxPairs];
    auto fetch_grad = [&](int64_t bb) {
        const float *go = dout + bb * dout_ld + D;
#pragma unroll
        for (int u = 0; u < kMaxPairs; u++) pg[u] = go[pij[u] >= 0 ? lane + 64 * u : 0];
    };
    views_fetch<NV>(sb, sst, b < B ? b : 0, n4, D, ld, lane, pre);
    fetch_grad(b < B ? b : 0);
    for (; b < B; b += stride) {
        gather_park<NV>(sx, n4, D, ld, lane, pre);
#pragma unroll
        for (int u = 0; u < kMaxPairs; u++) {
            const bool real = pij[u] >= 0;
            const int i = pij[u] >> 8, jj = pij[u] & 255;
            const float gv = real ? pg[u] : 0.f;
            ss[real ? i * kSLd + jj : F * kSLd] = gv;     // lanes past the last pair write the spare cell
            ss[real ? jj * kSLd + i : F * kSLd] = gv;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        {
            const int64_t nb = b + stride < B ? b + stride : b;
            views_fetch<NV>(sb, sst, nb, n4, D, ld, lane, pre);
            fetch_grad(nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int lds_off[NV];   // (unused by a bwd_sample that leaves the store to its caller)
#pragma unroll
            for (int u = 0; u < NV; u++) lds_off[u] = 0;
            bwd_sample<NV>(sx, ss, F, D, ld, dout + b * dout_ld, nullptr, lds_off, last_idx, relu_mask0, lane, vmax0);
        }
        {   // every feature's gradient to its own place
            const int lv = opaque(lane);
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const VecPlace p = vec_place(lv, u, n4, D, ld);
                *reinterpret_cast<f32x4 *>(db[p.f] + b * dst[p.f] + p.col) = *reinterpret_cast<const f32x4 *>(sx + p.lds);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (dx0_max != nullptr) fold_absmax_block(vmax0, dx0_max);
}

}  // namespace dlrm

using namespace dlrm;

static int check_interact(const float *d_x, int64_t B, int32_t F, int32_t D) {
    DLRM_REQUIRE(B >= 0 && F >= 1 && D >= 1, "bad shape B=%lld F=%d D=%d", (long long)B, F, D);
    DLRM_REQUIRE(F <= kFP, "dot interaction supports at most %d features, got %d", kFP, F);
    DLRM_REQUIRE(D % 8 == 0 && F * D <= 4096, "dot interaction needs dim %% 8 == 0 and F*dim <= 4096, got F=%d dim=%d", F, D);
    DLRM_REQUIRE(B == 0 || (d_x != nullptr && reinterpret_cast<uintptr_t>(d_x) % 16 == 0), "x must be 16-byte aligned");
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_forward(const float *d_x, int64_t B, int32_t F, int32_t D, float *d_out,
                                         int64_t out_ld, dlrm_stream_t stream) {
    if (int rc = check_interact(d_x, B, F, D)) return rc;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_out != nullptr && out_ld >= D + F * (F - 1) / 2, "out is NULL or out_ld too small");
    const size_t lds = sizeof(float) * kIntWaves * kFP * (D + 4);
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    const int nv = (F * D / 4 + 63) / 64;
#define DLRM_INT_FWD(NV)                                                                                            \
    case NV:                                                                                                        \
        DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_fwd_kernel<NV>),                   \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL(interact_dot_fwd_kernel<NV>, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), d_x, \
                           B, F, D, d_out, out_ld);                                                                 \
        break;
    switch (nv) {
        DLRM_INT_FWD(1) DLRM_INT_FWD(2) DLRM_INT_FWD(3) DLRM_INT_FWD(4) DLRM_INT_FWD(5) DLRM_INT_FWD(6) DLRM_INT_FWD(7)
        DLRM_INT_FWD(8) DLRM_INT_FWD(9) DLRM_INT_FWD(10) DLRM_INT_FWD(11) DLRM_INT_FWD(12) DLRM_INT_FWD(13)
        DLRM_INT_FWD(14) DLRM_INT_FWD(15) DLRM_INT_FWD(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_FWD
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_backward(const float *d_x, const float *d_dout, int64_t dout_ld, int64_t B,
                                          int32_t F, int32_t D, float *d_dx, int32_t relu_mask_feature0,
                                          dlrm_stream_t stream) {
    return dlrm::interact_dot_backward_max(d_x, d_dout, dout_ld, B, F, D, d_dx, relu_mask_feature0, nullptr, stream);
}

// the same launch; d_dx0_max (or NULL): max |gradient of feature 0| is folded into it (the hosted model's bottom MLP
// takes that gradient as an operand of its f16x2 products)
int dlrm::interact_dot_backward_max(const float *d_x, const float *d_dout, int64_t dout_ld, int64_t B, int32_t F, int32_t D,
                                    float *d_dx, int32_t relu_mask_feature0, float *d_dx0_max, dlrm_stream_t stream) {
    if (int rc = check_interact(d_x, B, F, D)) return rc;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_dout != nullptr && d_dx != nullptr && dout_ld >= D + F * (F - 1) / 2,
                 "dout/dx is NULL or dout_ld too small");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(d_dx) % 16 == 0, "dx must be 16-byte aligned");
    const size_t lds = sizeof(float) * kIntWaves * ((size_t)F * (D + 4) + F * kSLd + 4);
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    const int nv = (F * D / 4 + 63) / 64;
#define DLRM_INT_BWD(NV)                                                                                            \
    case NV:                                                                                                        \
        DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_bwd_kernel<NV>),                   \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL(interact_dot_bwd_kernel<NV>, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), d_x, \
                           d_dout, dout_ld, B, F, D, d_dx, relu_mask_feature0, d_dx0_max);                          \
        break;
    switch (nv) {
        DLRM_INT_BWD(1) DLRM_INT_BWD(2) DLRM_INT_BWD(3) DLRM_INT_BWD(4) DLRM_INT_BWD(5) DLRM_INT_BWD(6) DLRM_INT_BWD(7)
        DLRM_INT_BWD(8) DLRM_INT_BWD(9) DLRM_INT_BWD(10) DLRM_INT_BWD(11) DLRM_INT_BWD(12) DLRM_INT_BWD(13)
        DLRM_INT_BWD(14) DLRM_INT_BWD(15) DLRM_INT_BWD(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_BWD
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

static int check_gather(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, GatherArgs *g) {
    DLRM_REQUIRE(e != nullptr, "embbag descriptor is NULL");
    DLRM_REQUIRE(e->num_tables >= 1 && e->dim >= 1 && e->batch >= 0, "bad shape T=%d dim=%d batch=%lld", e->num_tables, e->dim,
                 (long long)e->batch);
    DLRM_REQUIRE(e->row_dtype == DLRM_F32, "the gathering interaction reads float32 rows (pool bf16 rows with dlrm_embbag_forward)");
    DLRM_REQUIRE((e->dim & (e->dim - 1)) == 0, "the gathering interaction needs a power-of-two dim, got %d", e->dim);
    DLRM_REQUIRE(e->idx_dtype == DLRM_I32 || e->idx_dtype == DLRM_I64, "idx_dtype %d", e->idx_dtype);
    DLRM_REQUIRE(e->d_shard_row_begin == nullptr, "row ranges of split tables are pooled by dlrm_embbag_forward");
    DLRM_REQUIRE(e->total_rows >= 1 && e->total_rows < (1ll << 31), "total_rows out of range");
    DLRM_REQUIRE(e->n_indices >= (int64_t)e->num_tables * e->batch, "one id per bag needs %lld ids, got %lld",
                 (long long)e->num_tables * e->batch, (long long)e->n_indices);
    if (int rc = check_interact(d_x0, e->batch, e->num_tables + 1, e->dim)) return rc;
    if (e->batch == 0) return DLRM_OK;
    DLRM_REQUIRE(e->d_weight && e->d_indices && e->d_offsets, "NULL table / indices / offsets");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(e->d_weight) % 16 == 0 && e->dim % 4 == 0 && x0_ld % 4 == 0 && x0_ld >= e->dim,
                 "rows must be 16-byte aligned");
    g->weight = static_cast<const float *>(e->d_weight); g->tro = e->d_table_row_offsets; g->total_rows = e->total_rows;
    g->indices = e->d_indices; g->offsets = e->d_offsets; g->T = e->num_tables; g->err = e->d_error_flag;
    g->x0 = d_x0; g->x0_ld = x0_ld;
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_forward_gather(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, float *d_out,
                                                int64_t out_ld, float *d_x_copy, dlrm_stream_t stream) {
    return dlrm::interact_dot_forward_gather_max(e, d_x0, x0_ld, d_out, out_ld, d_x_copy, nullptr, stream);
}

// the same launch; d_out_max (or NULL): max |out| is folded into it
int dlrm::interact_dot_forward_gather_max(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, float *d_out, int64_t out_ld,
                                          float *d_x_copy, float *d_out_max, dlrm_stream_t stream) {
    GatherArgs g{};
    if (int rc = check_gather(e, d_x0, x0_ld, &g)) return rc;
    const int64_t B = e->batch;
    const int F = e->num_tables + 1, D = e->dim;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_out != nullptr && out_ld >= D + F * (F - 1) / 2, "out is NULL or out_ld too small");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(d_x_copy) % 16 == 0, "x_copy must be 16-byte aligned");
    const size_t lds = sizeof(float) * kIntWaves * kFP * (D + 4) + sizeof(int) * (size_t)g.T * kChunk;
    const int64_t want = ceil_div(B, kChunk);
    const unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
    const int nv = (F * D / 4 + 63) / 64;
    const bool i64 = e->idx_dtype == DLRM_I64;
#define DLRM_INT_FWDG(NV)                                                                                              \
    case NV:                                                                                                           \
        if (i64) {                                                                                                     \
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_fwd_gather_kernel<NV, int64_t>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
            hipLaunchKernelGGL((interact_dot_fwd_gather_kernel<NV, int64_t>), dim3(blocks), dim3(kIntThreads), lds,    \
                               as_stream(stream), g, B, F, D, d_out, out_ld, d_x_copy, d_out_max);                     \
        } else {                                                                                                       \
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_fwd_gather_kernel<NV, int32_t>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
            hipLaunchKernelGGL((interact_dot_fwd_gather_kernel<NV, int32_t>), dim3(blocks), dim3(kIntThreads), lds,    \
                               as_stream(stream), g, B, F, D, d_out, out_ld, d_x_copy, d_out_max);                     \
        }                                                                                                              \
        break;
    switch (nv) {
        DLRM_INT_FWDG(1) DLRM_INT_FWDG(2) DLRM_INT_FWDG(3) DLRM_INT_FWDG(4) DLRM_INT_FWDG(5) DLRM_INT_FWDG(6) DLRM_INT_FWDG(7)
        DLRM_INT_FWDG(8) DLRM_INT_FWDG(9) DLRM_INT_FWDG(10) DLRM_INT_FWDG(11) DLRM_INT_FWDG(12) DLRM_INT_FWDG(13)
        DLRM_INT_FWDG(14) DLRM_INT_FWDG(15) DLRM_INT_FWDG(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_FWDG
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_backward_gather(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, const float *d_dout,
                                                 int64_t dout_ld, float *d_dx, int32_t relu_mask_feature0, dlrm_stream_t stream) {
    return dlrm::interact_dot_backward_gather_max(e, d_x0, x0_ld, d_dout, dout_ld, d_dx, relu_mask_feature0, nullptr, stream);
}

// the same launch; d_dx0_max (or NULL): max |gradient of feature 0| is folded into it
int dlrm::interact_dot_backward_gather_max(const dlrm_embbag *e, const float *d_x0, int64_t x0_ld, const float *d_dout,
                                           int64_t dout_ld, float *d_dx, int32_t relu_mask_feature0, float *d_dx0_max,
                                           dlrm_stream_t stream) {
    GatherArgs g{};
    if (int rc = check_gather(e, d_x0, x0_ld, &g)) return rc;
    const int64_t B = e->batch;
    const int F = e->num_tables + 1, D = e->dim;
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_dout != nullptr && d_dx != nullptr && dout_ld >= D + F * (F - 1) / 2, "dout/dx is NULL or dout_ld too small");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(d_dx) % 16 == 0, "dx must be 16-byte aligned");
    const size_t lds = sizeof(float) * kIntWaves * ((size_t)F * (D + 4) + F * kSLd + 4) + sizeof(int) * (size_t)g.T * kChunk;
    const int64_t want = ceil_div(B, kChunk);
    const unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
    const int nv = (F * D / 4 + 63) / 64;
    const bool i64 = e->idx_dtype == DLRM_I64;
#define DLRM_INT_BWDG(NV)                                                                                              \
    case NV:                                                                                                           \
        if (i64) {                                                                                                     \
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_bwd_gather_kernel<NV, int64_t>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
            hipLaunchKernelGGL((interact_dot_bwd_gather_kernel<NV, int64_t>), dim3(blocks), dim3(kIntThreads), lds,    \
                               as_stream(stream), g, d_dout, dout_ld, B, F, D, d_dx, relu_mask_feature0, d_dx0_max);   \
        } else {                                                                                                       \
            DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_bwd_gather_kernel<NV, int32_t>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
            hipLaunchKernelGGL((interact_dot_bwd_gather_kernel<NV, int32_t>), dim3(blocks), dim3(kIntThreads), lds,    \
                               as_stream(stream), g, d_dout, dout_ld, B, F, D, d_dx, relu_mask_feature0, d_dx0_max);   \
        }                                                                                                              \
        break;
    switch (nv) {
        DLRM_INT_BWDG(1) DLRM_INT_BWDG(2) DLRM_INT_BWDG(3) DLRM_INT_BWDG(4) DLRM_INT_BWDG(5) DLRM_INT_BWDG(6) DLRM_INT_BWDG(7)
        DLRM_INT_BWDG(8) DLRM_INT_BWDG(9) DLRM_INT_BWDG(10) DLRM_INT_BWDG(11) DLRM_INT_BWDG(12) DLRM_INT_BWDG(13)
        DLRM_INT_BWDG(14) DLRM_INT_BWDG(15) DLRM_INT_BWDG(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_BWDG
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

static int check_views(const dlrm_feature_view *v, int32_t F, int32_t D, const char *what, Views *out) {
    DLRM_REQUIRE(v != nullptr, "%s views are NULL", what);
    for (int f = 0; f < F; f++) {
        DLRM_REQUIRE(v[f].d_base != nullptr && reinterpret_cast<uintptr_t>(v[f].d_base) % 16 == 0 && v[f].stride % 4 == 0 &&
                     v[f].stride >= D, "%s view %d: base must be 16-byte aligned, stride a multiple of 4 floats >= dim", what, f);
        out->base[f] = v[f].d_base;
        out->stride[f] = v[f].stride;
    }
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_forward_views(const dlrm_feature_view *x_views, int64_t B, int32_t F, int32_t D, float *d_out,
                                               int64_t out_ld, dlrm_stream_t stream) {
    return dlrm::interact_dot_forward_views_max(x_views, B, F, D, d_out, out_ld, nullptr, stream);
}

// d_out_max (may be NULL): max |out| is folded into it (atomic max on the float's bits: zero it first)
int dlrm::interact_dot_forward_views_max(const dlrm_feature_view *x_views, int64_t B, int32_t F, int32_t D, float *d_out,
                                         int64_t out_ld, float *d_out_max, dlrm_stream_t stream) {
    DLRM_REQUIRE(B >= 0 && F >= 1 && F <= kFP && D >= 8 && (D & (D - 1)) == 0 && F * D <= 4096,
                 "views need 1 <= F <= %d, dim a power of two >= 8, F*dim <= 4096; got F=%d dim=%d", kFP, F, D);
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_out != nullptr && out_ld >= D + F * (F - 1) / 2, "out is NULL or out_ld too small");
    Views xv{};
    if (int rc = check_views(x_views, F, D, "x", &xv)) return rc;
    const size_t lds = sizeof(float) * kIntWaves * kFP * (D + 4) + 16 * kFP;
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    const int nv = (F * D / 4 + 63) / 64;
#define DLRM_INT_FWDV(NV)                                                                                              \
    case NV:                                                                                                           \
        DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_fwd_views_kernel<NV>),                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL(interact_dot_fwd_views_kernel<NV>, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), \
                           xv, B, F, D, d_out, out_ld, d_out_max);                                                     \
        break;
    switch (nv) {
        DLRM_INT_FWDV(1) DLRM_INT_FWDV(2) DLRM_INT_FWDV(3) DLRM_INT_FWDV(4) DLRM_INT_FWDV(5) DLRM_INT_FWDV(6) DLRM_INT_FWDV(7)
        DLRM_INT_FWDV(8) DLRM_INT_FWDV(9) DLRM_INT_FWDV(10) DLRM_INT_FWDV(11) DLRM_INT_FWDV(12) DLRM_INT_FWDV(13)
        DLRM_INT_FWDV(14) DLRM_INT_FWDV(15) DLRM_INT_FWDV(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_FWDV
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_interact_dot_backward_views(const dlrm_feature_view *x_views, const float *d_dout, int64_t dout_ld, int64_t B,
                                                int32_t F, int32_t D, const dlrm_feature_view *dx_views,
                                                int32_t relu_mask_feature0, dlrm_stream_t stream) {
    return dlrm::interact_dot_backward_views_max(x_views, d_dout, dout_ld, B, F, D, dx_views, relu_mask_feature0, nullptr, stream);
}

// d_dx0_max (may be NULL): max |gradient of feature 0| is folded into it
int dlrm::interact_dot_backward_views_max(const dlrm_feature_view *x_views, const float *d_dout, int64_t dout_ld, int64_t B,
                                          int32_t F, int32_t D, const dlrm_feature_view *dx_views, int32_t relu_mask_feature0,
                                          float *d_dx0_max, dlrm_stream_t stream) {
    DLRM_REQUIRE(B >= 0 && F >= 1 && F <= kFP && D >= 8 && (D & (D - 1)) == 0 && F * D <= 4096,
                 "views need 1 <= F <= %d, dim a power of two >= 8, F*dim <= 4096; got F=%d dim=%d", kFP, F, D);
    if (B == 0) return DLRM_OK;
    DLRM_REQUIRE(d_dout != nullptr && dout_ld >= D + F * (F - 1) / 2, "dout is NULL or dout_ld too small");
    Views xv{}, dv{};
    if (int rc = check_views(x_views, F, D, "x", &xv)) return rc;
    if (int rc = check_views(dx_views, F, D, "dx", &dv)) return rc;
    const size_t lds = sizeof(float) * kIntWaves * ((size_t)F * (D + 4) + F * kSLd + 4) + 32 * kFP;
    const int64_t want = ceil_div(B, kIntWaves);
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    const int nv = (F * D / 4 + 63) / 64;
#define DLRM_INT_BWDV(NV)                                                                                              \
    case NV:                                                                                                           \
        DLRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(interact_dot_bwd_views_kernel<NV>),                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL(interact_dot_bwd_views_kernel<NV>, dim3(blocks), dim3(kIntThreads), lds, as_stream(stream), \
                           xv, dv, d_dout, dout_ld, B, F, D, relu_mask_feature0, d_dx0_max);                           \
        break;
    switch (nv) {
        DLRM_INT_BWDV(1) DLRM_INT_BWDV(2) DLRM_INT_BWDV(3) DLRM_INT_BWDV(4) DLRM_INT_BWDV(5) DLRM_INT_BWDV(6) DLRM_INT_BWDV(7)
        DLRM_INT_BWDV(8) DLRM_INT_BWDV(9) DLRM_INT_BWDV(10) DLRM_INT_BWDV(11) DLRM_INT_BWDV(12) DLRM_INT_BWDV(13)
        DLRM_INT_BWDV(14) DLRM_INT_BWDV(15) DLRM_INT_BWDV(16)
        default: return set_error(DLRM_ERR_INVALID, "F*D = %d is too large for the dot interaction kernel (max 4096)", F * D);
    }
#undef DLRM_INT_BWDV
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
