// float32 GEMMs of the bottom/top MLP on the gfx950 matrix cores with fused epilogues: forward
// (x W^T + b, ReLU), backward-data (dz W, masked by the previous layer's ReLU) and backward-weight
// (dz^T x, split over the batch dimension, partial slabs added in a fixed order).
//
// Operands and results are float32 in memory.  Two kernel templates evaluate the same product in three ways; the
// layer descriptor's gemm_mode picks one:
//   gemm_split_kernel<.., 3>      (DLRM_GEMM_BF16X3, the operators' default) every element is split exactly into
//                                 three bf16 pieces when its tile goes from registers to LDS; six cross terms on
//                                 v_mfma_f32_32x32x16_bf16 with float32 accumulation: float32 accuracy at 8/3 of
//                                 the float32 matrix rate.  Described in front of the kernel, further down.
//   gemm_split_kernel<.., 2>      (DLRM_GEMM_F16X2) two f16 pieces of every element times a power-of-two scale
//                                 chosen from the operand's largest magnitude, three cross terms on
//                                 v_mfma_f32_32x32x16_f16: half the matrix work, representation error about one
//                                 float32 ulp per product.  The maxima come in through the descriptor (epilogues
//                                 leave max |output| behind for the next layer) or from a pass of absmax_kernel.
//   gemm_f32_kernel               (DLRM_GEMM_F32_MFMA) v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit for bit
//                                 a k-ordered fmaf chain.  Described next.
//
// One kernel template covers the three products.  C[i][j] = sum_r A(i,r) B(j,r)
// where each operand is addressed either with the reduction index contiguous
// ("RC": element (i,r) at base + i*ld + r) or with the output index contiguous
// ("RS": element (i,r) at base + r*ld + i):
//     forward        : A = x  (RC), B = W (RC)       C = y  [M,N]
//     backward data  : A = dz (RC), B = W (RS)       C = dx [M,K]
//     backward weight: A = dz (RS), B = x (RS)       C = dW [N,K]   (reduction over M)
//
// gemm_f32_kernel.  128x128 output tile per 256-thread workgroup, 32-deep reduction
// steps, 2x2 waves each owning 64x64 (2x2 MFMA tiles of 32x32, 64 accumulator
// registers).  Global -> registers -> LDS with the next step's loads issued
// before the current step's MFMAs (double-buffered LDS, one barrier per step).
// LDS images: RC operands [128][32+4] (144-byte rows: conflict-free
// ds_read_b128 fragments, each lane gets 4 consecutive r; lanes 32..63 take the
// next 4), RS operands [32][128] (ds_read_b32 fragments).  MFMA j of a group
// of four consumes r = 8q+j on lanes 0..31 and r = 8q+4+j on lanes 32..63 for
// both operands, so the pairing of reduction indices is consistent.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "skinny.h"

namespace dlrm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RC_LD = BK + 4;              // floats per LDS row of an RC operand
constexpr int RC_TILE = BM * RC_LD;        // 4608 floats

constexpr int TILE_MAX = RC_TILE;
constexpr int kGemmThreads = 256;

struct GemmParams {
    const float *A;
    int64_t lda;
    const float *B;
    int64_t ldb;
    float *C;
    int64_t ldc;
    int M, N, K;          // C is [M,N], reduction length K
    const float *bias;    // [N] or NULL (added per output column)
    int relu;             // max(.,0) after the bias
    const float *mask;    // [M,N] (ld mask_ld) or NULL: C = (mask > 0) ? C : 0
    int64_t mask_ld;
    int a_vec, b_vec;     // operand rows are 16-byte aligned: 16-byte global loads allowed
    int gx, gy, gz;       // logical grid: column tiles, row tiles, reduction splits
    int k_per_split;      // reduction elements per split (multiple of BK)
    int64_t slab_stride;  // elements between the partial outputs of consecutive splits
    float *colsum;        // RS A operand only: [splits][M] sums of A over this split's reduction range
                          // (the bias gradient of the weight-gradient product), or NULL
    const float *a_max, *b_max;   // f16x2 evaluation: device scalars holding max |A|, max |B| (float32 bit patterns)
    float *c_max;         // or NULL: max |C| over everything this launch stores is folded in with atomicMax
    const unsigned char *bp_h, *bp_l;   // gemm_f16p_kernel: the f16 planes of B, [reduction step][N rounded up to 256][32] f16,
    int64_t bp_ld;                      // zero padded; bp_ld = bytes per reduction step
};

// Global -> register staging of one operand tile (4 x 16 bytes per thread).
// RC: tile rows are output indices, 32 reduction elements (128 B) each: 8 lanes per row.
// RS: tile rows are reduction indices, 128 output elements (512 B) each: 32 lanes per row.
//
// Both loaders are branch-free, so that all of a step's loads are in flight together and the
// compiler can leave them outstanding across the MFMA block (a guarded load makes hipcc wait
// for each one in turn):
//   load_tile_vec    16-byte loads.  Needs 16-byte aligned rows and a full 32-deep reduction
//                    step; RC tiles may run past the last output row (rows are clamped), RS
//                    tiles must be full in the output index.
//   load_tile_scalar 4-byte loads from clamped addresses + zero select: any shape / alignment.
template <bool RC>
__device__ inline void load_tile_vec(const float *__restrict__ base, int64_t ld, int o0, int o_end, int r0,
                                     float4 (&reg)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if constexpr (RC) {
            const int o = o0 + (tid >> 3) + 32 * u;
            const int oc = o < o_end ? o : o_end - 1;
            // rows past the end are read from the last row: they only feed output rows that
            // the epilogue never stores, so no zeroing is needed
            reg[u] = *reinterpret_cast<const float4 *>(base + (int64_t)oc * ld + r0 + (tid & 7) * 4);
        } else {
            reg[u] = *reinterpret_cast<const float4 *>(base + (int64_t)(r0 + (tid >> 5) + 8 * u) * ld + o0 +
                                                       (tid & 31) * 4);
        }
    }
}

template <bool RC>
__device__ inline void load_tile_scalar(const float *__restrict__ base, int64_t ld, int o0, int o_end, int r0,
                                        int r_end, float4 (&reg)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int o, r;
        if constexpr (RC) { o = o0 + (tid >> 3) + 32 * u; r = r0 + (tid & 7) * 4; }
        else { r = r0 + (tid >> 5) + 8 * u; o = o0 + (tid & 31) * 4; }
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int oo = RC ? o : o + q, rr = RC ? r + q : r;
            const bool ok = oo < o_end && rr < r_end;
            const int oc = oo < o_end ? oo : o_end - 1, rc = rr < r_end ? rr : r_end - 1;
            const float x = RC ? base[(int64_t)oc * ld + rc] : base[(int64_t)rc * ld + oc];
            e[q] = ok ? x : 0.f;
        }
        reg[u] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

template <bool RC>
__device__ inline void store_tile(float *__restrict__ s, const float4 (&reg)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int off;
        if constexpr (RC) off = ((tid >> 3) + 32 * u) * RC_LD + (tid & 7) * 4;
        else off = ((tid >> 5) + 8 * u) * BM + (tid & 31) * 4;
        *reinterpret_cast<float4 *>(s + off) = reg[u];
    }
}

// fragment of 4 reduction elements (one per MFMA of a group of four) for the
// 32 output indices starting at o_base, reduction chunk q (8 elements)
template <bool RC>
__device__ inline void load_frag(const float *__restrict__ s, int o_base, int q, float (&f)[4]) {
    const int lane = threadIdx.x & 63;
    const int o = o_base + (lane & 31), h = lane >> 5;
    if constexpr (RC) {
        const float4 v = *reinterpret_cast<const float4 *>(s + o * RC_LD + q * 8 + h * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = s[(q * 8 + h * 4 + j) * BM + o];
    }
}

// Epilogue.  Register r of lane l holds C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of its 32x32 tile: a row
// per register, so direct stores would be 4 bytes per lane.  Each wave instead parks 32 rows of its
// (32 MI) x (32 NI) block at a time in LDS (`ct`: 4 waves x 32 x (32 NI + 4) floats, the staging buffers are
// free by now) and streams them out as 16-byte row segments -- 8 NI lanes per row piece -- with
// bias / ReLU / mask fused.
template <int MI, int NI, int WC = 2>     // (WC: waves side by side along the columns of the workgroup's tile)
__device__ inline void store_output(const GemmParams &p, const f32x16 (&acc)[MI][NI], float *ct_all, int m0, int n0,
                                    int bz) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = wave / WC, wc = wave % WC;
    float *C = p.C + (int64_t)bz * p.slab_stride;
    constexpr int CT_LD = 32 * NI + 4;
    constexpr int LPR = 8 * NI;           // lanes per row piece
    constexpr int RPI = 64 / LPR;         // rows per iteration
    float *ct = ct_all + wave * (32 * CT_LD);
    const int cn = n0 + wc * (32 * NI) + (lane % LPR) * 4;
    const bool c_vec = (reinterpret_cast<uintptr_t>(C) % 16 == 0) && (p.ldc % 4 == 0) &&
                       (p.mask == nullptr || ((reinterpret_cast<uintptr_t>(p.mask) % 16 == 0) && (p.mask_ld % 4 == 0)));
    float vmax = 0.f;     // largest magnitude this lane stores (for c_max)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
        if (cn < p.N) bv.x = p.bias[cn];
        if (cn + 1 < p.N) bv.y = p.bias[cn + 1];
        if (cn + 2 < p.N) bv.z = p.bias[cn + 2];
        if (cn + 3 < p.N) bv.w = p.bias[cn + 3];
    }
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        // the mask pieces of this row block are requested before the block is parked, so that they travel while the
        // accumulators go through LDS (one round trip to memory per block instead of one per group of rows)
        float4 mk[32 / RPI];
        const bool mask_vec = p.mask != nullptr && c_vec && cn + 3 < p.N;
        if (mask_vec) {
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int m = m0 + wr * (32 * MI) + mi * 32 + it * RPI + (lane / LPR);
                mk[it] = *reinterpret_cast<const float4 *>(p.mask + (int64_t)(m < p.M ? m : p.M - 1) * p.mask_ld + cn);
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                ct[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CT_LD + ni * 32 + (lane & 31)] = acc[mi][ni][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; it++) {
            const int row = it * RPI + (lane / LPR);
            const int m = m0 + wr * (32 * MI) + mi * 32 + row;
            float4 v = *reinterpret_cast<const float4 *>(ct + row * CT_LD + (lane % LPR) * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (m >= p.M || cn >= p.N) continue;
            float *dst = C + (int64_t)m * p.ldc + cn;
            if (c_vec && cn + 3 < p.N) {
                if (mask_vec) {
                    const float4 k = mk[it];
                    v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f;
                    v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(dst) = v;
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (cn + q < p.N) {
                        float x = e[q];
                        if (p.mask) x = p.mask[(int64_t)m * p.mask_ld + cn + q] > 0.f ? x : 0.f;
                        dst[q] = x;
                        vmax = fmaxf(vmax, fabsf(x));
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if (p.c_max != nullptr) fold_absmax(vmax, p.c_max);   // (non-negative floats order like their bit patterns)
}

// 64 MFMAs: one 32-deep reduction step of the wave's 64x64 output
template <bool A_RC, bool B_RC>
__device__ inline void mfma_step(const float *__restrict__ sa, const float *__restrict__ sb, int wr, int wc,
                                 f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float fa[2][4], fb[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; mi++) load_frag<A_RC>(sa, wr * 64 + mi * 32, q, fa[mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ni++) load_frag<B_RC>(sb, wc * 64 + ni * 32, q, fb[ni]);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][j], fb[ni][j], acc[mi][ni], 0, 0, 0);
    }
}

template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f32_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][TILE_MAX];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remap the linear
    // id so that every XCD works through one contiguous run of tiles (column tile fastest):
    // the tiles that share an operand row-block are then resident on the same L2 together.
    const int nwg = p.gx * p.gy * p.gz;
    const int lin = blockIdx.x, xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int bx = tile % p.gx, by = (tile / p.gx) % p.gy, bz = tile / (p.gx * p.gy);
    const int m0 = by * BM, n0 = bx * BN;
    const int k_begin = bz * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int steps = (k_end - k_begin + BK - 1) / BK;
    const int full_steps = (k_end - k_begin) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.f;

    // can this workgroup's tiles take the 16-byte loader on full reduction steps?
    // (an RS tile may run past the last output index as long as it stays inside the row's
    //  leading dimension: the surplus columns only feed outputs that are never stored)
    const bool vec_tile = p.a_vec && p.b_vec && (A_RC || m0 + BM <= p.M || m0 + BM <= p.lda) &&
                          (B_RC || n0 + BN <= p.N || n0 + BN <= p.ldb);

    float4 ra[4], rb[4];
    if (steps > 0) {
        if (vec_tile && full_steps > 0) {
            load_tile_vec<A_RC>(p.A, p.lda, m0, p.M, k_begin, ra);
            load_tile_vec<B_RC>(p.B, p.ldb, n0, p.N, k_begin, rb);
        } else {
            load_tile_scalar<A_RC>(p.A, p.lda, m0, p.M, k_begin, k_end, ra);
            load_tile_scalar<B_RC>(p.B, p.ldb, n0, p.N, k_begin, k_end, rb);
        }
        store_tile<A_RC>(smem[0][0], ra);
        store_tile<B_RC>(smem[0][1], rb);
    }
    __syncthreads();

    // bias gradient: column sums of dz fall out of the A tiles already staged in LDS; the
    // workgroups of the first column tile accumulate them, one output row per thread
    float csum = 0.f;
    const bool do_colsum = !A_RC && p.colsum != nullptr && bx == 0 && tid < BM;
    auto add_colsum = [&](const float *sa) {
        if constexpr (!A_RC) {
            if (do_colsum) {
#pragma unroll 8
                for (int r = 0; r < BK; r++) csum += sa[r * BM + tid];
            }
        }
    };

    int st = 0;
    if (vec_tile) {
        // steady state: the next step is a full one; straight-line code, loads stay in flight
        // behind the 64 MFMAs of the current step
        for (; st + 1 < full_steps; st++) {
            const int cur = st & 1;
            const int r0 = k_begin + (st + 1) * BK;
            load_tile_vec<A_RC>(p.A, p.lda, m0, p.M, r0, ra);
            load_tile_vec<B_RC>(p.B, p.ldb, n0, p.N, r0, rb);
            __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMA block
            add_colsum(smem[cur][0]);
            mfma_step<A_RC, B_RC>(smem[cur][0], smem[cur][1], wr, wc, acc);
            __builtin_amdgcn_sched_barrier(0);
            store_tile<A_RC>(smem[cur ^ 1][0], ra);
            store_tile<B_RC>(smem[cur ^ 1][1], rb);
            __syncthreads();
        }
    }
    // edge tiles, unaligned operands and the (partial) last steps
    for (; st < steps; st++) {
        const int cur = st & 1;
        const bool more = st + 1 < steps;
        if (more) {
            const int r0 = k_begin + (st + 1) * BK;
            load_tile_scalar<A_RC>(p.A, p.lda, m0, p.M, r0, k_end, ra);
            load_tile_scalar<B_RC>(p.B, p.ldb, n0, p.N, r0, k_end, rb);
        }
        add_colsum(smem[cur][0]);
        mfma_step<A_RC, B_RC>(smem[cur][0], smem[cur][1], wr, wc, acc);
        if (more) {
            store_tile<A_RC>(smem[cur ^ 1][0], ra);
            store_tile<B_RC>(smem[cur ^ 1][1], rb);
        }
        __syncthreads();
    }

    if (do_colsum && m0 + tid < p.M) p.colsum[(int64_t)bz * p.M + m0 + tid] = csum;

    __syncthreads();   // the staging buffers are free now
    store_output<2, 2>(p, acc, reinterpret_cast<float *>(&smem[0][0][0]), m0, n0, bz);
}

// ---------------------------------------------------------------------------------------------
// Same product, faster matrix pipe: every float32 operand element is split exactly into three bf16
// pieces a = a0 + a1 + a2 (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1); the pieces
// together carry 24+ mantissa bits, |a - (a0+a1+a2)| <= 2^-27 |a|) and the product is evaluated as
// the six leading cross terms a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 on v_mfma_f32_32x32x16_bf16
// with float32 accumulation.  Products of bf16 pieces are exact in float32 and the dropped terms are
// below 2^-26 |ab|, so the result carries float32 accuracy (measured against a float64 reference it
// is as close as the float32 MFMA path) at 6 x 32 cycles per 16-deep step instead of 8 x 64.
//
// Pipeline.  Global tiles are R x 32 float32 per operand (R = the workgroup tile's extent on that
// operand's output index, 128 or 64), loaded into registers two half-steps
// before their first use.  LDS holds two stages of one 16-deep half-step each (three bf16 planes
// per operand); while the MFMAs of half-step j (24 for a 128 x 128 tile) run out of stage j&1, the same wave splits the
// next half-step's 16 elements per lane and writes them to the other stage, so the split's VALU
// work and LDS writes sit in the shadow of the MFMAs and one barrier per half-step is enough.
//   RC planes: [R][16] bf16, 32-byte rows whose two 16-byte chunks swap places in every other
//              group of 8 rows (conflict-free ds_write_b64 of the pieces and ds_read_b128 fragments
//              of 8 k-values)
//   RS planes: [16][R] bf16, 2R-byte rows with the 16-byte chunks XOR-swizzled; fragments come
//              out of the hardware transposing read ds_read_b64_tr_b16 (two per fragment)
// 48 KB of LDS per workgroup at 128 x 128 (12 KB per 64 of R and operand); the register budget allows two
// workgroups per CU.
//
// The same pipeline with NP = 2 planes per operand carries the f16x2 evaluation (common.h: split2_pair): 12 MFMAs per
// half-step of a 128 x 128 tile instead of 24, 32 KB of LDS for the stages; the kernel is then bound by the LDS
// traffic and the split arithmetic rather than by the matrix pipe.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Workgroup tiles.  TM x TN outputs per 256-thread workgroup, TM, TN in {128, 64}; the four waves sit 2 x 2
// and own (TM/2) x (TN/2) each, i.e. MI x NI = (TM/64) x (TN/64) MFMA tiles of 32 x 32.  128 x 128 has the best
// ratio of matrix work to split / LDS work and is used whenever it yields enough workgroups to fill the chip;
// the smaller tiles exist for small batches (a local batch of 8192 over 8 GPUs, Criteo-Kaggle's 2048), where
// 128 x 128 leaves most compute units without a workgroup.
//
// An operand tile of R (= TM or TN) output indices x 32 reduction elements is loaded as two half-steps of
// R/64 16-byte pieces per thread each.
constexpr int plane_bytes(int R) { return R * 32; }   // R x 16 bf16

__device__ inline int rc_chunk_off(int row, int ch) {   // byte offset of 16-byte chunk ch (0/1) of row `row`
    return 32 * row + 16 * (ch ^ ((row >> 3) & 1));
}

// RS planes: [16][R] bf16.  The transposing read takes, per 32 lanes, four rows x 64 bytes; the swizzle puts
// those four pieces on four different 64-byte bank ranges (R = 128: rows are 256 bytes, a multiple of the bank
// window, so the row's low bits pick the range; R = 64: rows are 128 bytes, rows r and r+2 would collide).
template <int R>
__device__ inline int rs_chunk_off(int row, int ch) {   // byte offset of 16-byte chunk ch of row `row`
    if constexpr (R == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    else return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)));
}

// Register image of one R x 32 operand tile: reg[h*P + v] (P = R/64 pieces) belongs to half-step h.
//   RC: piece v = row (tid>>2) + 64v, reduction elements 16h + 4(tid&3) .. +3
//   RS: piece v = reduction row 16h + tid/(R/4) + (1024/R) v, output elements 4(tid % (R/4)) .. +3
// Addresses are a wave-uniform tile origin (scalar registers) plus a per-thread byte offset that
// does not change over the reduction loop.
template <bool RC, int R>
struct TileAddr {
    const float *origin;   // element (o0, 0) of the operand
    int64_t ld;
    int o0, o_end;
    unsigned off[R / 64];  // byte offsets of this thread's 16-byte pieces inside a half tile
};

template <bool RC, int R>
__device__ inline TileAddr<RC, R> make_tile_addr(const float *base, int64_t ld, int o0, int o_end) {
    TileAddr<RC, R> t;
    const int tid = threadIdx.x;
    constexpr int LPR = R / 4;   // RS: lanes per reduction row
    t.ld = ld; t.o0 = o0; t.o_end = o_end;
    if constexpr (RC) {
        t.origin = base + (int64_t)o0 * ld;
#pragma unroll
        for (int v = 0; v < R / 64; v++) {
            const int o = o0 + (tid >> 2) + 64 * v;
            const int oc = o < o_end ? o : o_end - 1;   // surplus rows only feed outputs that are never stored
            t.off[v] = (unsigned)((oc - o0) * (int)ld + (tid & 3) * 4) * 4u;
        }
    } else {
        t.origin = base + o0;
        // a piece that would start beyond the row's leading dimension re-reads the row's last piece: like the columns
        // between o_end and ld it only feeds outputs that are never stored
        int col = o0 + (tid % LPR) * 4;
        if (col + 4 > (int)ld) col = (int)ld - 4;
#pragma unroll
        for (int v = 0; v < R / 64; v++)
            t.off[v] = (unsigned)((tid / LPR + (256 / LPR) * v) * (int)ld + (col - o0)) * 4u;
    }
    return t;
}

// 16-byte loads of half h of the tile whose reduction range starts at r0 (needs 16-byte aligned rows
// and a full 32-deep tile; an RS tile must not run past the row's leading dimension)
template <bool RC, int R>
__device__ inline void load_half_vec(const TileAddr<RC, R> &t, int r0, int h, f32x4 *x) {
    // buffer loads: the wave-uniform half-tile origin travels in a scalar resource descriptor and each
    // piece costs one 32-bit offset register (no range check is wanted: the record count is the maximum)
    const float *o = RC ? t.origin + r0 + 16 * h : t.origin + (int64_t)(r0 + 16 * h) * t.ld;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(o), 0, 0xffffffff, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int v = 0; v < R / 64; v++)
        x[v] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)t.off[v], 0, 0));
}

// any shape / alignment: 4-byte loads from clamped addresses, zero beyond the operand's extent
template <bool RC, int R>
__device__ inline void load_half_scalar(const TileAddr<RC, R> &t, int r0, int r_end, int h, f32x4 *xo) {
    const int tid = threadIdx.x;
    constexpr int LPR = R / 4;
    const float *base = RC ? t.origin - (int64_t)t.o0 * t.ld : t.origin - t.o0;
#pragma unroll
    for (int v = 0; v < R / 64; v++) {
        int o, r;
        if constexpr (RC) { o = t.o0 + (tid >> 2) + 64 * v; r = r0 + 16 * h + (tid & 3) * 4; }
        else { r = r0 + 16 * h + tid / LPR + (256 / LPR) * v; o = t.o0 + (tid % LPR) * 4; }
        f32x4 x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int oo = RC ? o : o + q, rr = RC ? r + q : r;
            const bool ok = oo < t.o_end && rr < r_end;
            const int oc = oo < t.o_end ? oo : t.o_end - 1, rc = rr < r_end ? rr : r_end - 1;
            const float e = RC ? base[(int64_t)oc * t.ld + rc] : base[(int64_t)rc * t.ld + oc];
            x[q] = ok ? e : 0.f;
        }
        xo[v] = x;
    }
}

// split the registers of one half-step and write the pieces to the NP planes at `planes` (NP = 3: bf16 pieces,
// NP = 2: f16 pieces of the values times `scale`)
template <bool RC, int R, int NP>
__device__ inline void split_store_half(unsigned char *__restrict__ planes, const f32x4 *xs, float scale) {
    const int tid = threadIdx.x;
    constexpr int LPR = R / 4;
#pragma unroll
    for (int v = 0; v < R / 64; v++) {
        const f32x4 x = xs[v];
        int off;
        if constexpr (RC) off = rc_chunk_off((tid >> 2) + 64 * v, (tid & 3) >> 1) + 8 * (tid & 1);
        else off = rs_chunk_off<R>(tid / LPR + (256 / LPR) * v, (tid % LPR) >> 1) + 8 * (tid & 1);
        if constexpr (NP == 3) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pair(f32x2{x[0], x[1]}, h0, m0, l0);
            split3_pair(f32x2{x[2], x[3]}, h1, m1, l1);
            *reinterpret_cast<u32x2 *>(planes + off) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(planes + plane_bytes(R) + off) = u32x2{m0, m1};
            *reinterpret_cast<u32x2 *>(planes + 2 * plane_bytes(R) + off) = u32x2{l0, l1};
        } else {
            unsigned h0, l0, h1, l1;
            split2_pair(f32x2{x[0], x[1]}, scale, h0, l0);
            split2_pair(f32x2{x[2], x[3]}, scale, h1, l1);
            *reinterpret_cast<u32x2 *>(planes + off) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(planes + plane_bytes(R) + off) = u32x2{l0, l1};
        }
    }
}

// 8 consecutive reduction elements (lane half h takes 8h..8h+7 of the 16-deep half-step) of output
// index o_base + (lane & 31), from one plane
template <bool RC, int R>
__device__ inline bf16x8 load_frag16(const unsigned char *__restrict__ plane, int o_base) {
    const int lane = threadIdx.x & 63;
    if constexpr (RC) {
        return *reinterpret_cast<const bf16x8 *>(plane + rc_chunk_off(o_base + (lane & 31), lane >> 5));
    } else {
        // per 16-lane group the transposing read takes a 4-row x 16-column block; lane 4q+p of the
        // group supplies the address of row q, columns 4p..4p+3, and lane i receives column i
        const int li = lane & 15, g = lane >> 4, h = g >> 1;
        const int q = li >> 2, pp = li & 3;
        const int ch = (o_base >> 3) + 2 * (g & 1) + (pp >> 1);
        const int row0 = 8 * h + q;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4 *)(plane + rs_chunk_off<R>(row0, ch) + 8 * (pp & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4 *)(plane + rs_chunk_off<R>(row0 + 4, ch) + 8 * (pp & 1)));
        const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, t);
    }
}

// Compiler fence for register values: uses of the registers after this point cannot be moved above
// it.  Keeps the bf16 split of a prefetched tile in the half-step it belongs to (it is plain
// arithmetic on loaded registers, which the compiler would otherwise start right behind the loads,
// waiting for them with whole MFMA blocks still to come).
__device__ inline void pin_here(f32x4 &a) { asm volatile("" : "+v"(a)); }

// the fragments of one 16-deep half-step of the wave's output block, out of one stage
template <int MI, int NI, int NP>
struct Frags {
    bf16x8 a[MI][NP], b[NI][NP];   // (raw 16-bit pieces: bf16 for NP = 3, f16 for NP = 2)
};

// piece of A and piece of B that meet in cross term t, in the order mfma_frags issues the terms (smallest first):
// three bf16 pieces -> six terms, two f16 pieces -> three terms
constexpr int kTermA3[6] = {2, 1, 0, 1, 0, 0};
constexpr int kTermB3[6] = {0, 1, 2, 0, 1, 0};
constexpr int kTermA2[3] = {1, 0, 0};
constexpr int kTermB2[3] = {0, 1, 0};
template <int NP> constexpr int n_terms() { return NP == 3 ? 6 : 3; }
template <int NP> constexpr int term_a(int t) { return NP == 3 ? kTermA3[t] : kTermA2[t]; }
template <int NP> constexpr int term_b(int t) { return NP == 3 ? kTermB3[t] : kTermB2[t]; }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// The reads are issued in the order the MFMAs first need them: the first MFMA waits for two fragments, not for
// all of them (LDS reads return in order, so the wait before a use covers every read issued ahead of it).
template <bool A_RC, bool B_RC, int TM, int TN, int NP>
__device__ inline void load_frags(const unsigned char *__restrict__ stage, int wr, int wc, Frags<TM / 64, TN / 64, NP> &f) {
    constexpr int MI = TM / 64, NI = TN / 64;
    const unsigned char *pa = stage, *pb = stage + NP * plane_bytes(TM);
    bool have_a[MI][NP] = {}, have_b[NI][NP] = {};
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int t = 0; t < n_terms<NP>(); t++) {
                const int x = term_a<NP>(t), y = term_b<NP>(t);
                if (!have_a[mi][x]) {
                    f.a[mi][x] = load_frag16<A_RC, TM>(pa + x * plane_bytes(TM), wr * (TM / 2) + mi * 32);
                    have_a[mi][x] = true;
                }
                if (!have_b[ni][y]) {
                    f.b[ni][y] = load_frag16<B_RC, TN>(pb + y * plane_bytes(TN), wc * (TN / 2) + ni * 32);
                    have_b[ni][y] = true;
                }
            }
}

// the cross terms for each of the wave's 32x32 accumulators, smallest terms first.  Two f16 pieces: the low pieces are
// stored 2^11 times too large (common.h: split2_pair), so the two mixed terms go to their own accumulator `lo`, which
// the epilogue adds times 2^-11.
template <int MI, int NI, int NP>
__device__ inline void mfma_frags(const Frags<MI, NI, NP> &f, f32x16 (&acc)[MI][NI], f32x16 (&lo)[MI][NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            if constexpr (NP == 3) {
                f32x16 c = acc[mi][ni];
#pragma unroll
                for (int t = 0; t < n_terms<NP>(); t++)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mi][term_a<NP>(t)], f.b[ni][term_b<NP>(t)], c, 0, 0, 0);
                acc[mi][ni] = c;
            } else {
                const f16x8 ah = __builtin_bit_cast(f16x8, f.a[mi][0]), al = __builtin_bit_cast(f16x8, f.a[mi][1]);
                const f16x8 bh = __builtin_bit_cast(f16x8, f.b[ni][0]), bl = __builtin_bit_cast(f16x8, f.b[ni][1]);
                f32x16 c = lo[mi][ni];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
                lo[mi][ni] = c;
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mi][ni], 0, 0, 0);
            }
        }
}

// Scheduling recipe for one half-step region (3 (MI+NI) fragment reads, 6 MI NI MFMAs, the split of
// 4 (MI+NI) elements per lane -- 18 vector operations per piece -- and its LDS writes): all reads first, then
// the split operations spread evenly behind the MFMAs, the writes of an operand once its pieces are complete.
// (128 x 128: three split operations behind every MFMA.)
template <int I, int MI, int NI, int NP>
__device__ inline void interleave_from() {
    constexpr int kMfma = n_terms<NP>() * MI * NI, kValu = (NP == 3 ? 18 : 20) * (MI + NI), kWrites = (NP * (MI + NI)) / 4;
    if constexpr (I < kMfma) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (kValu * (I + 1)) / kMfma - (kValu * I) / kMfma, 0);
        if constexpr (I == kMfma / 2 - 1 || I == kMfma - 1) __builtin_amdgcn_sched_group_barrier(0x200, kWrites, 0);
        interleave_from<I + 1, MI, NI, NP>();
    }
}

template <int MI, int NI, int NP>
__device__ inline void interleave_half() {
    __builtin_amdgcn_sched_group_barrier(0x100, NP * (MI + NI), 0);
    interleave_from<0, MI, NI, NP>();
}

template <bool A_RC, bool B_RC, int TM, int TN, int NP>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_split_kernel(GemmParams p) {
    constexpr int MI = TM / 64, NI = TN / 64, PA = TM / 64, PB = TN / 64;
    constexpr int kStage = NP * plane_bytes(TM) + NP * plane_bytes(TN);   // A planes then B planes
    constexpr int kOutStaging = 4 * 32 * (32 * NI + 4) * 4;               // the epilogue parks accumulators here
    constexpr int kStages = 2 * kStage > kOutStaging ? 2 * kStage : kOutStaging;
    constexpr int kColsum = A_RC ? 0 : kGemmThreads * 16;                   // running bias-gradient sums, one f32x4 per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[kStages + kColsum];
    unsigned char *const s0 = smem, *const s1 = smem + kStage;
    // f16 pieces: power-of-two scales from the operands' largest magnitudes (wave-uniform scalars)
    float scale_a = 1.f, scale_b = 1.f;
    if constexpr (NP == 2) {
        scale_a = split2_scale(__float_as_uint(*p.a_max));
        scale_b = split2_scale(__float_as_uint(*p.b_max));
    }

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int nwg = p.gx * p.gy * p.gz;
    const int lin = blockIdx.x, xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int bx = tile % p.gx, by = (tile / p.gx) % p.gy, bz = tile / (p.gx * p.gy);
    const int m0 = by * TM, n0 = bx * TN;
    const int k_begin = bz * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int steps = (k_end - k_begin + BK - 1) / BK;
    const int full_steps = (k_end - k_begin) / BK;

    f32x16 acc[MI][NI], acc_lo[MI][NI];   // (acc_lo: the mixed terms of the f16 pieces; unused with bf16 pieces)
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[mi][ni][r] = 0.f; acc_lo[mi][ni][r] = 0.f; }

    // (a_vec / b_vec: 16-byte aligned rows, so the leading dimensions are multiples of 4 elements; edge tiles of a
    // reduction-strided operand read whole pieces up to the leading dimension, see make_tile_addr)
    const bool vec_tile = p.a_vec && p.b_vec && (A_RC || p.lda >= 4) && (B_RC || p.ldb >= 4);

    // bias gradient (weight-gradient product only): this thread's four output rows of the A tile,
    // summed over every staged reduction row
    // (the running sums live in LDS, one slot per thread that only this thread touches: four registers fewer in a
    //  loop that is at the register limit with the two accumulator sets of the f16 pieces)
    const bool do_colsum = !A_RC && p.colsum != nullptr && bx == 0;
    f32x4 *const cs_slot = reinterpret_cast<f32x4 *>(smem + kStages) + threadIdx.x;
    if constexpr (!A_RC) *cs_slot = f32x4{0.f, 0.f, 0.f, 0.f};
    // split half-step h of a raw tile pair into `stage` (`live` false: the registers hold no new
    // tile and the stage is never read; only the bias-gradient sum must skip them)
    auto stage_half = [&](unsigned char *stage, f32x4 (&xa)[2 * PA], f32x4 (&xb)[2 * PB], int h, bool live) {
        if constexpr (!A_RC) {
            // (no branch: control flow inside a half-step would cut the region the scheduling recipe describes)
            f32x4 add = xa[h * PA];
#pragma unroll
            for (int v = 1; v < PA; v++) add += xa[h * PA + v];
            *cs_slot += (do_colsum && live) ? add : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        split_store_half<A_RC, TM, NP>(stage, &xa[h * PA], scale_a);
        split_store_half<B_RC, TN, NP>(stage + NP * plane_bytes(TM), &xb[h * PB], scale_b);
    };

    // Tiles [t_begin, t_end) of the reduction through the two-stage pipeline.  VEC: 16-byte loads
    // (full 32-deep tiles of aligned operands), else clamped 4-byte loads with zero fill.
    const TileAddr<A_RC, TM> ta = make_tile_addr<A_RC, TM>(p.A, p.lda, m0, p.M);
    const TileAddr<B_RC, TN> tb = make_tile_addr<B_RC, TN>(p.B, p.ldb, n0, p.N);
    f32x4 xa[2 * PA], xb[2 * PB], ya[2 * PA], yb[2 * PB];
    // zero the elements of half h of a raw RC tile whose reduction index is k_end or beyond (tile at r0)
    auto mask_tail = [&](f32x4 *x, int pieces, int r0, int h) {
        const int r = r0 + 16 * h + (tid & 3) * 4;
        for (int v = 0; v < pieces; v++)
#pragma unroll
            for (int q = 0; q < 4; q++) x[v][q] = r + q < k_end ? x[v][q] : 0.f;
    };
    auto run = [&](auto vec_tag, auto mask_tag, int t_begin, int t_end) {
        constexpr bool VEC = decltype(vec_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value;   // 16-byte loads of a partial last tile, surplus elements zeroed
        // request half h of tile tt (clamped: surplus requests re-read the last tile)
        auto load = [&](int tt, int h, f32x4 (&ra)[2 * PA], f32x4 (&rb)[2 * PB]) {
            const int tc = tt < t_end ? tt : t_end - 1;
            const int r0 = k_begin + tc * BK;
            if constexpr (VEC) {
                load_half_vec<A_RC, TM>(ta, r0, h, &ra[h * PA]);
                load_half_vec<B_RC, TN>(tb, r0, h, &rb[h * PB]);
            } else {
                load_half_scalar<A_RC, TM>(ta, r0, k_end, h, &ra[h * PA]);
                load_half_scalar<B_RC, TN>(tb, r0, k_end, h, &rb[h * PB]);
            }
        };
        auto pin = [&](f32x4 (&ra)[2 * PA], f32x4 (&rb)[2 * PB], int h) {
#pragma unroll
            for (int v = 0; v < PA; v++) pin_here(ra[h * PA + v]);
#pragma unroll
            for (int v = 0; v < PB; v++) pin_here(rb[h * PB + v]);
        };
        // One tile = two half-steps.  On entry stage 0 holds half 0 of tile tt, (ua, ub) hold the raw
        // tile tt and (va, vb) half 0 of tile tt+1 (in flight or landed).  Every half-step requests the
        // half tile that is staged two half-steps later, into the registers staged one half-step ago.
        auto tile_step = [&](int tt, f32x4 (&ua)[2 * PA], f32x4 (&ub)[2 * PB], f32x4 (&va)[2 * PA], f32x4 (&vb)[2 * PB]) {
            Frags<MI, NI, NP> f;
            load(tt + 1, 1, va, vb);
            __builtin_amdgcn_sched_barrier(0);   // the requests go out ahead of the MFMA block
            load_frags<A_RC, B_RC, TM, TN, NP>(s0, wr, wc, f);   // reads first: the writes below go to the other stage
            pin(ua, ub, 1);
            if constexpr (MASK) { mask_tail(&ua[PA], PA, k_begin + tt * BK, 1); mask_tail(&ub[PB], PB, k_begin + tt * BK, 1); }
            stage_half(s1, ua, ub, 1, true);
            mfma_frags<MI, NI, NP>(f, acc, acc_lo);
            interleave_half<MI, NI, NP>();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            load(tt + 2, 0, ua, ub);
            __builtin_amdgcn_sched_barrier(0);
            load_frags<A_RC, B_RC, TM, TN, NP>(s1, wr, wc, f);
            pin(va, vb, 0);
            if constexpr (MASK) {
                const int tn = tt + 1 < t_end ? tt + 1 : t_end - 1;
                mask_tail(&va[0], PA, k_begin + tn * BK, 0); mask_tail(&vb[0], PB, k_begin + tn * BK, 0);
            }
            stage_half(s0, va, vb, 0, tt + 1 < t_end);
            mfma_frags<MI, NI, NP>(f, acc, acc_lo);
            interleave_half<MI, NI, NP>();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        load(t_begin, 0, xa, xb);
        load(t_begin, 1, xa, xb);
        load(t_begin + 1, 0, ya, yb);
        if constexpr (MASK) { mask_tail(&xa[0], PA, k_begin + t_begin * BK, 0); mask_tail(&xb[0], PB, k_begin + t_begin * BK, 0); }
        stage_half(s0, xa, xb, 0, true);
        __syncthreads();
        for (int t = t_begin; t < t_end; t += 2) {
            tile_step(t, xa, xb, ya, yb);
            if (t + 1 >= t_end) break;
            tile_step(t + 1, ya, yb, xa, xb);
        }
    };
    int t_done = 0;
    if (vec_tile && full_steps > 0) {
        run(std::true_type{}, std::false_type{}, 0, full_steps);
        t_done = full_steps;
    }
    if constexpr (A_RC && B_RC) {
        // partial last tile of reduction-contiguous operands whose rows are long enough to be read past k_end
        // (padded leading dimensions): 16-byte loads, the surplus elements zeroed in registers
        const int64_t r_hi = (int64_t)k_begin + (int64_t)steps * BK;
        if (t_done < steps && vec_tile && r_hi <= p.lda && r_hi <= p.ldb) {
            run(std::true_type{}, std::true_type{}, t_done, steps);
            t_done = steps;
        }
    }
    if (t_done < steps) run(std::false_type{}, std::false_type{}, t_done, steps);   // edge tiles, unaligned operands

    if constexpr (NP == 2) {      // mixed terms in, operand scales out (two exact multiplications: either factor alone stays in range)
        const float inv_a = 1.0f / scale_a, inv_b = 1.0f / scale_b;
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[mi][ni][r] = (__builtin_fmaf(acc_lo[mi][ni][r], kSplit2LowInv, acc[mi][ni][r]) * inv_a) * inv_b;
    }
    float *fsm = reinterpret_cast<float *>(smem);
    if constexpr (!A_RC) {
        if (do_colsum) {
            // thread (tid / (TM/4), tid % (TM/4)) holds the sums of output rows 4 (tid % (TM/4)).. over its reduction rows
            constexpr int LPR = TM / 4, GROUPS = 256 / LPR;
            const f32x4 cs = *cs_slot;
            *reinterpret_cast<f32x4 *>(fsm + (tid / LPR) * TM + (tid % LPR) * 4) = cs;
            __syncthreads();
            if (tid < TM && m0 + tid < p.M) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < GROUPS; w++) a += fsm[w * TM + tid];
                p.colsum[(int64_t)bz * p.M + m0 + tid] = a;
            }
            __syncthreads();
        }
    }
    store_output<MI, NI>(p, acc, fsm, m0, n0, bz);
}

// ---------------------------------------------------------------------------------------------
// f16x2 evaluation of C = A B^T with B cut into its two f16 planes BEFOREHAND (dlrm_split_weights: B is a weight
// matrix, a few MB that every one of the hundreds of row tiles would otherwise split again), for the forward
// (A = x, B = W) and backward-data (A = dz, B = W^T, planes of the transposed matrix) products of the big layers.
//
// 128 x 128 outputs per 256-thread workgroup, 32-deep reduction steps.  Wave w owns rows 32w .. 32w+31 and all 128
// columns (4 MFMA tiles of 32 x 32, two accumulators each: see split2_pair).  Nobody but that wave needs its A rows:
// they arrive as raw float32 in a wave-private piece of LDS (direct-to-LDS loads, coalesced), lane (row r, half h)
// reads ITS 16 reduction elements of the step -- 64 bytes of row r -- back, splits them in registers and uses the
// pieces as its MFMA A fragments (first 8 elements in the first of the step's two 16-deep MFMA slices, the other 8 in
// the second: the reduction index of MFMA position (h, j) of slice s is 16 h + 8 s + j, and the B planes are read in
// the same order).  So A waits for no barrier and every A element is split by exactly one lane; only the B planes
// (16 KB per step: 128 x 32 f16, two planes) are shared, in a two-stage ring with one barrier per step.  No operand
// passes through registers on its way to LDS.  Against gemm_split_kernel<.., 2>: half the split arithmetic, no LDS
// write instructions, half the barriers.  (Reading A straight from global memory into the fragments was measured
// first: every lane of such a load is on another row, i.e. another cache line, and the loads took four times the
// address cycles of coalesced ones -- no faster than the kernel it was to replace.)
//   LDS image of a plane: [128][32] f16, 64-byte rows whose 16-byte chunks are XOR-swizzled with bits 2..3 of the row
//   (conflict-free ds_write_b128 of the incoming pieces and ds_read_b128 of the fragments)

__device__ inline int bp_lds_off(int n, int c) { return n * 64 + 16 * (c ^ ((n >> 2) & 3)); }

// TAIL (K is no multiple of 32): elements at or beyond k_left are zeroed (every step of such a launch pays the
// selects: the choice is per kernel, not per step, so that a step's body stays one straight-line region)
template <bool TAIL>
__device__ inline void f16p_split(const f32x4 (&r)[4], float scale, int k_left, bf16x8 (&f)[2][2]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < 2; s++) {
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            f32x4 x = r[2 * s + q];
            if constexpr (TAIL) {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = (8 * s + 4 * q + e) < k_left ? x[e] : 0.f;
            }
            split2_pair(f32x2{x[0], x[1]}, scale, h[2 * q], l[2 * q]);
            split2_pair(f32x2{x[2], x[3]}, scale, h[2 * q + 1], l[2 * q + 1]);
        }
        f[s][0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
        f[s][1] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    }
}

// scheduling recipe of a step (6 NI MFMAs): the raw A rows and the first slice's fragments of the first four column
// tiles early, every further pair of fragments once the MFMAs of the tile before it have issued, the next step's split
// (about 80 vector operations) spread behind the MFMAs, the LDS writes of the register path behind the last ones
template <int I, int NI, int WRITES>
__device__ inline void f16p_interleave() {
    constexpr int kMfma = 6 * NI, kValu = 84;
    if constexpr (I == 0) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
    if constexpr (I < kMfma) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (I % 3 == 2 && I / 3 < 2 * NI - 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (kValu * (I + 1)) / kMfma - (kValu * I) / kMfma, 0);
        if constexpr (WRITES > 0 && I >= kMfma - WRITES) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        f16p_interleave<I + 1, NI, WRITES>();
    }
}

// ADMA / BDMA: the A rows / B planes travel global -> LDS directly; else through registers (coalesced 16-byte loads at
// the start of a step, ds_write_b128 at its end)
// NI: 32-column MFMA tiles per wave (4: 128 x 128 outputs per workgroup, two workgroups per compute unit; 8: 128 x 256,
// 256 accumulator registers per wave, one workgroup per compute unit -- one wave per SIMD with the whole register file:
// half the split arithmetic and A traffic per MFMA)
// ABL (diagnostic builds only, -DDLRM_F16P_ABLATE; results are wrong, timings tell what a step waits for): bit 0 = no
// global loads inside the loop, bit 1 = no split arithmetic (raw bits as pieces), bit 2 = no barrier per step,
// bit 4 = one column tile's B fragments reused for all (a quarter of the LDS reads), bit 5 / bit 6 = no loads of A / of B
// inside the loop
// AST: stages of raw A rows (ADMA only).  2: the rows of step t+2 are requested during step t and split during step
// t+1 -- one step to arrive; 3: those of step t+3, two steps to arrive (80 KB of LDS per workgroup, still two per
// compute unit).  Measured equal (the kernel does not wait for A; with all of its parts it runs at the chip's power
// limit, 1.7 GHz in back-to-back launches: every part's cost shows up in the clock); only 2 is built outside the
// diagnostic build.
// TRANSPOSED_W changes no code: it puts the product into the kernel's name (false: x W^T, the forward pass; true: dz W,
// backward-data), so that a profiler's per-kernel sums tell the two apart
template <bool TAIL, bool ADMA, bool BDMA, int NI, int ABL = 0, int AST = 2, bool TRANSPOSED_W = false>
__global__ __launch_bounds__(kGemmThreads, NI == 8 ? 1 : 2) void gemm_f16p_kernel(GemmParams p) {
    constexpr int kPStage = 4096 * NI;          // bytes per B stage: h plane, l plane, 32 NI rows of 64 bytes each
    // LDS: two stages of B planes (16 KB each, shared by the four waves), then two stages of raw float32 A rows
    // (16 KB each: 4 KB per wave, read only by the wave that fetched them)
    constexpr int kAStage = 4 * 32 * 128, kAOff = 2 * kPStage;
    static_assert(AST == 2 || (AST == 3 && ADMA), "three A stages exist for the direct-to-LDS path only");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kPStage + AST * kAStage];
    static_assert(2 * kPStage + 2 * kAStage >= 4 * 32 * (64 + 4) * 4, "the epilogue parks 32 x 64 per wave at a time");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void lds_void;
    const float scale_a = split2_scale(__float_as_uint(*p.a_max));
    const float scale_b = split2_scale(__float_as_uint(*p.b_max));

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    (void)tid;
    const int nwg = p.gx * p.gy;
    const int lin = blockIdx.x, xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int bx = tile % p.gx, by = tile / p.gx;
    const int m0 = by * 128, n0 = bx * 32 * NI;
    const int steps = (p.K + 31) / 32;

    // A: a step of this wave's 32 rows is 4 KB of float32 (128 bytes per row), fetched by four direct-to-LDS loads of
    // 8 rows each.  Within one load 8 consecutive lanes cover one row's 128 bytes (coalesced), in an order that leaves
    // chunk c of row r at position c ^ ((r >> 1) & 7) of the row's image: lane (r, half) then reads ITS 64 bytes --
    // chunks 4 half .. 4 half + 3 -- with four conflict-free ds_read_b128.  Rows past M are clamped (they only feed
    // outputs that are never stored).
    const float *a_origin = p.A + (int64_t)m0 * p.lda;
    unsigned a_off[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pr = lane >> 3, pc = lane & 7;
        const int row = m0 + 32 * wave + 8 * j + pr;
        const int rowc = row < p.M ? row : p.M - 1;
        a_off[j] = (unsigned)((rowc - m0) * (int)p.lda * 4 + 16 * (ADMA ? (pc ^ (pr >> 1) ^ (4 * (j & 1))) : pc));
    }
    // (register path: lane (pr, pc) of load j holds chunk pc of row 8 j + pr and writes it to position pc ^ ((row >> 1) & 7))
    const int a_wr = (lane >> 3) * 128 + 16 * ((lane & 7) ^ ((lane >> 4) & 3));       // + 1024 j, and ^ 64 for odd j
    const int a_lds = kAOff + uwave * 4096;                           // this wave's rows inside an A stage (+ kAStage: other stage)
    const int a_rd = l31 * 128, a_sw = (l31 >> 1) & 7;
    // B planes go global -> LDS directly as well (64 lanes x 16 bytes land side by side, 16 rows of the image per
    // instruction).  Waves 0, 1 bring the lower / upper half of the h plane's rows, waves 2, 3 of the l plane's, NI
    // instructions each; the swizzle is applied on the global side: the lane that fills position pp of row n fetches
    // the row's chunk pp ^ ((n >> 2) & 3).
    const unsigned char *b_origin = (uwave >> 1 ? p.bp_l : p.bp_h) + ((int64_t)n0 + 16 * NI * (uwave & 1)) * 64;
    const unsigned b_off = (unsigned)((lane >> 2) * 64 + 16 * (BDMA ? ((lane & 3) ^ ((lane >> 4) & 3)) : (lane & 3)));
    const int b_wr = (lane >> 2) * 64 + 16 * ((lane & 3) ^ ((lane >> 4) & 3));        // + 1024 j (register path)
    const unsigned b_off16 = 1024u;                     // (16 rows of a step are 1 KB of the plane)
    const int b_lds = (uwave >> 1) * (2048 * NI) + (uwave & 1) * (1024 * NI);       // this wave's share of a B stage
    const int rd0 = bp_lds_off(l31, 2 * half), rd1 = bp_lds_off(l31, 2 * half + 1);   // slice 0 / 1, + 2048 per column tile

    f32x16 hi[NI], lo[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int r = 0; r < 16; r++) { hi[ni][r] = 0.f; lo[ni][r] = 0.f; }

    auto dma_a = [&](int t, int stage_off) {
        const int tc = t < steps ? t : steps - 1;     // (surplus requests re-read the last step)
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a_origin + 32 * tc), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + a_lds + stage_off + 1024 * j), 16, (int)a_off[j], 0, 0, 0);
    };
    auto dma_b = [&](int t, int stage) {
        const int tc = t < steps ? t : steps - 1;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(b_origin + tc * p.bp_ld), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < NI; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + b_lds + stage * kPStage + 1024 * j), 16, (int)(b_off + j * b_off16), 0, 0, 0);
    };
    // register path: request now, write to the image later
    auto load_a = [&](int t, u32x4 (&r)[4]) {
        const int tc = t < steps ? t : steps - 1;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a_origin + 32 * tc), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)a_off[j], 0, 0);
    };
    auto store_a = [&](int stage_off, const u32x4 (&r)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<u32x4 *>(smem + a_lds + stage_off + 1024 * j + (a_wr ^ (64 * (j & 1)))) = r[j];
    };
    auto load_b = [&](int t, u32x4 (&r)[NI]) {
        const int tc = t < steps ? t : steps - 1;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(b_origin + tc * p.bp_ld), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < NI; j++) r[j] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(b_off + j * b_off16), 0, 0);
    };
    auto store_b = [&](int stage, const u32x4 (&r)[NI]) {
#pragma unroll
        for (int j = 0; j < NI; j++) *reinterpret_cast<u32x4 *>(smem + b_lds + stage * kPStage + 1024 * j + b_wr) = r[j];
    };
    // this lane's 16 float32 of a step out of the wave's A image
    auto read_a = [&](int stage_off, f32x4 (&r)[4]) {
        const unsigned char *base = smem + a_lds + stage_off + a_rd;
#pragma unroll
        for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const f32x4 *>(base + 16 * ((4 * half + q) ^ a_sw));
    };
    // the 6 NI MFMAs of one step: slice s pairs A pieces af[s] with the B fragments at chunk (2 half + s)
    auto mfma_step = [&](const unsigned char *stage, const bf16x8 (&af)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const f16x8 ah = __builtin_bit_cast(f16x8, af[s][0]), al = __builtin_bit_cast(f16x8, af[s][1]);
            const int rd = s ? rd1 : rd0;
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int nb = (ABL & 16) ? 0 : ni * 2048;
                const f16x8 bh = *reinterpret_cast<const f16x8 *>(stage + rd + nb);
                const f16x8 bl = *reinterpret_cast<const f16x8 *>(stage + 2048 * NI + rd + nb);
                f32x16 c = lo[ni];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
                lo[ni] = c;
                hi[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hi[ni], 0, 0, 0);
            }
        }
    };

    bf16x8 fa[2][2], fb[2][2];    // A pieces of two steps in turn
    u32x4 ast[4], bst[NI];        // register path: an operand's pieces between their loads and their LDS writes
    // step t (parity par): MFMAs on (B stage par, f_c); meanwhile the raw A of step t+1 (A stage par^1) is read and
    // split into f_n, the B planes of step t+1 travel to B stage par^1, the A rows of step t+2 to A stage par (whose
    // rows were split during the previous step)
    // (three A stages: the stage step t's requests fill and the one its split reads rotate -- scalar offsets)
    int a_fill = 0, a_split = kAStage;
    auto step = [&](int t, int par, bf16x8 (&f_c)[2][2], bf16x8 (&f_n)[2][2]) {
        constexpr bool kLoadA = !(ABL & 1) && !(ABL & 32), kLoadB = !(ABL & 1) && !(ABL & 64);
        if constexpr (!kLoadB) {}
        else if constexpr (BDMA) dma_b(t + 1, par ^ 1);   // (every wave has passed the barrier behind its last read of that stage)
        else load_b(t + 1, bst);
        if constexpr (!kLoadA) {}
        else if constexpr (ADMA) dma_a(t + AST, AST == 2 ? par * kAStage : a_fill);
        else load_a(t + 2, ast);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ADMA && kLoadA) {
            // the A rows of step t+1 have landed (requests issued since then may be in flight: NI + 4 of this step, with
            // three stages also the 4 A requests of the step before; fewer if B goes through registers -- those are
            // counted by the compiler's own waits)
            __builtin_amdgcn_s_waitcnt(0x0F70 | ((kLoadB ? NI : 0) + 4 * (AST - 1)));
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 raw[4];
        read_a(AST == 2 ? (par ^ 1) * kAStage : a_split, raw);
        if constexpr (AST == 3) {
            a_fill = a_fill == 2 * kAStage ? 0 : a_fill + kAStage;
            a_split = a_split == 2 * kAStage ? 0 : a_split + kAStage;
        }
        if constexpr (ABL & 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) f_n[q >> 1][q & 1] = __builtin_bit_cast(bf16x8, raw[q]);
        } else f16p_split<TAIL>(raw, scale_a, p.K - 32 * (t + 1) - 16 * half, f_n);    // (step t + 1's elements)
        mfma_step(smem + par * kPStage, f_c);
        if constexpr (!BDMA) store_b(par ^ 1, bst);
        if constexpr (!ADMA) store_a(par * kAStage, ast);
        f16p_interleave<0, NI, (BDMA ? 0 : NI) + (ADMA ? 0 : 4)>();
        __builtin_amdgcn_sched_barrier(0);
        // this wave's pieces of the next B stage have landed (A requests issued behind them may stay in flight)
        if constexpr (BDMA && kLoadB) __builtin_amdgcn_s_waitcnt(0x0F70 | (kLoadA ? 4 : 0));
        if constexpr (!(ABL & 4)) __syncthreads();
    };
    if constexpr (ADMA) dma_a(0, 0); else load_a(0, ast);
    if constexpr (BDMA) dma_b(0, 0); else load_b(0, bst);
    if constexpr (!ADMA) store_a(0, ast);
    if constexpr (!BDMA) store_b(0, bst);
    if constexpr (ADMA) dma_a(1, kAStage);
    else { load_a(1, ast); store_a(kAStage, ast); }
    if constexpr (AST == 3) dma_a(2, 2 * kAStage);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ADMA || BDMA) __builtin_amdgcn_s_waitcnt(ADMA ? (0x0F70 | (4 * (AST - 1))) : 0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x4 raw[4];
        read_a(0, raw);
        f16p_split<TAIL>(raw, scale_a, p.K - 16 * half, fa);
    }
    __syncthreads();
    // pairs of steps, no branch inside the loop (control flow that merges the 128 accumulator registers of several
    // step bodies costs copies and spills); an odd last step follows it
    int t = 0;
    for (; t + 1 < steps; t += 2) {
        step(t, 0, fa, fb);
        step(t + 1, 1, fb, fa);
    }
    if (t < steps) step(t, 0, fa, fb);

    const float inv_a = 1.0f / scale_a, inv_b = 1.0f / scale_b;
    float *fsm = reinterpret_cast<float *>(smem);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // (surplus A requests must not land in the epilogue's staging area)
    __syncthreads();
#pragma unroll
    for (int hn = 0; hn < NI / 2; hn++) {      // 64 columns at a time: mixed terms in, operand scales out, then the usual epilogue
        f32x16 o[1][2];
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                o[0][ni][r] = (__builtin_fmaf(lo[2 * hn + ni][r], kSplit2LowInv, hi[2 * hn + ni][r]) * inv_a) * inv_b;
        store_output<1, 2, 1>(p, o, fsm, m0, n0 + 64 * hn, 0);
    }
}

// ---- gemm_f16p_kernel on v_mfma_f32_16x16x32_f16 ----------------------------------------------------------------
// Same product, same operands, same pipeline (both operands global -> LDS directly, two stages each, A rows wave
// private, one barrier per step), the matrix work as 48 MFMAs of 16 x 16 x 32 per step instead of 24 of 32 x 32 x 16:
// an accumulator element is read and written once per 32 multiply-adds instead of once per 16, and on operands that
// toggle the multipliers the part sustains 12 % more of them (tools/f16p_shape_bench.hip: 0.65 against 0.58 of
// 2.5 PFLOP/s) -- the kernel runs at the power limit, so that is time.  What changes is who holds what:
//   A: lane (r = lane & 15, g = lane >> 4) owns reduction elements 8g .. 8g+7 of rows r and 16 + r of the wave's 32: the
//      32 bytes at chunks 2g, 2g+1 of a row's 128; chunk c of row r sits at position c ^ aq_sw(r) of the row's image
//   B: lane (n, g) reads chunk g of column n's 64 bytes for each of the 8 column blocks; chunk c of row n sits at
//      position c ^ (0, 2, 3, 1)[(n >> 2) & 3]
// (both conflict-free for the four 16-lane groups a ds_read_b128 is served in), and the 16 x 16 result blocks: lane
// (n, g) holds column n, rows 4g .. 4g+3.
__device__ inline int bq_swz(int n) { return (0x1320 >> (4 * ((n >> 2) & 3))) & 3; }
__device__ inline int aq_swz(int r) { const int x = (r >> 1) & 7; return 6 * ((x >> 2) & 1) + (x & 1); }

// r[2m], r[2m+1]: the lane's 8 elements of row block m; elements at or beyond k_left (counted from the lane's first)
// are zeroed in a TAIL launch
template <bool TAIL>
__device__ inline void f16q_split(const f32x4 (&r)[4], float scale, int k_left, bf16x8 (&f)[2][2]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int m = 0; m < 2; m++) {
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            f32x4 x = r[2 * m + q];
            if constexpr (TAIL) {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = (4 * q + e) < k_left ? x[e] : 0.f;
            }
            split2_pair(f32x2{x[0], x[1]}, scale, h[2 * q], l[2 * q]);
            split2_pair(f32x2{x[2], x[3]}, scale, h[2 * q + 1], l[2 * q + 1]);
        }
        f[m][0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
        f[m][1] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    }
}

// a step's 48 MFMAs: the raw A rows and the fragments of the first two column blocks early, the fragments of column
// block cb + 2 behind the MFMAs of block cb, the next step's split spread behind all of them
template <int I>
__device__ inline void f16q_interleave() {
    constexpr int kMfma = 48, kValu = 84;
    if constexpr (I == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    if constexpr (I < kMfma) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (I % 6 == 5 && I / 6 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (kValu * (I + 1)) / kMfma - (kValu * I) / kMfma, 0);
        f16q_interleave<I + 1>();
    }
}

// the epilogue of store_output for 32 rows x 64 columns held as 2 x 4 blocks of 16 x 16 (one wave per 32 rows)
__device__ inline void store_output16(const GemmParams &p, const f32x4 (&acc)[2][4], float *ct_all, int m0, int n0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int CT_LD = 64 + 4, LPR = 16, RPI = 4;
    float *ct = ct_all + wave * (32 * CT_LD);
    const int cn = n0 + (lane % LPR) * 4;
    const bool c_vec = (reinterpret_cast<uintptr_t>(p.C) % 16 == 0) && (p.ldc % 4 == 0) &&
                       (p.mask == nullptr || ((reinterpret_cast<uintptr_t>(p.mask) % 16 == 0) && (p.mask_ld % 4 == 0)));
    float vmax = 0.f;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
        if (cn < p.N) bv.x = p.bias[cn];
        if (cn + 1 < p.N) bv.y = p.bias[cn + 1];
        if (cn + 2 < p.N) bv.z = p.bias[cn + 2];
        if (cn + 3 < p.N) bv.w = p.bias[cn + 3];
    }
    float4 mk[32 / RPI];
    const bool mask_vec = p.mask != nullptr && c_vec && cn + 3 < p.N;
    if (mask_vec) {
#pragma unroll
        for (int it = 0; it < 32 / RPI; it++) {
            const int m = m0 + wave * 32 + it * RPI + (lane / LPR);
            mk[it] = *reinterpret_cast<const float4 *>(p.mask + (int64_t)(m < p.M ? m : p.M - 1) * p.mask_ld + cn);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) ct[(16 * m + 4 * (lane >> 4) + r) * CT_LD + 16 * cb + (lane & 15)] = acc[m][cb][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / RPI; it++) {
        const int row = it * RPI + (lane / LPR);
        const int m = m0 + wave * 32 + row;
        float4 v = *reinterpret_cast<const float4 *>(ct + row * CT_LD + (lane % LPR) * 4);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (m >= p.M || cn >= p.N) continue;
        float *dst = p.C + (int64_t)m * p.ldc + cn;
        if (c_vec && cn + 3 < p.N) {
            if (mask_vec) {
                const float4 k = mk[it];
                v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f;
                v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4 *>(dst) = v;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (cn + q < p.N) {
                    float x = e[q];
                    if (p.mask) x = p.mask[(int64_t)m * p.mask_ld + cn + q] > 0.f ? x : 0.f;
                    dst[q] = x;
                    vmax = fmaxf(vmax, fabsf(x));
                }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (p.c_max != nullptr) fold_absmax(vmax, p.c_max);
}

template <bool TAIL, bool TRANSPOSED_W>      // (TRANSPOSED_W: part of the name only, as in gemm_f16p_kernel)
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f16q_kernel(GemmParams p) {
    constexpr int kPStage = 16384, kAStage = 16384, kAOff = 2 * kPStage;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kPStage + 2 * kAStage];
    typedef __attribute__((address_space(3))) void lds_void;
    const float scale_a = split2_scale(__float_as_uint(*p.a_max));
    const float scale_b = split2_scale(__float_as_uint(*p.b_max));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int nwg = p.gx * p.gy;
    const int lin = blockIdx.x, xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int bx = tile % p.gx, by = tile / p.gx;
    const int m0 = by * 128, n0 = bx * 128;
    const int steps = (p.K + 31) / 32;

    // requests: as in gemm_f16p_kernel, with this kernel's swizzles applied on the global side
    const float *a_origin = p.A + (int64_t)m0 * p.lda;
    unsigned a_off[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pr = lane >> 3, pc = lane & 7;
        const int row = m0 + 32 * wave + 8 * j + pr;
        const int rowc = row < p.M ? row : p.M - 1;
        a_off[j] = (unsigned)((rowc - m0) * (int)p.lda * 4 + 16 * (pc ^ aq_swz(8 * j + pr)));
    }
    const int a_lds = kAOff + uwave * 4096;
    int a_rd[2][2];       // [row block][first / second chunk] of this lane inside the wave's A image
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 2; q++) a_rd[m][q] = (16 * m + l15) * 128 + 16 * ((2 * g + q) ^ aq_swz(16 * m + l15));
    const unsigned char *b_origin = (uwave >> 1 ? p.bp_l : p.bp_h) + ((int64_t)n0 + 64 * (uwave & 1)) * 64;
    const unsigned b_off = (unsigned)((lane >> 2) * 64 + 16 * ((lane & 3) ^ bq_swz(lane >> 2)));
    const int b_lds = (uwave >> 1) * 8192 + (uwave & 1) * 4096;
    const int rd = l15 * 64 + 16 * (g ^ bq_swz(l15));        // + 1024 per column block, + 8192: the l plane

    f32x4 hi[2][8], lo[2][8];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int cb = 0; cb < 8; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) { hi[m][cb][r] = 0.f; lo[m][cb][r] = 0.f; }

    auto dma_a = [&](int t, int stage) {
        const int tc = t < steps ? t : steps - 1;     // (surplus requests re-read the last step)
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a_origin + 32 * tc), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + a_lds + stage * kAStage + 1024 * j), 16, (int)a_off[j], 0, 0, 0);
    };
    auto dma_b = [&](int t, int stage) {
        const int tc = t < steps ? t : steps - 1;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(b_origin + tc * p.bp_ld), 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + b_lds + stage * kPStage + 1024 * j), 16, (int)(b_off + 1024u * j), 0, 0, 0);
    };
    auto read_a = [&](int stage, f32x4 (&r)[4]) {
        const unsigned char *base = smem + a_lds + stage * kAStage;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int q = 0; q < 2; q++) r[2 * m + q] = *reinterpret_cast<const f32x4 *>(base + a_rd[m][q]);
    };
    auto mfma_step = [&](const unsigned char *stage, const bf16x8 (&af)[2][2]) {
        const f16x8 ah0 = __builtin_bit_cast(f16x8, af[0][0]), al0 = __builtin_bit_cast(f16x8, af[0][1]);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, af[1][0]), al1 = __builtin_bit_cast(f16x8, af[1][1]);
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(stage + rd + 1024 * cb);
            const f16x8 bl = *reinterpret_cast<const f16x8 *>(stage + 8192 + rd + 1024 * cb);
            f32x4 c = lo[0][cb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c, 0, 0, 0);
            lo[0][cb] = c;
            hi[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, hi[0][cb], 0, 0, 0);
            c = lo[1][cb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c, 0, 0, 0);
            lo[1][cb] = c;
            hi[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, hi[1][cb], 0, 0, 0);
        }
    };

    bf16x8 fa[2][2], fb[2][2];
    // step t (parity par): as in gemm_f16p_kernel -- MFMAs on (B stage par, f_c); the raw A of step t+1 (A stage par^1)
    // is read and split into f_n; B of step t+1 travels to B stage par^1, A of step t+2 to A stage par
    auto step = [&](int t, int par, bf16x8 (&f_c)[2][2], bf16x8 (&f_n)[2][2]) {
        dma_b(t + 1, par ^ 1);
        dma_a(t + 2, par);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);          // the A rows of step t+1 have landed (this step's 8 requests may fly)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 raw[4];
        read_a(par ^ 1, raw);
        f16q_split<TAIL>(raw, scale_a, p.K - 32 * (t + 1) - 8 * g, f_n);
        mfma_step(smem + par * kPStage, f_c);
        f16q_interleave<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F74);              // this wave's pieces of the next B stage have landed
        __syncthreads();
    };
    dma_a(0, 0);
    dma_b(0, 0);
    dma_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F74);
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x4 raw[4];
        read_a(0, raw);
        f16q_split<TAIL>(raw, scale_a, p.K - 8 * g, fa);
    }
    __syncthreads();
    int t = 0;
    for (; t + 1 < steps; t += 2) {       // (pairs of steps without a branch, the odd last one behind them)
        step(t, 0, fa, fb);
        step(t + 1, 1, fb, fa);
    }
    if (t < steps) step(t, 0, fa, fb);

    const float inv_a = 1.0f / scale_a, inv_b = 1.0f / scale_b;
    float *fsm = reinterpret_cast<float *>(smem);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // (surplus A requests must not land in the epilogue's staging area)
    __syncthreads();
#pragma unroll
    for (int hn = 0; hn < 2; hn++) {         // 64 columns at a time: mixed terms in, operand scales out, then the usual epilogue
        f32x4 o[2][4];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    o[m][cb][r] = (__builtin_fmaf(lo[m][4 * hn + cb][r], kSplit2LowInv, hi[m][4 * hn + cb][r]) * inv_a) * inv_b;
        store_output16(p, o, fsm, m0, n0 + 64 * hn);
    }
}

// ---- the planes gemm_f16p_kernel reads: for a weight matrix W [N,K], its f16 pieces (times the scale of max |W|) as
// [K rounded up to 32, in steps of 32][N rounded up to 256][32] (forward: reduction over K) and, transposed, [N rounded
// up to 32, in steps of 32][K rounded up to 256][32] (backward-data: reduction over N), zero in the padding.  Step-major:
// the 64 bytes that one reduction step needs of consecutive rows lie side by side, so a wave's 1 KB request covers
// eight whole 128-byte lines (row-major planes cost sixteen half lines per request, and the vector memory path's time
// goes by the line).  Layout of the buffer: forward h, forward l, transposed h, transposed l.
struct PlaneDims { int64_t f_rows, f_ld, t_rows, t_ld; };     // rows and (rounded) reduction length, in f16 elements, of the two images
static PlaneDims plane_dims(int N, int K) {
    return PlaneDims{ceil_div(N, 256) * 256, ceil_div(K, 32) * 32, ceil_div(K, 256) * 256, ceil_div(N, 32) * 32};
}
static size_t plane_bytes_total(int N, int K) {
    const PlaneDims d = plane_dims(N, K);
    return (size_t)2 * 2 * (d.f_rows * d.f_ld + d.t_rows * d.t_ld);
}

struct SplitJob {
    const float *w;
    int64_t ld;
    int N, K;
    const unsigned *amax;
    _Float16 *fh, *fl, *th, *tl;
    PlaneDims d;
};
struct SplitJobs { SplitJob j[kAbsmaxMany]; };

__device__ inline void split2_one(float x, float scale, _Float16 &h, _Float16 &l) {
    const float xs = x * scale;
    h = (_Float16)xs;
    l = (_Float16)((xs - (float)h) * kSplit2LowScale);
}

__global__ __launch_bounds__(256) void split_weights_kernel(SplitJobs jobs) {
    const SplitJob j = jobs.j[blockIdx.y];
    const float scale = split2_scale(*j.amax);
    // forward image: consecutive threads along k (coalesced reads and writes)
    const int64_t nf = j.d.f_rows * j.d.f_ld, per_step = j.d.f_rows * 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / per_step, rem = i - t * per_step;
        const int64_t n = rem >> 5, k = 32 * t + (rem & 31);
        const float x = (n < j.N && k < j.K) ? j.w[n * j.ld + k] : 0.f;
        _Float16 h, l;
        split2_one(x, scale, h, l);
        j.fh[i] = h; j.fl[i] = l;
    }
    // transposed image: consecutive threads along n (coalesced writes; the strided reads of a wave stay within the
    // 64 lines it walks along k)
    const int64_t n_blocks = j.d.t_ld / 32, k_blocks = j.d.t_rows / 32;     // 32 x 32 blocks, one per wave trip
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < n_blocks * k_blocks; b += (int64_t)gridDim.x * 4) {
        const int64_t nb = b % n_blocks, kb = b / n_blocks;
        const int64_t n = nb * 32 + (lane & 31);
#pragma unroll 4
        for (int kk = (lane >> 5); kk < 32; kk += 2) {
            const int64_t k = kb * 32 + kk;
            const float x = (n < j.N && k < j.K) ? j.w[n * j.ld + k] : 0.f;
            _Float16 h, l;
            split2_one(x, scale, h, l);
            j.th[(nb * j.d.t_rows + k) * 32 + (lane & 31)] = h; j.tl[(nb * j.d.t_rows + k) * 32 + (lane & 31)] = l;
        }
    }
}

// out[i] = sum_s slabs[s*stride + i] for an [rows, cols] matrix (ld_out).  A 256-thread workgroup owns
// 64 consecutive outputs; thread (q, e) adds the slabs s = q, q+4, ... of output e with eight loads in
// flight, and the four partial sums meet in LDS in a fixed order (deterministic).  One launch serves two
// such problems with the same number of slabs (a weight gradient and its bias gradient): workgroups
// [0, blocks_a) take the first, the rest the second.
// (struct SlabSum: common.h)
// W = 4: every thread takes four consecutive outputs with 16-byte loads and one 16-byte store (rows, slabs and the
// output start on 16-byte boundaries); the sums and their order per output are those of W = 1.
__device__ inline bool slab_sum_vec(const SlabSum &p) {
    return p.cols % 4 == 0 && p.stride % 4 == 0 && p.ld_out % 4 == 0 && reinterpret_cast<uintptr_t>(p.slabs) % 16 == 0 &&
           reinterpret_cast<uintptr_t>(p.out) % 16 == 0;
}
template <int W>
__device__ inline void reduce_slabs_w(const SlabSum &p, int splits, unsigned bid, unsigned nblk, float (*part)[64 * 4]) {
    typedef float vecw __attribute__((ext_vector_type(W)));
    const float *__restrict__ slabs = p.slabs;
    const int64_t n = (int64_t)p.rows * p.cols;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t i0 = (int64_t)bid * 64 * W; i0 < n; i0 += (int64_t)nblk * 64 * W) {
        const int64_t i = i0 + (int64_t)e * W;
        vecw a = 0.f;
        if (i < n) {
            int s = q;
            for (; s + 28 < splits; s += 32) {
                vecw v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const vecw *>(slabs + (int64_t)(s + 4 * u) * p.stride + i);
#pragma unroll
                for (int u = 0; u < 8; u++) a += v[u];
            }
            for (; s < splits; s += 4) a += *reinterpret_cast<const vecw *>(slabs + (int64_t)s * p.stride + i);
        }
        *reinterpret_cast<vecw *>(&part[q][e * W]) = a;
        __syncthreads();
        if (q == 0 && i < n) {
            const vecw t = (*reinterpret_cast<const vecw *>(&part[0][e * W]) + *reinterpret_cast<const vecw *>(&part[1][e * W])) +
                           (*reinterpret_cast<const vecw *>(&part[2][e * W]) + *reinterpret_cast<const vecw *>(&part[3][e * W]));
            const int r = (int)(i / p.cols), c = (int)(i - (int64_t)r * p.cols);
            *reinterpret_cast<vecw *>(p.out + (int64_t)r * p.ld_out + c) = t;
        }
        __syncthreads();
    }
}
__device__ inline void reduce_slabs_body(const SlabSum &p, int splits, unsigned bid, unsigned nblk) {
    __shared__ __attribute__((aligned(16))) float part[4][64 * 4];
    if (slab_sum_vec(p)) reduce_slabs_w<4>(p, splits, bid, nblk, part);
    else reduce_slabs_w<1>(p, splits, bid, nblk, part);
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(SlabSum pa, SlabSum pb, int splits, unsigned blocks_a) {
    if (blockIdx.x >= blocks_a) reduce_slabs_body(pb, splits, blockIdx.x - blocks_a, gridDim.x - blocks_a);
    else reduce_slabs_body(pa, splits, blockIdx.x, blocks_a);
}

// the same for up to kReduceJobs problems with their own numbers of slabs: workgroups [first[j], first[j+1]) take job j
constexpr int kReduceJobs = 24;
struct ReduceJobs {
    SlabSum p[kReduceJobs];
    int splits[kReduceJobs];
    unsigned first[kReduceJobs + 1];
    int n;
};
__global__ __launch_bounds__(256) void reduce_slabs_many_kernel(ReduceJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.first[j + 1]) j++;
    reduce_slabs_body(jobs.p[j], jobs.splits[j], blockIdx.x - jobs.first[j], jobs.first[j + 1] - jobs.first[j]);
}

static bool slab_sum_vec_host(const SlabSum &p) {
    return p.cols % 4 == 0 && p.stride % 4 == 0 && p.ld_out % 4 == 0 && reinterpret_cast<uintptr_t>(p.slabs) % 16 == 0 &&
           reinterpret_cast<uintptr_t>(p.out) % 16 == 0;
}
static unsigned reduce_blocks(const SlabSum &p) {
    const int64_t per = slab_sum_vec_host(p) ? 256 : 64;
    const int64_t n = ceil_div((int64_t)p.rows * p.cols, per);
    return (unsigned)(n < 4096 ? (n < 1 ? 1 : n) : 4096);
}

int reduce_slabs_many(const SlabReduce *items, int n, hipStream_t s) {
    DLRM_REQUIRE(n >= 0 && (items != nullptr || n == 0), "bad arguments");
    ReduceJobs jobs{};
    auto flush = [&]() -> int {
        if (jobs.n == 0) return DLRM_OK;
        hipLaunchKernelGGL(reduce_slabs_many_kernel, dim3(jobs.first[jobs.n]), dim3(256), 0, s, jobs);
        DLRM_LAUNCH_CHECK();
        jobs.n = 0;
        return DLRM_OK;
    };
    auto add = [&](const SlabSum &p, int splits) -> int {
        if (jobs.n == kReduceJobs)
            if (int rc = flush()) return rc;
        if (jobs.n == 0) jobs.first[0] = 0;
        jobs.p[jobs.n] = p;
        jobs.splits[jobs.n] = splits;
        jobs.first[jobs.n + 1] = jobs.first[jobs.n] + reduce_blocks(p);
        jobs.n++;
        return DLRM_OK;
    };
    for (int i = 0; i < n; i++) {
        if (items[i].splits <= 0) continue;
        if (int rc = add(items[i].w, items[i].splits)) return rc;
        if (items[i].has_b)
            if (int rc = add(items[i].b, items[i].splits)) return rc;
    }
    return flush();
}

// dW (and db, if wanted) from their partial sums, one launch
// (pending: the reduction is not launched but written down for reduce_slabs_many)
static int launch_reduce(const float *slabs, int64_t stride, int N, int K, float *dw, int64_t dw_ld, const float *db_slabs,
                         float *db, int splits, hipStream_t s, SlabReduce *pending = nullptr) {
    SlabSum a{slabs, stride, N, K, dw, dw_ld};
    SlabSum b{db_slabs, (int64_t)N, 1, N, db, (int64_t)N};
    const unsigned blocks_a = reduce_blocks(a);
    const unsigned blocks_b = db ? reduce_blocks(b) : 0u;
    if (pending) {
        pending->w = a; pending->b = b; pending->splits = splits; pending->has_b = db != nullptr;
        return DLRM_OK;
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, s, a, b, splits, blocks_a);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

__global__ void sgd_step_kernel(float *__restrict__ w, const float *__restrict__ g, int64_t n, float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        w[i] -= lr * g[i];
}

// Tile shapes by falling efficiency.  A launch takes the first one that gives every compute unit a
// workgroup (measured at batch 8192: 256 workgroups of 128 x 128 beat 512 of 128 x 64; below 256 the
// smaller tiles win), else the one with the most workgroups (ties: the larger tile).  A split reduction
// aims at two workgroups per compute unit.
struct TileShape { int tm, tn; };
static const TileShape kTiles[4] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
constexpr int64_t kWantWgs = 512, kTileWgs = 256;

static int splits_for(int M_red, int64_t tiles) {
    // enough workgroups for 256 CUs x 2, each split at least 8 reduction steps deep
    int64_t want = ceil_div(kWantWgs, tiles);
    int64_t max_by_depth = M_red / (BK * 8);
    if (max_by_depth < 1) max_by_depth = 1;
    int64_t s = want < max_by_depth ? want : max_by_depth;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// tile for an [M,N] output; red > 0: the reduction (of that length) may be split over workgroups as well
static TileShape pick_tile(int64_t M, int64_t N, int64_t red, int *splits_out) {
    static const int forced = [] { const char *e = getenv("DLRM_GEMM_TILE"); return e ? atoi(e) : -1; }();   // tuning aid
    int best = 0, best_splits = 1;
    int64_t best_wgs = -1;
    for (int i = 0; i < 4; i++) {
        const int64_t tiles = ceil_div(M, kTiles[i].tm) * ceil_div(N, kTiles[i].tn);
        const int sp = red > 0 ? splits_for((int)red, tiles) : 1;
        const int64_t wgs = tiles * sp;
        if (forced == i || (forced < 0 && wgs >= kTileWgs)) { best = i; best_splits = sp; break; }
        if (forced < 0 && wgs > best_wgs) { best = i; best_splits = sp; best_wgs = wgs; }
    }
    if (splits_out) *splits_out = best_splits;
    return kTiles[best];
}

template <bool A_RC, bool B_RC, int TM, int TN>
static void launch_split(const GemmParams &p, int pieces, unsigned nwg, hipStream_t s) {
    if (pieces == 3) hipLaunchKernelGGL((gemm_split_kernel<A_RC, B_RC, TM, TN, 3>), dim3(nwg), dim3(kGemmThreads), 0, s, p);
    else hipLaunchKernelGGL((gemm_split_kernel<A_RC, B_RC, TM, TN, 2>), dim3(nwg), dim3(kGemmThreads), 0, s, p);
}

// ---- largest magnitude of an operand (f16x2 evaluation): one pass over a [rows, cols] matrix, folded into *slot with
// atomicMax on the bit patterns (non-negative floats order like unsigned integers; the slot must start at 0)
struct AbsmaxJob {
    const float *x;
    int64_t rows, ld;
    int cols, vec, lanes_per_row;   // vec: rows start on 16-byte boundaries; lanes_per_row: a power of two, at most 64
    unsigned *slot;
};
struct AbsmaxJobs { AbsmaxJob j[kAbsmaxMany]; };

// lanes_per_row lanes share a row; a wave takes 64 / lanes_per_row rows at a time; workgroup `block` of `n_blocks`
__device__ inline void absmax_rows(const AbsmaxJob &a, int64_t block, int64_t n_blocks) {
    const int lane = threadIdx.x & 63;
    const int lr = lane / a.lanes_per_row, lc = lane % a.lanes_per_row;
    const int rows_per_wave = 64 / a.lanes_per_row;
    const int64_t wave = block * 4 + (threadIdx.x >> 6), n_waves = n_blocks * 4;
    const int cols4 = a.vec ? a.cols / 4 : 0;     // 16-byte pieces per row
    unsigned m = 0;
    for (int64_t r = wave * rows_per_wave + lr; r < a.rows; r += n_waves * rows_per_wave) {
        const float *row = a.x + r * a.ld;
        for (int c = lc; c < cols4; c += a.lanes_per_row) {
            const float4 v = *reinterpret_cast<const float4 *>(row + 4 * c);
            const unsigned p = __float_as_uint(v.x) & 0x7fffffffu, q = __float_as_uint(v.y) & 0x7fffffffu;
            const unsigned t = __float_as_uint(v.z) & 0x7fffffffu, u = __float_as_uint(v.w) & 0x7fffffffu;
            m = max(max(m, max(p, q)), max(t, u));
        }
        for (int c = 4 * cols4 + lc; c < a.cols; c += a.lanes_per_row) m = max(m, __float_as_uint(row[c]) & 0x7fffffffu);
    }
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, k, 64));
    // one atomic per workgroup, and only if it can still raise the slot (thousands of atomics on one address would
    // queue up behind each other; a stale read here costs one atomic too many, never a wrong result)
    __shared__ unsigned part[4];
    if (lane == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m > *reinterpret_cast<volatile unsigned *>(a.slot)) atomicMax(a.slot, m);
    }
}

__global__ __launch_bounds__(256) void absmax_kernel(AbsmaxJob a) { absmax_rows(a, blockIdx.x, gridDim.x); }

// several small matrices (a model's weight matrices) in one launch: blockIdx.y picks the matrix
__global__ __launch_bounds__(256) void absmax_many_kernel(AbsmaxJobs jobs) { absmax_rows(jobs.j[blockIdx.y], blockIdx.x, gridDim.x); }

static AbsmaxJob absmax_job(const float *x, int64_t rows, int cols, int64_t ld, float *slot, int64_t *blocks) {
    AbsmaxJob a{};
    a.x = x; a.rows = rows; a.ld = ld; a.cols = cols; a.slot = reinterpret_cast<unsigned *>(slot);
    a.vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && ld % 4 == 0;
    const int per_row = a.vec ? (cols + 3) / 4 : cols;
    a.lanes_per_row = 1;
    while (a.lanes_per_row < 64 && a.lanes_per_row < per_row) a.lanes_per_row <<= 1;
    const int64_t b = ceil_div(ceil_div(rows, 64 / a.lanes_per_row), 4);
    *blocks = b < 1 ? 1 : (b > 2048 ? 2048 : b);
    return a;
}

static int launch_absmax(const float *x, int64_t rows, int cols, int64_t ld, float *slot, hipStream_t s) {
    int64_t blocks = 1;
    const AbsmaxJob a = absmax_job(x, rows, cols, ld, slot, &blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

int absmax_many(const AbsmaxItem *items, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kAbsmaxMany) {
        AbsmaxJobs jobs{};
        const int cnt = n - i0 < kAbsmaxMany ? n - i0 : kAbsmaxMany;
        int64_t most = 1;
        for (int i = 0; i < cnt; i++) {
            int64_t b = 1;
            jobs.j[i] = absmax_job(items[i0 + i].x, items[i0 + i].rows, items[i0 + i].cols, items[i0 + i].ld, items[i0 + i].slot, &b);
            most = b > most ? b : most;
        }
        most = most > 64 ? 64 : most;     // (small matrices: a few dozen workgroups each)
        hipLaunchKernelGGL(absmax_many_kernel, dim3((unsigned)most, (unsigned)cnt), dim3(256), 0, s, jobs);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

// Slots for callers that bring no operand maxima: a ring inside the loaded code object (no allocation at launch).
// A launch takes the next pair; the ring is long enough that a pair is not handed out again while a launch that was
// given it can still be in flight on any stream of an ordinary host (a captured graph keeps the pairs it was
// captured with: hosts that replay graphs beside direct launches pass their own slots in the descriptors).
constexpr unsigned kMaxRing = 8192;
static __device__ unsigned g_max_ring[2 * kMaxRing];
static std::atomic<void *> g_max_ring_ptr[64];   // (two host threads may ask at once: both find the same address)
static std::atomic<unsigned> g_max_ring_next{0};

static int ring_pair(float **out) {
    int dev = -1;
    DLRM_HIP(hipGetDevice(&dev));
    DLRM_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d outside 0..63", dev);
    void *q = g_max_ring_ptr[dev].load(std::memory_order_acquire);
    if (q == nullptr) {
        DLRM_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(g_max_ring)));
        DLRM_REQUIRE(q != nullptr, "hipGetSymbolAddress returned NULL");
        g_max_ring_ptr[dev].store(q, std::memory_order_release);
    }
    *out = static_cast<float *>(q) + 2 * (g_max_ring_next.fetch_add(1) % kMaxRing);
    return DLRM_OK;
}

// fills p.a_max / p.b_max for the f16x2 evaluation: the caller's slots, or a pass over the operand into ring slots
static int operand_maxima(GemmParams &p, const float *given_a, int64_t rows_a, int cols_a, const float *given_b, int64_t rows_b,
                          int cols_b, hipStream_t s) {
    p.a_max = given_a; p.b_max = given_b;
    if (given_a && given_b) return DLRM_OK;
    float *pair = nullptr;
    if (int rc = ring_pair(&pair)) return rc;
    DLRM_HIP(hipMemsetAsync(pair, 0, 2 * sizeof(float), s));
    if (!given_a) {
        if (int rc = launch_absmax(p.A, rows_a, cols_a, p.lda, pair, s)) return rc;
        p.a_max = pair;
    }
    if (!given_b) {
        if (int rc = launch_absmax(p.B, rows_b, cols_b, p.ldb, pair + 1, s)) return rc;
        p.b_max = pair + 1;
    }
    return DLRM_OK;
}

template <bool A_RC, bool B_RC>
static int launch_gemm(GemmParams p, int splits, TileShape ts, int mode, hipStream_t s) {
    DLRM_REQUIRE(mode == DLRM_GEMM_DEFAULT || mode == DLRM_GEMM_F32_MFMA || mode == DLRM_GEMM_BF16X3 || mode == DLRM_GEMM_F16X2,
                 "gemm_mode must be DLRM_GEMM_DEFAULT, _F32_MFMA, _BF16X3 or _F16X2, got %d", mode);
    const int pieces = mode == DLRM_GEMM_F16X2 ? 2 : 3;
    DLRM_REQUIRE(pieces == 3 || (p.a_max != nullptr && p.b_max != nullptr), "f16x2 evaluation without operand maxima");
    if (mode == DLRM_GEMM_F32_MFMA) ts = TileShape{BM, BN};    // the float32-MFMA kernel has the one tile
    p.gx = (int)ceil_div(p.N, ts.tn); p.gy = (int)ceil_div(p.M, ts.tm); p.gz = splits;
    const int64_t nwg = (int64_t)p.gx * p.gy * p.gz;
    DLRM_REQUIRE(nwg < (1ll << 31), "GEMM grid too large");
    if (mode == DLRM_GEMM_F32_MFMA)
        hipLaunchKernelGGL((gemm_f32_kernel<A_RC, B_RC>), dim3((unsigned)nwg), dim3(kGemmThreads), 0, s, p);
    else if (ts.tm == 128 && ts.tn == 128) launch_split<A_RC, B_RC, 128, 128>(p, pieces, (unsigned)nwg, s);
    else if (ts.tm == 128) launch_split<A_RC, B_RC, 128, 64>(p, pieces, (unsigned)nwg, s);
    else if (ts.tn == 128) launch_split<A_RC, B_RC, 64, 128>(p, pieces, (unsigned)nwg, s);
    else launch_split<A_RC, B_RC, 64, 64>(p, pieces, (unsigned)nwg, s);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

static bool aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

// DLRM_F16P=0 keeps the pre-split planes unused (A/B timing)
static bool f16p_enabled() {
    static const bool on = [] { const char *e = getenv("DLRM_F16P"); return !(e && e[0] == '0'); }();
    return on;
}

// C [M,N] = A [M,K] times the planes of a weight matrix (`transposed`: of its transpose), through gemm_f16p_kernel when
// the shape suits it: p carries A, C, the epilogue fields and both maxima.  *done = false: the caller takes its
// general path (small grids keep the smaller tiles of gemm_split_kernel).
static int launch_f16p(GemmParams p, const dlrm_weight_planes *wp, bool transposed, const float *w, int64_t w_ld, bool *done,
                       hipStream_t s) {
    *done = false;
    if (wp == nullptr || !f16p_enabled()) return DLRM_OK;
    DLRM_REQUIRE(wp->d_planes != nullptr && wp->d_w == w && wp->w_ld == w_ld &&
                     (transposed ? (wp->N == p.K && wp->K == p.N) : (wp->N == p.N && wp->K == p.K)),
                 "w_planes do not belong to this layer's weight matrix");
    const int64_t k32 = ceil_div(p.K, 32) * 32;
    if (!p.a_vec || (p.K % 32 != 0 && p.lda < k32)) return DLRM_OK;
    // DLRM_F16P_PATH=1 (tuning aid): both operands through registers instead of global -> LDS directly (the mixed
    // forms were measured too, all within 5 %; the kernel keeps the switches, two forms are built)
    static const int path = [] { const char *e = getenv("DLRM_F16P_PATH"); return e ? (atoi(e) != 0) : 0; }();
    // 128 x 256 tiles -- NI = 8, one wave per SIMD with 256 accumulator registers, half the split arithmetic per MFMA --
    // were measured slower than two 128 x 128 workgroups per compute unit on every layer of the MLPerf model at batch
    // 65536 (1024 x 1024 forward: 543 against 486 us on one box): with one wave per SIMD nothing covers a wave's
    // waits at the barrier and in the epilogue.  The kernel keeps the template parameter; only NI = 4 is built.
    const bool wide = false;
    p.gx = (int)ceil_div(p.N, wide ? 256 : 128); p.gy = (int)ceil_div(p.M, 128); p.gz = 1;
    if ((int64_t)p.gx * p.gy < kTileWgs) return DLRM_OK;
    const PlaneDims d = plane_dims(wp->N, wp->K);
    const unsigned char *base = static_cast<const unsigned char *>(wp->d_planes);
    if (!transposed) { p.bp_h = base; p.bp_l = base + 2 * d.f_rows * d.f_ld; p.bp_ld = 64 * d.f_rows; }
    else {
        base += 4 * d.f_rows * d.f_ld;
        p.bp_h = base; p.bp_l = base + 2 * d.t_rows * d.t_ld; p.bp_ld = 64 * d.t_rows;
    }
    const dim3 grid((unsigned)(p.gx * p.gy)), block(kGemmThreads);
    const bool tail = p.K % 32 != 0;
#define DLRM_F16P_LAUNCH(A, B)                                                                         \
    do {                                                                                               \
        if (tail && transposed) hipLaunchKernelGGL((gemm_f16p_kernel<true, A, B, 4, 0, 2, true>), grid, block, 0, s, p);        \
        else if (tail) hipLaunchKernelGGL((gemm_f16p_kernel<true, A, B, 4, 0, 2, false>), grid, block, 0, s, p);              \
        else if (transposed) hipLaunchKernelGGL((gemm_f16p_kernel<false, A, B, 4, 0, 2, true>), grid, block, 0, s, p);        \
        else hipLaunchKernelGGL((gemm_f16p_kernel<false, A, B, 4, 0, 2, false>), grid, block, 0, s, p);                       \
    } while (0)
#ifdef DLRM_F16P_ABLATE
    static const int abl = [] { const char *e = getenv("DLRM_F16P_ABLATE"); return e ? atoi(e) : 0; }();
#define DLRM_F16P_ABL(V) case V: hipLaunchKernelGGL((gemm_f16p_kernel<false, true, true, 4, V>), grid, block, 0, s, p); break;
    if (abl != 0 && !tail) {
        switch (abl) {
        DLRM_F16P_ABL(1) DLRM_F16P_ABL(2) DLRM_F16P_ABL(4) DLRM_F16P_ABL(16) DLRM_F16P_ABL(5) DLRM_F16P_ABL(18) DLRM_F16P_ABL(7)
        DLRM_F16P_ABL(23) DLRM_F16P_ABL(32) DLRM_F16P_ABL(64)
        case 1000: hipLaunchKernelGGL((gemm_f16p_kernel<false, true, true, 4, 0, 3>), grid, block, 0, s, p); break;
        default: return DLRM_ERR_INVALID;
        }
        DLRM_LAUNCH_CHECK();
        *done = true;
        return DLRM_OK;
    }
#undef DLRM_F16P_ABL
#endif
    // DLRM_F16P_SHAPE=32: the 32 x 32 x 16 kernel (A/B aid); default: the 16 x 16 x 32 one
    static const int shape = [] { const char *e = getenv("DLRM_F16P_SHAPE"); return e ? atoi(e) : 16; }();
    if (shape == 16 && path == 0) {
        if (tail && transposed) hipLaunchKernelGGL((gemm_f16q_kernel<true, true>), grid, block, 0, s, p);
        else if (tail) hipLaunchKernelGGL((gemm_f16q_kernel<true, false>), grid, block, 0, s, p);
        else if (transposed) hipLaunchKernelGGL((gemm_f16q_kernel<false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f16q_kernel<false, false>), grid, block, 0, s, p);
    } else if (path == 0) DLRM_F16P_LAUNCH(true, true);
    else DLRM_F16P_LAUNCH(false, false);
#undef DLRM_F16P_LAUNCH
    DLRM_LAUNCH_CHECK();
    *done = true;
    return DLRM_OK;
}
// rows of an operand start on 16-byte boundaries (otherwise the kernel falls back to 4-byte loads)
static int vec_ok(const void *p, int64_t ld) { return aligned16(p) && ld % 4 == 0; }

// weight gradient [N,K] over a batch of M: tile and reduction split per evaluation mode
static TileShape wgrad_shape(int64_t M, int N, int K, int mode, int *splits) {
    if (mode == DLRM_GEMM_F32_MFMA) {
        *splits = splits_for((int)M, ceil_div(N, BM) * ceil_div(K, BN));
        return TileShape{BM, BN};
    }
    return pick_tile(N, K, M, splits);
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_absmax(const float *d_x, int64_t rows, int32_t cols, int64_t ld, float *d_slot, dlrm_stream_t stream) {
    DLRM_REQUIRE(rows >= 0 && cols >= 1 && ld >= cols && d_slot != nullptr, "bad arguments");
    if (rows == 0) return DLRM_OK;
    DLRM_REQUIRE(d_x != nullptr, "NULL operand");
    return launch_absmax(d_x, rows, cols, ld, d_slot, as_stream(stream));
}

extern "C" int dlrm_weight_planes_bytes(int32_t N, int32_t K, size_t *bytes) {
    DLRM_REQUIRE(bytes != nullptr && N >= 1 && K >= 1, "bad arguments");
    *bytes = plane_bytes_total(N, K);
    return DLRM_OK;
}

extern "C" int dlrm_split_weights(const dlrm_weight_planes *items, int32_t n, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 0 && (n == 0 || items != nullptr), "bad arguments");
    for (int i0 = 0; i0 < n; i0 += kAbsmaxMany) {
        SplitJobs jobs{};
        const int cnt = n - i0 < kAbsmaxMany ? n - i0 : kAbsmaxMany;
        int64_t most = 1;
        for (int i = 0; i < cnt; i++) {
            const dlrm_weight_planes &w = items[i0 + i];
            DLRM_REQUIRE(w.d_w && w.d_w_absmax && w.d_planes && w.N >= 1 && w.K >= 1 && w.w_ld >= w.K, "bad weight_planes item %d", i0 + i);
            DLRM_REQUIRE(aligned16(w.d_planes), "planes must be 16-byte aligned");
            SplitJob &j = jobs.j[i];
            j.w = w.d_w; j.ld = w.w_ld; j.N = w.N; j.K = w.K; j.amax = reinterpret_cast<const unsigned *>(w.d_w_absmax);
            j.d = plane_dims(w.N, w.K);
            j.fh = static_cast<_Float16 *>(w.d_planes); j.fl = j.fh + j.d.f_rows * j.d.f_ld;
            j.th = j.fl + j.d.f_rows * j.d.f_ld; j.tl = j.th + j.d.t_rows * j.d.t_ld;
            const int64_t b = ceil_div(j.d.f_rows * j.d.f_ld, 256 * 8);
            most = b > most ? b : most;
        }
        most = most > 256 ? 256 : most;
        hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)most, (unsigned)cnt), dim3(256), 0, as_stream(stream), jobs);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

extern "C" int dlrm_linear_forward(const dlrm_linear *l, dlrm_stream_t stream) {
    DLRM_REQUIRE(l != nullptr, "linear descriptor is NULL");
    DLRM_REQUIRE(l->M >= 0 && l->N >= 1 && l->K >= 1, "bad shape M=%lld N=%d K=%d", (long long)l->M, l->N, l->K);
    if (l->M == 0) return DLRM_OK;
    DLRM_REQUIRE(l->M < (1ll << 31), "M too large");
    DLRM_REQUIRE(l->d_x && l->d_w && l->d_y, "NULL operand");
    DLRM_REQUIRE(l->x_ld >= l->K && l->w_ld >= l->K && l->y_ld >= l->N, "leading dimension too small");
    // the two shapes that are no matrix product: a few input columns, or one output column
    if (skinny_k_forward_ok(l->N, l->K))
        return skinny_k_forward(l->d_x, l->x_ld, l->d_w, l->w_ld, l->d_bias, l->d_y, l->y_ld, (int)l->M, l->N, l->K,
                                l->act == DLRM_ACT_RELU, l->d_y_absmax, as_stream(stream));
    if (l->N == 1 && gemv_ok(l->d_x, l->x_ld, l->d_w, l->K))
        return gemv_forward(l->d_x, l->x_ld, l->d_w, l->d_bias, l->d_y, l->y_ld, (int)l->M, l->K, l->act == DLRM_ACT_RELU,
                            l->d_y_absmax, as_stream(stream));
    GemmParams p{};
    p.A = l->d_x; p.lda = l->x_ld; p.B = l->d_w; p.ldb = l->w_ld; p.C = l->d_y; p.ldc = l->y_ld;
    p.M = (int)l->M; p.N = l->N; p.K = l->K; p.bias = l->d_bias; p.relu = l->act == DLRM_ACT_RELU;
    p.a_vec = vec_ok(p.A, p.lda); p.b_vec = vec_ok(p.B, p.ldb);
    p.k_per_split = (int)ceil_div(l->K, BK) * BK;
    p.c_max = l->d_y_absmax;
    if (l->gemm_mode == DLRM_GEMM_F16X2) {
        if (int rc = operand_maxima(p, l->d_x_absmax, l->M, l->K, l->d_w_absmax, l->N, l->K, as_stream(stream))) return rc;
        bool done = false;
        if (int rc = launch_f16p(p, l->w_planes, false, l->d_w, l->w_ld, &done, as_stream(stream))) return rc;
        if (done) return DLRM_OK;
    }
    return launch_gemm<true, true>(p, 1, pick_tile(p.M, p.N, 0, nullptr), l->gemm_mode, as_stream(stream));
}

extern "C" int dlrm_linear_backward_data(const dlrm_linear_bwd *l, dlrm_stream_t stream) {
    DLRM_REQUIRE(l != nullptr, "linear_bwd descriptor is NULL");
    DLRM_REQUIRE(l->M >= 0 && l->N >= 1 && l->K >= 1, "bad shape");
    if (l->M == 0) return DLRM_OK;
    DLRM_REQUIRE(l->M < (1ll << 31), "M too large");
    DLRM_REQUIRE(l->d_dz && l->d_w && l->d_dx, "NULL operand");
    DLRM_REQUIRE(l->dz_ld >= l->N && l->w_ld >= l->K && l->dx_ld >= l->K, "leading dimension too small");
    if (l->N == 1 && gemv_ok(l->d_dx, l->dx_ld, l->d_w, l->K) &&
        (l->d_relu_mask == nullptr || gemv_ok(l->d_relu_mask, l->relu_mask_ld, l->d_w, l->K)))
        return gemv_backward_data(l->d_dz, l->dz_ld, l->d_w, l->d_dx, l->dx_ld, l->d_relu_mask, l->relu_mask_ld, (int)l->M,
                                  l->K, l->d_dx_absmax, as_stream(stream));
    GemmParams p{};
    p.A = l->d_dz; p.lda = l->dz_ld; p.B = l->d_w; p.ldb = l->w_ld; p.C = l->d_dx; p.ldc = l->dx_ld;
    p.M = (int)l->M; p.N = l->K; p.K = l->N;  // C = dx [M,K], reduction over N
    p.mask = l->d_relu_mask; p.mask_ld = l->relu_mask_ld;
    p.a_vec = vec_ok(p.A, p.lda); p.b_vec = vec_ok(p.B, p.ldb);
    p.k_per_split = (int)ceil_div(l->N, BK) * BK;
    p.c_max = l->d_dx_absmax;
    if (l->gemm_mode == DLRM_GEMM_F16X2) {    // (B = W [N,K]: rows are reduction indices)
        if (int rc = operand_maxima(p, l->d_dz_absmax, l->M, l->N, l->d_w_absmax, l->N, l->K, as_stream(stream))) return rc;
        bool done = false;
        if (int rc = launch_f16p(p, l->w_planes, true, l->d_w, l->w_ld, &done, as_stream(stream))) return rc;
        if (done) return DLRM_OK;
    }
    return launch_gemm<true, false>(p, 1, pick_tile(p.M, p.N, 0, nullptr), l->gemm_mode, as_stream(stream));
}

// row ranges the one-pass kernels of the narrow layers (few inputs, or one output) may cut the batch into: they
// stream dz once, so many short ranges cost little and keep the chip busy at small batches
static int skinny_chunk_cap(int64_t M) {
    const int64_t c = M / 32;
    return (int)(c < 1 ? 1 : (c > 512 ? 512 : c));
}

extern "C" int dlrm_linear_backward_weight_workspace_bytes(int64_t M, int32_t N, int32_t K, size_t *bytes) {
    DLRM_REQUIRE(bytes != nullptr && M >= 0 && N >= 1 && K >= 1, "bad arguments");
    // the split depends on the tile, which depends on the evaluation mode: room for either
    int s1 = 1, s2 = 1;
    wgrad_shape(M, N, K, DLRM_GEMM_BF16X3, &s1);
    wgrad_shape(M, N, K, DLRM_GEMM_F32_MFMA, &s2);
    int splits = s1 > s2 ? s1 : s2;
    if ((skinny_k_forward_ok(N, K) || N == 1) && skinny_chunk_cap(M) > splits) splits = skinny_chunk_cap(M);
    *bytes = sizeof(float) * ((size_t)splits * N * K + (size_t)splits * N) + 256;
    return DLRM_OK;
}

static int backward_weight_impl(const dlrm_linear_bwd *l, void *d_workspace, size_t workspace_bytes,
                                dlrm_stream_t stream, SlabReduce *pending = nullptr) {
    if (pending) pending->splits = 0;
    DLRM_REQUIRE(l != nullptr, "linear_bwd descriptor is NULL");
    DLRM_REQUIRE(l->M >= 1 && l->N >= 1 && l->K >= 1, "bad shape");
    DLRM_REQUIRE(l->M < (1ll << 31), "M too large");
    DLRM_REQUIRE(l->d_dz && l->d_x && l->d_dw, "NULL operand");
    DLRM_REQUIRE(l->dz_ld >= l->N && l->x_ld >= l->K && l->dw_ld >= l->K, "leading dimension too small");
    size_t need = 0;
    dlrm_linear_backward_weight_workspace_bytes(l->M, l->N, l->K, &need);
    DLRM_REQUIRE(d_workspace != nullptr && workspace_bytes >= need, "workspace too small: %zu < %zu", workspace_bytes,
                 need);
    DLRM_REQUIRE(aligned16(d_workspace), "workspace must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    int splits = 1;
    const TileShape ts = wgrad_shape(l->M, l->N, l->K, l->gemm_mode, &splits);
    float *slabs = static_cast<float *>(d_workspace);
    const bool few_inputs = skinny_k_forward_ok(l->N, l->K);
    if (few_inputs || (l->N == 1 && gemv_backward_weight_ok(l->d_x, l->x_ld, l->K))) {
        // partial sums over row ranges, then the same fixed-order reduction as the split-batch product
        const int cap = skinny_chunk_cap(l->M) > splits ? skinny_chunk_cap(l->M) : splits;   // (what the workspace query sized)
        float *dbp = l->d_db ? slabs + (size_t)cap * l->N * l->K : nullptr;
        int chunks = 0;
        const int rc = few_inputs ? skinny_k_backward_weight(l->d_dz, l->dz_ld, l->d_x, l->x_ld, slabs, dbp, (int)l->M, l->N,
                                                             l->K, cap, &chunks, s)
                                  : gemv_backward_weight(l->d_dz, l->dz_ld, l->d_x, l->x_ld, slabs, dbp, (int)l->M, l->K, cap,
                                                         &chunks, s);
        if (rc) return rc;
        return launch_reduce(slabs, (int64_t)l->N * l->K, l->N, l->K, l->d_dw, l->dw_ld, dbp, l->d_db, chunks, s, pending);
    }
    GemmParams p{};
    p.A = l->d_dz; p.lda = l->dz_ld; p.B = l->d_x; p.ldb = l->x_ld;
    p.M = l->N; p.N = l->K; p.K = (int)l->M;  // C = dW [N,K], reduction over the batch
    p.a_vec = vec_ok(p.A, p.lda); p.b_vec = vec_ok(p.B, p.ldb);
    p.k_per_split = (int)ceil_div(ceil_div(l->M, splits), BK) * BK;
    if (l->gemm_mode == DLRM_GEMM_F16X2)
        if (int rc = operand_maxima(p, l->d_dz_absmax, l->M, l->N, l->d_x_absmax, l->M, l->K, s)) return rc;
    float *db_partial = slabs + (size_t)splits * l->N * l->K;
    if (splits == 1) {
        p.C = l->d_dw; p.ldc = l->dw_ld;
        p.colsum = l->d_db;
        if (int rc = launch_gemm<false, false>(p, 1, ts, l->gemm_mode, s)) return rc;
    } else {
        p.C = slabs; p.ldc = l->K; p.slab_stride = (int64_t)l->N * l->K;
        p.colsum = l->d_db ? db_partial : nullptr;
        const int real_splits = (int)ceil_div(l->M, p.k_per_split);
        if (int rc = launch_gemm<false, false>(p, real_splits, ts, l->gemm_mode, s)) return rc;
        if (int rc = launch_reduce(slabs, p.slab_stride, l->N, l->K, l->d_dw, l->dw_ld, db_partial, l->d_db, real_splits, s,
                                   pending))
            return rc;
    }
    return DLRM_OK;
}

extern "C" int dlrm_linear_backward_weight(const dlrm_linear_bwd *l, void *d_workspace, size_t workspace_bytes,
                                           dlrm_stream_t stream) {
    return backward_weight_impl(l, d_workspace, workspace_bytes, stream);
}

int dlrm::linear_backward_weight_deferred(const dlrm_linear_bwd *l, void *d_workspace, size_t workspace_bytes,
                                          SlabReduce *pending, dlrm_stream_t stream) {
    DLRM_REQUIRE(pending != nullptr, "pending is NULL");
    return backward_weight_impl(l, d_workspace, workspace_bytes, stream, pending);
}

// ---- the same product with the partial-sum slabs in a buffer that is part of the loaded code object ----
// `scratch` below is the device array; in host code its name is only a handle for hipGetSymbolAddress
// (every GPU holds its own copy at its own address), so host code uses the looked-up pointer, never the name.
static __device__ __align__(256) unsigned char scratch[DLRM_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
static std::atomic<void *> g_scratch_ptr[kMaxDevices];   // [device] address of that device's copy of `scratch`, NULL = not asked yet

static int static_scratch(void **out) {
    int dev = -1;
    DLRM_HIP(hipGetDevice(&dev));
    DLRM_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        DLRM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        DLRM_REQUIRE(p != nullptr, "hipGetSymbolAddress returned NULL for the static scratch array");
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return DLRM_OK;
}

extern "C" int dlrm_linear_backward_weight_static(const dlrm_linear_bwd *l, dlrm_stream_t stream) {
    DLRM_REQUIRE(l != nullptr, "linear_bwd descriptor is NULL");
    DLRM_REQUIRE(l->M >= 1 && l->N >= 1 && l->K >= 1, "bad shape");
    size_t need = 0;
    if (int rc = dlrm_linear_backward_weight_workspace_bytes(l->M, l->N, l->K, &need)) return rc;
    DLRM_REQUIRE(need <= (size_t)DLRM_STATIC_SCRATCH_BYTES,
                 "layer M=%lld N=%d K=%d needs %zu bytes of partial sums, the static scratch holds %zu: use "
                 "dlrm_linear_backward_weight with a workspace", (long long)l->M, l->N, l->K, need,
                 (size_t)DLRM_STATIC_SCRATCH_BYTES);
    void *ws = nullptr;
    if (int rc = static_scratch(&ws)) return rc;
    return backward_weight_impl(l, ws, (size_t)DLRM_STATIC_SCRATCH_BYTES, stream);
}

extern "C" int dlrm_sgd_step(float *d_w, const float *d_g, int64_t n, float lr, dlrm_stream_t stream) {
    DLRM_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_w && d_g, "NULL operand");
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 2048 ? ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(sgd_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_w, d_g, n, lr);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}
