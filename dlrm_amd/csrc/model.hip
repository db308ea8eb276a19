// The whole DLRM train / inference step behind one handle, for hosts without a Python runtime: the C++ host of
// the training loop of dlrm_libtorch/src/dlrmtest.cpp (main(), :473-545) at the Criteo shapes -- EmbeddingBags
// over every table (struct emb, :144-210), bottom MLP, dot or concat interaction, top MLP (struct Net2,
// :217-278), loss, backward, SGD on the dense parameters, fused sparse update of the rows (:505-535).
//
// Host code only: every computation is one of the library's own C-ABI operators, launched on up to four
// streams -- the caller's ("main"), "emb" (EmbeddingBag kernels and their exchanges beside the bottom MLP),
// "prep" (the sparse update's sort, which needs the ids only, beside the forward pass) and "wgrad" (weight
// gradients beside the chain of data gradients at small batches).
//
// Sharded over the ranks of a dlrm_comm (one process per GPU, RCCL): every rank looks its tables up for the
// GLOBAL batch.  Whole tables: pooled vectors go to the rank that owns each batch slice with one all-to-all,
// their gradients come back with another.  Row-wise split tables: every rank pools the rows of its range, a
// reduce-scatter over the batch adds the partial sums, an all-gather brings the gradients of the global batch
// back.  The MLPs are data parallel (one all-reduce of the flat gradient buffer); losses are means over the
// global batch, so per-rank gradients add.  The reference is one process; the layout follows its striping of
// tables over parallel units (include/data2.h:150, src/emb_cache2.cpp:701-709).
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "skinny.h"

namespace dlrm {

static int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }
// leading dimension of a matrix whose columns are a GEMM reduction / output index: wide ones are whole
// 128-column tiles, so that every tile can use 16-byte loads (the pad columns are zero and never stored to)
static int64_t padk(int64_t n) { return n <= 128 ? pad4(n) : (n + 127) / 128 * 128; }

struct Layer {
    int N, K;
    int64_t w_off, w_ld, b_off;   // offsets into the flat parameter / gradient buffers
    int s_x = -1, s_w = -1, s_dz = -1;   // f16x2 evaluation: slots of max |input|, max |W|, max |dz| in dlrm_model::amax
    dlrm_weight_planes planes{};         // f16x2 evaluation: the matrix cut into its f16 pieces once per pass (d_planes NULL: not kept)
    size_t ws_off = 0, ws_bytes = 0;     // this layer's share of the weight-gradient workspace (partial sums wait there for the stack's one reduction)
};

struct Act {   // a [B, n] activation: pointer + leading dimension
    float *p = nullptr;
    int64_t ld = 0;
};

}  // namespace dlrm

using namespace dlrm;

struct dlrm_model {
    dlrm_model_config cfg;
    std::vector<int64_t> table_rows;
    std::vector<int> bottom, top, owner;
    int T, D, F, B, world, rank;
    int64_t Bg;
    bool sharded, dot;
    bool gather = false;     // one id per bag: the interaction kernels read the rows themselves
    bool gather_bwd = true;  // ... the backward kernel too (false: the forward kernel leaves a copy of the block for it)
    bool views = false;      // sharded: the interaction reads / writes the exchange buffers in place (no unpack / pack)
    std::vector<dlrm_feature_view> x_views, dx_views;
    // placement
    std::vector<int> my_tables, rw_tables, feature_order;
    std::vector<std::vector<int>> tables_of_rank;
    std::vector<int64_t> local_rows, tro, shard_begin_h;
    int Tl, Trw, n_tw;
    int64_t total_rows;
    // device memory
    std::vector<void *> allocs;
    void *weight = nullptr;
    float *opt_state = nullptr;
    int64_t *d_tro = nullptr, *d_shard_begin = nullptr;
    int32_t *err_flag = nullptr;
    float *params = nullptr, *grads = nullptr;
    int64_t n_params = 0;
    std::vector<Layer> bot, topl;
    float *x_dense = nullptr, *labels = nullptr, *feat = nullptr, *dfeat = nullptr, *inter = nullptr, *dinter = nullptr;
    int64_t x_ld = 0, feat_ld = 0, inter_ld = 0;
    int emb_col0 = 0, top_in = 0;
    std::vector<Act> bot_acts, bot_dz, top_acts, top_dz;
    float *pred = nullptr, *dpred = nullptr, *loss = nullptr;
    // f16x2 evaluation of the big layers: largest magnitudes of their operands, one slot per tensor, zeroed at the start
    // of every pass; amax_valid[slot] (host) = a kernel that fills the slot has been enqueued in this pass
    float *amax = nullptr;
    std::vector<char> amax_valid;
    bool f16x2 = false;
    bool fuse_last = false;  // a training step's last layer, loss gradient and that layer's backward-data pass are one launch
    void *ws_dense = nullptr, *ws_sparse = nullptr, *ws_sparse_rw = nullptr;
    size_t ws_dense_bytes = 0, ws_sparse_bytes = 0, ws_sparse_rw_bytes = 0;
    float *a2a_send = nullptr, *a2a_recv = nullptr, *g_send = nullptr, *g_recv = nullptr, *rw_part = nullptr, *rw_local = nullptr;
    std::vector<int64_t> send_counts, recv_counts;
    // streams
    hipStream_t s_emb = nullptr, s_prep = nullptr, s_wgrad = nullptr, s_reduce = nullptr;
    std::vector<hipEvent_t> events;
    size_t next_event = 0;
    // optional timing of operator classes: event pairs around the launches, on the launch stream
    struct TimeRec { int cls; hipEvent_t e0, e1; };
    std::vector<TimeRec> trecs;
    size_t n_trecs = 0;
    uint32_t time_mask = 0;
    // captured steps (cfg.graph_steps): one instantiated graph per set of batch buffers
    struct Captured { const void *idx, *off; int64_t n; hipStream_t stream; hipGraphExec_t exec; hipGraph_t graph; uint64_t used; };
    std::vector<Captured> graphs;
    uint64_t graph_clock = 0;
};

namespace dlrm {

template <typename T>
static int dalloc(dlrm_model *m, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    DLRM_HIP(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    m->allocs.push_back(q);
    if (zero) DLRM_HIP(hipMemset(q, 0, sizeof(T) * (count ? count : 1)));
    *p = static_cast<T *>(q);
    return DLRM_OK;
}

// stream a waits for everything queued on stream b so far
static int wait_for(dlrm_model *m, hipStream_t a, hipStream_t b) {
    if (a == b) return DLRM_OK;
    hipEvent_t e = m->events[m->next_event++ % m->events.size()];
    DLRM_HIP(hipEventRecord(e, b));
    DLRM_HIP(hipStreamWaitEvent(a, e, 0));
    return DLRM_OK;
}

// runs `fn` (one operator call on stream s); when its class is being timed, between two events on s
template <typename F>
static int timed(dlrm_model *m, int cls, hipStream_t s, F &&fn) {
    if (!(m->time_mask & (1u << cls)) || m->n_trecs >= m->trecs.size()) return fn();
    dlrm_model::TimeRec &r = m->trecs[m->n_trecs++];
    r.cls = cls;
    DLRM_HIP(hipEventRecord(r.e0, s));
    const int rc = fn();
    DLRM_HIP(hipEventRecord(r.e1, s));
    return rc;
}

static bool skinny_shape(int N, int K) { return (K <= 16 && N >= 64) || N == 1; }

static void build_stack(const std::vector<int> &sizes, int64_t &off, std::vector<Layer> &out) {
    for (size_t i = 0; i + 1 < sizes.size(); i++) {
        Layer l;
        l.K = sizes[i]; l.N = sizes[i + 1];
        l.w_ld = padk(l.K);
        l.w_off = off; off += (int64_t)l.N * l.w_ld;
        l.b_off = off; off += pad4(l.N);
        out.push_back(l);
    }
}

static dlrm_embbag table_desc(const dlrm_model *m, bool rw, const void *d_indices, const void *d_offsets, int64_t n_indices) {
    dlrm_embbag e{};
    const int esz = m->cfg.idx_dtype == DLRM_I64 ? 8 : 4;
    e.d_weight = m->weight;
    e.total_rows = m->total_rows > 0 ? m->total_rows : 1;
    e.dim = m->D; e.row_dtype = m->cfg.row_dtype; e.idx_dtype = m->cfg.idx_dtype;
    e.d_indices = d_indices; e.n_indices = n_indices; e.batch = m->Bg;
    e.include_last_offset = 1; e.pooling = m->cfg.pooling; e.d_error_flag = m->err_flag;
    if (!rw) {
        e.num_tables = m->Tl; e.d_table_row_offsets = m->d_tro; e.d_offsets = d_offsets;
    } else {
        e.num_tables = m->Trw; e.d_table_row_offsets = m->d_tro + m->Tl;
        e.d_offsets = static_cast<const char *>(d_offsets) + (int64_t)m->Tl * m->Bg * esz;
        e.d_shard_row_begin = m->d_shard_begin;
    }
    return e;
}

// ---- f16x2 evaluation: the operands' largest magnitudes travel with the tensors.  A GEMM's epilogue folds max |output|
// into the slot of the tensor it writes; tensors that come out of other kernels (the interaction, the narrow layers,
// the parameters after an SGD step) are measured by a pass of their own, enqueued right before their first reader.
static bool big_layer(const dlrm_model *m, const Layer &l) { return m->f16x2 && !skinny_shape(l.N, l.K); }

static int begin_pass_maxima(dlrm_model *m, hipStream_t s) {
    if (!m->f16x2) return DLRM_OK;
    DLRM_HIP(hipMemsetAsync(m->amax, 0, m->amax_valid.size() * sizeof(float), s));
    std::fill(m->amax_valid.begin(), m->amax_valid.end(), 0);
    // the weight matrices of the big layers, one launch (they change once per step, in the SGD kernel)
    std::vector<AbsmaxItem> items;
    for (auto *st : {&m->bot, &m->topl})
        for (const Layer &l : *st)
            if (big_layer(m, l)) {
                items.push_back(AbsmaxItem{m->params + l.w_off, l.N, l.K, l.w_ld, m->amax + l.s_w});
                m->amax_valid[l.s_w] = 1;
            }
    if (items.empty()) return DLRM_OK;
    if (int rc = absmax_many(items.data(), (int)items.size(), s)) return rc;
    // ... and their f16 pieces, one launch: the forward and backward-data products of this pass read them instead of
    // splitting the matrix again in every one of their row tiles
    std::vector<dlrm_weight_planes> planes;
    for (auto *st : {&m->bot, &m->topl})
        for (const Layer &l : *st)
            if (l.planes.d_planes) planes.push_back(l.planes);
    return planes.empty() ? DLRM_OK : dlrm_split_weights(planes.data(), (int32_t)planes.size(), s);
}

static int need_max(dlrm_model *m, int slot, const float *x, int cols, int64_t ld, int64_t rows, hipStream_t s) {
    if (slot < 0 || m->amax_valid[slot]) return DLRM_OK;
    m->amax_valid[slot] = 1;
    return dlrm_absmax(x, rows, cols, ld, m->amax + slot, s);
}

// slot of the tensor a GEMM writes (y_slot / dx_slot, -1: nobody needs it): returns the pointer and marks the slot filled
static float *out_slot(dlrm_model *m, int slot) {
    if (slot < 0) return nullptr;
    m->amax_valid[slot] = 1;
    return m->amax + slot;
}

static int linear_fwd(dlrm_model *m, const Layer &l, const float *x, int64_t x_ld, float *y, int64_t y_ld, bool relu,
                      hipStream_t s, int y_slot = -1) {
    dlrm_linear d{};
    d.M = m->B; d.N = l.N; d.K = l.K; d.d_x = x; d.x_ld = x_ld; d.d_w = m->params + l.w_off; d.w_ld = l.w_ld;
    d.d_bias = m->params + l.b_off; d.d_y = y; d.y_ld = y_ld; d.act = relu ? DLRM_ACT_RELU : DLRM_ACT_NONE;
    d.gemm_mode = m->cfg.gemm_mode;
    if (big_layer(m, l)) {
        if (int rc = need_max(m, l.s_x, x, l.K, x_ld, m->B, s)) return rc;
        if (int rc = need_max(m, l.s_w, d.d_w, l.K, l.w_ld, l.N, s)) return rc;
        d.gemm_mode = DLRM_GEMM_F16X2;
        d.d_x_absmax = m->amax + l.s_x; d.d_w_absmax = m->amax + l.s_w;
        if (l.planes.d_planes) d.w_planes = &l.planes;
    }
    if (m->f16x2) d.d_y_absmax = out_slot(m, y_slot);    // (the narrow layers' kernels leave their maximum behind too)
    return timed(m, skinny_shape(l.N, l.K) ? DLRM_TIME_LINEAR_FORWARD_SKINNY : DLRM_TIME_LINEAR_FORWARD, s,
                 [&] { return dlrm_linear_forward(&d, s); });
}

static int linear_dgrad(dlrm_model *m, const Layer &l, int64_t col0, int K, const float *dz, int64_t dz_ld, float *dx,
                        int64_t dx_ld, const float *mask, int64_t mask_ld, hipStream_t s, int dx_slot = -1) {
    dlrm_linear_bwd d{};
    d.M = m->B; d.N = l.N; d.K = K; d.d_dz = dz; d.dz_ld = dz_ld; d.d_w = m->params + l.w_off + col0; d.w_ld = l.w_ld;
    d.d_dx = dx; d.dx_ld = dx_ld; d.d_relu_mask = mask; d.relu_mask_ld = mask_ld; d.gemm_mode = m->cfg.gemm_mode;
    if (big_layer(m, l)) {
        if (int rc = need_max(m, l.s_dz, dz, l.N, dz_ld, m->B, s)) return rc;
        if (int rc = need_max(m, l.s_w, m->params + l.w_off, l.K, l.w_ld, l.N, s)) return rc;   // (of the whole matrix)
        d.gemm_mode = DLRM_GEMM_F16X2;
        d.d_dz_absmax = m->amax + l.s_dz; d.d_w_absmax = m->amax + l.s_w;
        if (l.planes.d_planes && col0 == 0 && K == l.K) d.w_planes = &l.planes;    // (a column slice of W splits in the kernel)
    }
    if (m->f16x2) d.d_dx_absmax = out_slot(m, dx_slot);
    return timed(m, l.N == 1 ? DLRM_TIME_LINEAR_BACKWARD_DATA_SKINNY : DLRM_TIME_LINEAR_BACKWARD_DATA, s,
                 [&] { return dlrm_linear_backward_data(&d, s); });
}

// (the maxima of dz and x are in their slots by now: stack_backward sees to dz, the forward pass to x)
// pending: the layer's partial sums stay in its workspace, to be added up by the stack's one reduce_slabs_many launch
static int linear_wgrad(dlrm_model *m, const Layer &l, const float *dz, int64_t dz_ld, const float *x, int64_t x_ld, hipStream_t s,
                        SlabReduce *pending) {
    dlrm_linear_bwd d{};
    d.M = m->B; d.N = l.N; d.K = l.K; d.d_dz = dz; d.dz_ld = dz_ld; d.d_x = x; d.x_ld = x_ld;
    d.d_dw = m->grads + l.w_off; d.dw_ld = l.w_ld; d.d_db = m->grads + l.b_off; d.gemm_mode = m->cfg.gemm_mode;
    if (big_layer(m, l)) {
        d.gemm_mode = DLRM_GEMM_F16X2;
        d.d_dz_absmax = m->amax_valid[l.s_dz] ? m->amax + l.s_dz : nullptr;
        d.d_x_absmax = m->amax_valid[l.s_x] ? m->amax + l.s_x : nullptr;
    }
    return timed(m, skinny_shape(l.N, l.K) ? DLRM_TIME_LINEAR_BACKWARD_WEIGHT_SKINNY : DLRM_TIME_LINEAR_BACKWARD_WEIGHT, s,
                 [&] {
                     // DLRM_WGRAD_DEFER=0 (A/B aid): every layer adds its partial sums up by itself
                     static const bool defer = [] { const char *e = getenv("DLRM_WGRAD_DEFER"); return !(e && e[0] == '0'); }();
                     if (!defer) {
                         pending->splits = 0;
                         return dlrm_linear_backward_weight(&d, m->ws_dense, m->ws_dense_bytes, s);   // (one area for all: as before)
                     }
                     return linear_backward_weight_deferred(&d, static_cast<char *>(m->ws_dense) + l.ws_off, l.ws_bytes, pending, s);
                 });
}

// Backward of one MLP stack (MLPStack.backward): the weight gradient of every layer goes to the wgrad stream
// (it needs the layer's dz only), the chain of data gradients stays on `s`.
static int stack_backward(dlrm_model *m, const std::vector<Layer> &st, const float *x_in, int64_t x_in_ld,
                          const std::vector<Act> &acts, const float *dz_last, int64_t dz_last_ld, const std::vector<Act> &dz_bufs,
                          float *dx_first, int64_t dx_first_ld, hipStream_t s, bool last_dgrad_done = false) {
    const int n = (int)st.size();
    hipStream_t sw = m->s_wgrad ? m->s_wgrad : s;
    std::vector<SlabReduce> pending((size_t)n);
    for (int i = n - 1; i >= 0; i--) {
        const float *dz = i == n - 1 ? dz_last : dz_bufs[i].p;
        const int64_t dz_ld = i == n - 1 ? dz_last_ld : dz_bufs[i].ld;
        const float *x = i == 0 ? x_in : acts[i - 1].p;
        const int64_t xl = i == 0 ? x_in_ld : acts[i - 1].ld;
        if (big_layer(m, st[i]))                          // (a dz that no GEMM wrote: measured here, ahead of both readers)
            if (int rc = need_max(m, st[i].s_dz, dz, st[i].N, dz_ld, m->B, s)) return rc;
        if (int rc = wait_for(m, sw, s)) return rc;      // dz of this layer is complete
        if (int rc = linear_wgrad(m, st[i], dz, dz_ld, x, xl, sw, &pending[(size_t)i])) return rc;
        if (i == n - 1 && last_dgrad_done) continue;      // (dz_bufs[n - 2] came with the loss gradient)
        if (i > 0) {
            if (int rc = linear_dgrad(m, st[i], 0, st[i].K, dz, dz_ld, dz_bufs[i - 1].p, dz_bufs[i - 1].ld, acts[i - 1].p,
                                      acts[i - 1].ld, s, big_layer(m, st[i - 1]) ? st[i - 1].s_dz : -1))
                return rc;
        } else if (dx_first) {
            if (int rc = linear_dgrad(m, st[0], 0, st[0].K, dz, dz_ld, dx_first, dx_first_ld, nullptr, 0, s)) return rc;
        }
    }
    // the partial sums of all layers of the stack, one launch (a reduction per layer is a string of 5-50 us launches
    // on the weight-gradient stream: 0.1 ms of a 0.9 ms step at a local batch of 8192)
    return timed(m, DLRM_TIME_LINEAR_BACKWARD_WEIGHT_REDUCE, sw, [&] { return reduce_slabs_many(pending.data(), n, sw); });
}

static int emb_forward(dlrm_model *m, const void *idx, const void *off, int64_t n_idx, hipStream_t s) {
    const int D = m->D;
    const int64_t B = m->B;
    if (!m->sharded) {
        dlrm_embbag e = table_desc(m, false, idx, off, n_idx);
        return timed(m, DLRM_TIME_EMBBAG_FORWARD, s, [&] { return dlrm_embbag_forward(&e, m->feat + m->emb_col0, m->feat_ld, s); });
    }
    if (m->Tl) {
        dlrm_embbag e = table_desc(m, false, idx, off, n_idx);
        if (int rc = timed(m, DLRM_TIME_EMBBAG_FORWARD, s, [&] { return dlrm_embbag_forward(&e, m->a2a_send, (int64_t)m->Tl * D, s); }))
            return rc;
    }
    if (int rc = timed(m, DLRM_TIME_COMM_FORWARD_EXCHANGE, s, [&] {
            return dlrm_comm_all_to_all(m->cfg.comm, m->a2a_send, m->send_counts.data(), m->a2a_recv, m->recv_counts.data(), s);
        }))
        return rc;
    // the block of every source rank into its columns of the feature block, one launch (not needed when the
    // interaction reads the blocks where they arrived)
    if (!m->views) {
        std::vector<dlrm_copy_item> items;
        int64_t o = 0;
        int first = 0;
        for (int j = 0; j < m->world; j++) {
            const int cnt = (int)m->tables_of_rank[j].size();
            if (cnt)
                items.push_back(dlrm_copy_item{m->feat + m->emb_col0 + (int64_t)first * D, m->feat_ld, m->a2a_recv + o,
                                               (int64_t)cnt * D, B, (int64_t)cnt * D});
            o += B * cnt * D;
            first += cnt;
        }
        if (int rc = dlrm_copy_2d_many(items.data(), (int32_t)items.size(), s)) return rc;
    }
    if (m->Trw) {
        dlrm_embbag e = table_desc(m, true, idx, off, n_idx);
        if (int rc = timed(m, DLRM_TIME_EMBBAG_FORWARD, s, [&] { return dlrm_embbag_forward(&e, m->rw_part, (int64_t)m->Trw * D, s); }))
            return rc;
        if (int rc = timed(m, DLRM_TIME_COMM_FORWARD_EXCHANGE, s, [&] {
                return dlrm_comm_reduce_scatter_sum(m->cfg.comm, m->rw_part, m->rw_local, B * m->Trw * D, s);
            }))
            return rc;
        if (!m->views)
            if (int rc = dlrm_copy_2d(m->feat + m->emb_col0 + (int64_t)m->n_tw * D, m->feat_ld, m->rw_local, (int64_t)m->Trw * D, B,
                                      (int64_t)m->Trw * D, s))
                return rc;
    }
    return DLRM_OK;
}

static dlrm_sparse_opt sparse_opt(const dlrm_model *m) {
    dlrm_sparse_opt o{};
    o.optimizer = m->cfg.emb_optimizer; o.dup_mode = DLRM_DUP_ACCUMULATE; o.lr = m->cfg.lr; o.eps = m->cfg.eps;
    o.d_opt_state = m->opt_state;
    return o;
}

static int sparse_update(dlrm_model *m, bool rw, const void *idx, const void *off, int64_t n_idx, const float *grad, int64_t grad_ld,
                         hipStream_t s) {
    dlrm_embbag e = table_desc(m, rw, idx, off, n_idx);
    dlrm_sparse_opt o = sparse_opt(m);
    void *ws = rw ? m->ws_sparse_rw : m->ws_sparse;
    const size_t wsb = rw ? m->ws_sparse_rw_bytes : m->ws_sparse_bytes;
    return timed(m, DLRM_TIME_EMBBAG_BACKWARD_UPDATE, s, [&] {
        if (m->cfg.presort) return dlrm_embbag_backward_apply(&e, grad, grad_ld, &o, ws, wsb, s);
        return dlrm_embbag_backward_update(&e, grad, grad_ld, &o, ws, wsb, s);
    });
}

static int emb_backward(dlrm_model *m, const void *idx, const void *off, int64_t n_idx, hipStream_t s) {
    const int D = m->D;
    const int64_t B = m->B;
    if (m->cfg.presort && m->s_prep)
        if (int rc = wait_for(m, s, m->s_prep)) return rc;
    if (!m->sharded) return sparse_update(m, false, idx, off, n_idx, m->dfeat + m->emb_col0, m->feat_ld, s);
    if (!m->views) {
        std::vector<dlrm_copy_item> items;
        int64_t o = 0;
        int first = 0;
        for (int j = 0; j < m->world; j++) {
            const int cnt = (int)m->tables_of_rank[j].size();
            if (cnt)
                items.push_back(dlrm_copy_item{m->g_send + o, (int64_t)cnt * D, m->dfeat + m->emb_col0 + (int64_t)first * D,
                                               m->feat_ld, B, (int64_t)cnt * D});
            o += B * cnt * D;
            first += cnt;
        }
        if (int rc = dlrm_copy_2d_many(items.data(), (int32_t)items.size(), s)) return rc;
    }
    if (int rc = timed(m, DLRM_TIME_COMM_BACKWARD_EXCHANGE, s, [&] {
            return dlrm_comm_all_to_all(m->cfg.comm, m->g_send, m->recv_counts.data(), m->g_recv, m->send_counts.data(), s);
        }))
        return rc;
    if (m->Tl)
        if (int rc = sparse_update(m, false, idx, off, n_idx, m->g_recv, (int64_t)m->Tl * D, s)) return rc;
    if (m->Trw) {
        if (!m->views)
            if (int rc = dlrm_copy_2d(m->rw_local, (int64_t)m->Trw * D, m->dfeat + m->emb_col0 + (int64_t)m->n_tw * D, m->feat_ld, B,
                                      (int64_t)m->Trw * D, s))
                return rc;
        if (int rc = timed(m, DLRM_TIME_COMM_BACKWARD_EXCHANGE, s, [&] {
                return dlrm_comm_all_gather(m->cfg.comm, m->rw_local, m->rw_part, B * m->Trw * D, s);
            }))
            return rc;
        if (int rc = sparse_update(m, true, idx, off, n_idx, m->rw_part, (int64_t)m->Trw * D, s)) return rc;
    }
    return DLRM_OK;
}

static int forward_pass(dlrm_model *m, const void *idx, const void *off, int64_t n_idx, bool train, hipStream_t main) {
    hipStream_t emb = m->s_emb ? m->s_emb : main;
    // the embedding side (lookups, and the exchange of a sharded model: the longest chain ahead of the interaction) needs
    // what the main stream holds at the call -- the caller's ids, earlier readers of its buffers -- and nothing of what
    // this pass puts there first (operand maxima, weight planes): its event is recorded before those
    if (!m->gather)
        if (int rc = wait_for(m, emb, main)) return rc;
    if (int rc = begin_pass_maxima(m, main)) return rc;
    // The sort of the batch's ids (first half of the sparse update) runs on its own stream from HERE on the device --
    // behind everything the main stream holds at this point -- but its dozen launches are handed to the driver only
    // after the main stream's first kernels: at small batches the host would otherwise keep the main stream waiting
    // for as long as it takes to enqueue them.
    hipStream_t prep = m->s_prep ? m->s_prep : main;
    hipEvent_t prep_from = nullptr;
    bool prep_todo = train && m->cfg.presort;
    if (prep_todo && prep != main) {
        prep_from = m->events[m->next_event++ % m->events.size()];
        DLRM_HIP(hipEventRecord(prep_from, main));
    }
    auto launch_prepare = [&]() -> int {
        if (!prep_todo) return DLRM_OK;
        prep_todo = false;
        if (prep_from) DLRM_HIP(hipStreamWaitEvent(prep, prep_from, 0));
        if (m->Tl) {
            dlrm_embbag e = table_desc(m, false, idx, off, n_idx);
            if (int rc = timed(m, DLRM_TIME_EMBBAG_BACKWARD_PREPARE, prep,
                               [&] { return dlrm_embbag_backward_prepare(&e, m->ws_sparse, m->ws_sparse_bytes, prep); }))
                return rc;
        }
        if (m->Trw && m->sharded) {
            dlrm_embbag e = table_desc(m, true, idx, off, n_idx);
            if (int rc = timed(m, DLRM_TIME_EMBBAG_BACKWARD_PREPARE, prep,
                               [&] { return dlrm_embbag_backward_prepare(&e, m->ws_sparse_rw, m->ws_sparse_rw_bytes, prep); }))
                return rc;
        }
        return DLRM_OK;
    };
    if (prep == main)      // (no stream of its own: program order is device order)
        if (int rc = launch_prepare()) return rc;
    if (!m->gather)
        if (int rc = emb_forward(m, idx, off, n_idx, emb)) return rc;
    // bottom MLP on main
    const float *x = m->x_dense;
    int64_t xl = m->x_ld;
    for (size_t i = 0; i < m->bot.size(); i++) {
        const int y_slot = i + 1 < m->bot.size() && big_layer(m, m->bot[i + 1]) ? m->bot[i + 1].s_x : -1;
        if (int rc = linear_fwd(m, m->bot[i], x, xl, m->bot_acts[i].p, m->bot_acts[i].ld, true, main, y_slot)) return rc;
        x = m->bot_acts[i].p; xl = m->bot_acts[i].ld;
    }
    if (m->gather) {
        // one id per bag: feature f >= 1 of a sample is a table row, read inside the interaction kernel
        dlrm_embbag e = table_desc(m, false, idx, off, n_idx);
        if (int rc = timed(m, DLRM_TIME_INTERACT_DOT_FORWARD, main, [&] {
                // (a training step whose backward pass reads the block -- not the one that gathers the rows again --
                // gets it written in place around feature 0)
                float *out_max = big_layer(m, m->topl[0]) ? out_slot(m, m->topl[0].s_x) : nullptr;
                return interact_dot_forward_gather_max(&e, m->feat, m->feat_ld, m->inter, m->inter_ld,
                                                       train && !m->gather_bwd ? m->feat : nullptr, out_max, main);
            }))
            return rc;
        if (int rc = launch_prepare()) return rc;
    } else {
        // (the event goes to the embedding stream ahead of the sort's launches: two streams may share a hardware queue,
        //  a queue runs its packets in order, and an event recorded behind the sort would make the interaction wait
        //  for the sort as well as for the exchange)
        if (int rc = wait_for(m, main, emb)) return rc;
        if (m->dot)
            if (int rc = timed(m, DLRM_TIME_INTERACT_DOT_FORWARD, main, [&] {
                    if (m->views) {
                        float *out_max = big_layer(m, m->topl[0]) ? out_slot(m, m->topl[0].s_x) : nullptr;
                        return interact_dot_forward_views_max(m->x_views.data(), m->B, m->F, m->D, m->inter, m->inter_ld, out_max, main);
                    }
                    return dlrm_interact_dot_forward(m->feat, m->B, m->F, m->D, m->inter, m->inter_ld, main);
                }))
                return rc;
        if (int rc = launch_prepare()) return rc;
    }
    x = m->inter; xl = m->inter_ld;
    // (the last layer of a training step that fuses it with the loss gradient is launched by the step)
    const size_t n_top = m->topl.size() - (train && m->fuse_last ? 1 : 0);
    for (size_t i = 0; i < n_top; i++) {
        const bool relu = m->cfg.final_relu || i + 1 < m->topl.size();
        const int y_slot = i + 1 < m->topl.size() && big_layer(m, m->topl[i + 1]) ? m->topl[i + 1].s_x : -1;
        if (int rc = linear_fwd(m, m->topl[i], x, xl, m->top_acts[i].p, m->top_acts[i].ld, relu, main, y_slot)) return rc;
        x = m->top_acts[i].p; xl = m->top_acts[i].ld;
    }
    return DLRM_OK;
}

// plan_table_wise of the Python host: biggest table first, to the rank with the fewest lookups so far, ties to
// the one holding the fewest bytes, then the lower rank
static void greedy_owner(const int64_t *rows, int T, int world, std::vector<int> &owner) {
    std::vector<int> order(T);
    for (int t = 0; t < T; t++) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a] > rows[b]; });
    std::vector<double> mem(world, 0.0), lk(world, 0.0);
    owner.assign(T, 0);
    for (int t : order) {
        int best = 0;
        for (int r = 1; r < world; r++)
            if (lk[r] < lk[best] || (lk[r] == lk[best] && mem[r] < mem[best])) best = r;
        owner[t] = best;
        mem[best] += (double)rows[t];
        lk[best] += 1.0;
    }
}

}  // namespace dlrm

extern "C" int dlrm_model_destroy(dlrm_model *m) {
    if (!m) return DLRM_OK;
    (void)hipDeviceSynchronize();
    for (void *p : m->allocs) (void)hipFree(p);
    for (auto &g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
    for (auto &r : m->trecs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    if (m->s_emb) (void)hipStreamDestroy(m->s_emb);
    if (m->s_prep) (void)hipStreamDestroy(m->s_prep);
    if (m->s_wgrad) (void)hipStreamDestroy(m->s_wgrad);
    if (m->s_reduce) (void)hipStreamDestroy(m->s_reduce);
    delete m;
    return DLRM_OK;
}

extern "C" int dlrm_model_create(const dlrm_model_config *c, dlrm_model **out) {
    DLRM_REQUIRE(c != nullptr && out != nullptr, "NULL argument");
    DLRM_REQUIRE(c->num_tables >= 1 && c->table_rows != nullptr && c->dim >= 1 && c->dense_in >= 1, "bad table / feature sizes");
    DLRM_REQUIRE(c->n_bottom >= 1 && c->bottom != nullptr && c->n_top >= 1 && c->top != nullptr, "bad MLP description");
    DLRM_REQUIRE(c->top[c->n_top - 1] == 1, "the top MLP must end in one output");
    DLRM_REQUIRE(c->local_batch >= 1 && c->local_batch < (1ll << 31), "bad local_batch");
    DLRM_REQUIRE(c->world >= 1 && c->rank >= 0 && c->rank < c->world, "rank %d of %d", c->rank, c->world);
    DLRM_REQUIRE(c->interaction == DLRM_INTERACT_DOT || c->interaction == DLRM_INTERACT_CAT, "interaction %d", c->interaction);
    DLRM_REQUIRE(c->loss == DLRM_LOSS_BCE || c->loss == DLRM_LOSS_MSE, "loss %d", c->loss);
    DLRM_REQUIRE(c->row_dtype == DLRM_F32 || c->row_dtype == DLRM_BF16, "row_dtype %d", c->row_dtype);
    DLRM_REQUIRE(c->idx_dtype == DLRM_I32 || c->idx_dtype == DLRM_I64, "idx_dtype %d", c->idx_dtype);
    DLRM_REQUIRE(c->interaction != DLRM_INTERACT_DOT || c->bottom[c->n_bottom - 1] == c->dim,
                 "the dot interaction needs the bottom MLP to end in dim outputs");
    const bool sharded = c->world > 1 || c->force_exchange;
    DLRM_REQUIRE(!sharded || c->comm != nullptr, "a sharded model needs a communicator");

    dlrm_model *m = new dlrm_model();
    m->cfg = *c;
    m->T = c->num_tables; m->D = c->dim; m->F = m->T + 1; m->B = (int)c->local_batch; m->world = c->world; m->rank = c->rank;
    m->Bg = (int64_t)m->B * m->world;
    m->sharded = sharded; m->dot = c->interaction == DLRM_INTERACT_DOT;
    m->gather = c->one_id_per_bag && !sharded && m->dot && c->row_dtype == DLRM_F32 && c->num_tables + 1 <= 32 &&
                (c->dim & (c->dim - 1)) == 0 && c->dim >= 8 && (int64_t)(c->num_tables + 1) * c->dim <= 4096;
    {   // DLRM_GATHER_BWD=0: the forward kernel leaves a copy of the gathered block for the backward pass (for A/B timing)
        const char *e = getenv("DLRM_GATHER_BWD");
        m->gather_bwd = !(e && e[0] == '0');
    }
    m->table_rows.assign(c->table_rows, c->table_rows + m->T);
    m->bottom.assign(c->bottom, c->bottom + c->n_bottom);
    m->top.assign(c->top, c->top + c->n_top);
    m->cfg.table_rows = nullptr; m->cfg.bottom = nullptr; m->cfg.top = nullptr; m->cfg.table_owner = nullptr;
    auto fail = [&](int rc) { dlrm_model_destroy(m); return rc; };

    // ---- placement
    if (c->table_owner) m->owner.assign(c->table_owner, c->table_owner + m->T);
    else greedy_owner(c->table_rows, m->T, m->world, m->owner);
    m->tables_of_rank.assign(m->world, {});
    for (int t = 0; t < m->T; t++) {
        if (m->owner[t] < -1 || m->owner[t] >= m->world) return fail(set_error(DLRM_ERR_INVALID, "table %d: owner %d", t, m->owner[t]));
        if (m->owner[t] >= 0) m->tables_of_rank[m->owner[t]].push_back(t);
        else m->rw_tables.push_back(t);
    }
    for (int r = 0; r < m->world; r++)
        for (int t : m->tables_of_rank[r]) m->feature_order.push_back(t);
    m->n_tw = (int)m->feature_order.size();
    for (int t : m->rw_tables) m->feature_order.push_back(t);
    if (!sharded) {
        if (!m->rw_tables.empty()) return fail(set_error(DLRM_ERR_INVALID, "row-wise tables need a sharded model"));
        m->my_tables = m->feature_order;
    } else {
        m->my_tables = m->tables_of_rank[m->rank];
    }
    m->Tl = (int)m->my_tables.size();
    m->Trw = (int)m->rw_tables.size();
    m->tro.push_back(0);
    for (int t : m->my_tables) { m->local_rows.push_back(m->table_rows[t]); m->shard_begin_h.push_back(0); }
    for (int t : m->rw_tables) {
        const int64_t rows = m->table_rows[t];
        if (rows < m->world) return fail(set_error(DLRM_ERR_INVALID, "row-wise table %d has fewer rows than ranks", t));
        const int64_t lo = rows * m->rank / m->world, hi = rows * (m->rank + 1) / m->world;
        m->local_rows.push_back(hi - lo);
        m->shard_begin_h.push_back(lo);
    }
    for (int64_t r : m->local_rows) m->tro.push_back(m->tro.back() + r);
    m->total_rows = m->tro.back();

    // ---- streams and events
    // (single GPU: the sparse update beside the bottom MLP's backward pass -- an HBM-bound kernel beside matrix-bound
    //  ones: local batch 8192 and 16384 +2.6 %; Criteo-Kaggle's 2048 -5 %, so not below 8192; 65536 +1.2 %, not taken
    //  by default: the kernels that share the chip then take 15-25 % longer each and per-operator times stop
    //  describing the operators)
    if (c->overlap < 0 ? (sharded || (c->local_batch >= 8192 && c->local_batch <= 32768)) : c->overlap != 0)
        if (hipStreamCreateWithFlags(&m->s_emb, hipStreamNonBlocking) != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "stream"));
    m->cfg.presort = c->presort != 0;
    if (m->cfg.presort)
        if (hipStreamCreateWithFlags(&m->s_prep, hipStreamNonBlocking) != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "stream"));
    if (c->wgrad_stream < 0 ? c->local_batch <= 32768 : c->wgrad_stream != 0)
        if (hipStreamCreateWithFlags(&m->s_wgrad, hipStreamNonBlocking) != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "stream"));
    if (sharded && c->comm_dense != nullptr)      // the dense gradients' all-reduce beside the rest of the backward pass
        if (hipStreamCreateWithFlags(&m->s_reduce, hipStreamNonBlocking) != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "stream"));
    m->events.resize(64);
    for (auto &e : m->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "event"));

    // ---- tables
    const int esz = c->row_dtype == DLRM_F32 ? 4 : 2;
    int rc = DLRM_OK;
    {
        char *w = nullptr;
        rc = dalloc(m, &w, (size_t)(m->total_rows > 0 ? m->total_rows : 1) * m->D * esz, false);
        if (rc) return fail(rc);
        m->weight = w;
    }
    if ((rc = dalloc(m, &m->d_tro, m->tro.size()))) return fail(rc);
    if (hipMemcpy(m->d_tro, m->tro.data(), sizeof(int64_t) * m->tro.size(), hipMemcpyHostToDevice) != hipSuccess)
        return fail(set_error(DLRM_ERR_HIP, "copy of the row offsets failed"));
    if (m->Trw) {
        if ((rc = dalloc(m, &m->d_shard_begin, (size_t)m->Trw))) return fail(rc);
        if (hipMemcpy(m->d_shard_begin, m->shard_begin_h.data() + m->Tl, sizeof(int64_t) * m->Trw, hipMemcpyHostToDevice) != hipSuccess)
            return fail(set_error(DLRM_ERR_HIP, "copy of the shard ranges failed"));
    }
    if (c->init_tables)
        for (int lt = 0; lt < m->Tl + m->Trw; lt++) {
            // every table is its own stream of uniform [0,1) values keyed by the GLOBAL table id, so a sharded
            // model holds the same numbers as a single-GPU one
            const int t = lt < m->Tl ? m->my_tables[lt] : m->rw_tables[lt - m->Tl];
            rc = dlrm_table_init_uniform(static_cast<char *>(m->weight) + m->tro[lt] * m->D * esz, c->row_dtype,
                                         m->local_rows[lt] * m->D, (c->seed << 20) + 7919ull * (uint64_t)(t + 1),
                                         m->shard_begin_h[lt] * m->D, nullptr);
            if (rc) return fail(rc);
        }
    if (c->emb_optimizer == DLRM_OPT_ADAGRAD_ROWWISE)
        if ((rc = dalloc(m, &m->opt_state, (size_t)(m->total_rows > 0 ? m->total_rows : 1)))) return fail(rc);
    if ((rc = dalloc(m, &m->err_flag, 1))) return fail(rc);

    // ---- dense parameters: one flat buffer and its twin for the gradients
    m->top_in = m->dot ? m->D + m->F * (m->F - 1) / 2 : m->bottom.back() + m->T * m->D;
    std::vector<int> bsz{c->dense_in}, tsz{m->top_in};
    bsz.insert(bsz.end(), m->bottom.begin(), m->bottom.end());
    tsz.insert(tsz.end(), m->top.begin(), m->top.end());
    int64_t off = 0;
    build_stack(bsz, off, m->bot);
    build_stack(tsz, off, m->topl);
    m->n_params = off;
    // the big layers' products on f16 pieces when the batch is large enough for the matrix pipe to be the limit (local
    // batch 8192: +2 %, 16384: +12 %, 65536: +29 %; at Criteo-Kaggle's 2048 the layers are short launches whose time
    // the measuring passes only add to: -6 %)
    m->f16x2 = c->gemm_mode == DLRM_GEMM_F16X2 || (c->gemm_mode == DLRM_GEMM_DEFAULT && m->B >= 8192);
    {
        int n_slots = 0;
        for (auto *st : {&m->bot, &m->topl})
            for (Layer &l : *st) { l.s_x = n_slots++; l.s_w = n_slots++; l.s_dz = n_slots++; }
        m->amax_valid.assign((size_t)n_slots, 0);
        if ((rc = dalloc(m, &m->amax, (size_t)n_slots))) return fail(rc);
    }
    if ((rc = dalloc(m, &m->params, (size_t)off))) return fail(rc);
    if ((rc = dalloc(m, &m->grads, (size_t)off))) return fail(rc);
    if (m->f16x2)
        for (auto *st : {&m->bot, &m->topl})
            for (Layer &l : *st)
                if (big_layer(m, l)) {
                    size_t bytes = 0;
                    if ((rc = dlrm_weight_planes_bytes(l.N, l.K, &bytes))) return fail(rc);
                    unsigned char *pl = nullptr;
                    if ((rc = dalloc(m, &pl, bytes))) return fail(rc);
                    l.planes = dlrm_weight_planes{m->params + l.w_off, l.w_ld, l.N, l.K, m->amax + l.s_w, pl};
                }

    // ---- activations
    const size_t B = (size_t)m->B;
    m->x_ld = pad4(c->dense_in);
    if ((rc = dalloc(m, &m->x_dense, B * m->x_ld))) return fail(rc);
    if ((rc = dalloc(m, &m->labels, B))) return fail(rc);
    m->inter_ld = padk(m->top_in);
    if (m->dot) {
        m->feat_ld = (int64_t)m->F * m->D;
        m->emb_col0 = m->D;
        if ((rc = dalloc(m, &m->feat, B * m->feat_ld))) return fail(rc);
        if ((rc = dalloc(m, &m->dfeat, B * m->feat_ld))) return fail(rc);
        if ((rc = dalloc(m, &m->inter, B * m->inter_ld))) return fail(rc);
        if ((rc = dalloc(m, &m->dinter, B * m->inter_ld))) return fail(rc);
    } else {
        m->feat_ld = m->inter_ld;
        m->emb_col0 = m->bottom.back();
        if ((rc = dalloc(m, &m->feat, B * m->feat_ld))) return fail(rc);
        if ((rc = dalloc(m, &m->dfeat, B * m->feat_ld))) return fail(rc);
        m->inter = m->feat; m->dinter = m->dfeat;
    }
    auto make_acts = [&](const std::vector<int> &widths, std::vector<Act> &acts, size_t count) {
        for (size_t i = 0; i < count; i++) {
            Act a;
            a.ld = padk(widths[i]);
            if (int r = dalloc(m, &a.p, B * a.ld)) return r;
            acts.push_back(a);
        }
        return (int)DLRM_OK;
    };
    if ((rc = make_acts(m->bottom, m->bot_acts, m->bottom.size() - 1))) return fail(rc);
    m->bot_acts.push_back(Act{m->feat, m->feat_ld});            // the last bottom layer writes feature 0 / the concat's head
    if ((rc = make_acts(m->bottom, m->bot_dz, m->bottom.size() - 1))) return fail(rc);
    if ((rc = make_acts(m->top, m->top_acts, m->top.size() - 1))) return fail(rc);
    if ((rc = dalloc(m, &m->pred, B))) return fail(rc);
    m->top_acts.push_back(Act{m->pred, 1});
    if ((rc = make_acts(m->top, m->top_dz, m->top.size() - 1))) return fail(rc);
    if ((rc = dalloc(m, &m->dpred, B))) return fail(rc);
    if ((rc = dalloc(m, &m->loss, 1))) return fail(rc);
    {   // DLRM_FUSE_LAST=0 keeps the last layer, the loss gradient and the layer's backward-data pass apart (for A/B timing)
        const char *e = getenv("DLRM_FUSE_LAST");
        const size_t n = m->topl.size();
        m->fuse_last = !(e && e[0] == '0') && n >= 2 && m->topl[n - 1].N == 1 &&
                       gemv_forward_loss_backward_ok(m->top_acts[n - 2].p, m->top_acts[n - 2].ld, m->params + m->topl[n - 1].w_off,
                                                     m->top_dz[n - 2].p, m->top_dz[n - 2].ld, m->topl[n - 1].K);
    }

    // ---- workspaces: the largest any layer / any batch of the configured shape needs
    // (every layer its own share: the partial sums of a stack's layers are added up together, behind the last of them)
    size_t need = 0;
    for (auto *st : {&m->bot, &m->topl})
        for (Layer &l : *st) {
            size_t b = 0;
            if ((rc = dlrm_linear_backward_weight_workspace_bytes(m->B, l.N, l.K, &b))) return fail(rc);
            l.ws_off = need;
            l.ws_bytes = b;
            need += (b + 255) / 256 * 256;
        }
    m->ws_dense_bytes = need;
    {
        char *p = nullptr;
        if ((rc = dalloc(m, &p, need, false))) return fail(rc);
        m->ws_dense = p;
    }

    // ---- exchange buffers
    if (sharded) {
        const int64_t D = m->D;
        m->send_counts.assign(m->world, (int64_t)m->B * m->Tl * D);
        for (int j = 0; j < m->world; j++) m->recv_counts.push_back((int64_t)m->B * (int64_t)m->tables_of_rank[j].size() * D);
        if ((rc = dalloc(m, &m->a2a_send, (size_t)(m->Bg * m->Tl * D)))) return fail(rc);
        if ((rc = dalloc(m, &m->a2a_recv, (size_t)((int64_t)m->B * m->n_tw * D)))) return fail(rc);
        if ((rc = dalloc(m, &m->g_send, (size_t)((int64_t)m->B * m->n_tw * D)))) return fail(rc);
        if ((rc = dalloc(m, &m->g_recv, (size_t)(m->Bg * m->Tl * D)))) return fail(rc);
        if (m->Trw) {
            if ((rc = dalloc(m, &m->rw_part, (size_t)(m->Bg * m->Trw * D)))) return fail(rc);
            if ((rc = dalloc(m, &m->rw_local, (size_t)((int64_t)m->B * m->Trw * D)))) return fail(rc);
        }
    }
    if (sharded && m->dot && (m->D & (m->D - 1)) == 0 && m->D >= 8 && m->F <= 32 && m->F * m->D <= 4096 && !getenv("DLRM_NO_VIEWS")) {
        // feature f of the interaction, where the exchanges leave it / where its gradient has to go
        m->views = true;
        const int64_t D = m->D;
        m->x_views.push_back(dlrm_feature_view{m->feat, m->feat_ld});
        m->dx_views.push_back(dlrm_feature_view{m->dfeat, m->feat_ld});
        int64_t o = 0;
        for (int j = 0; j < m->world; j++) {
            const int64_t cnt = (int64_t)m->tables_of_rank[j].size();
            for (int64_t k = 0; k < cnt; k++) {
                m->x_views.push_back(dlrm_feature_view{m->a2a_recv + o + k * D, cnt * D});
                m->dx_views.push_back(dlrm_feature_view{m->g_send + o + k * D, cnt * D});
            }
            o += (int64_t)m->B * cnt * D;
        }
        for (int64_t k = 0; k < m->Trw; k++) {      // row-wise tables: pooled sums in, gradients out, the same buffer
            m->x_views.push_back(dlrm_feature_view{m->rw_local + k * D, (int64_t)m->Trw * D});
            m->dx_views.push_back(dlrm_feature_view{m->rw_local + k * D, (int64_t)m->Trw * D});
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(set_error(DLRM_ERR_HIP, "model initialisation failed"));
    *out = m;
    return DLRM_OK;
}

extern "C" int dlrm_model_info(const dlrm_model *m, dlrm_model_info_t *info) {
    DLRM_REQUIRE(m != nullptr && info != nullptr, "NULL argument");
    info->d_params = m->params; info->d_grads = m->grads; info->n_params = m->n_params;
    info->d_weight = m->weight; info->total_rows = m->total_rows; info->d_opt_state = m->opt_state;
    info->num_local_tables = m->Tl + m->Trw; info->num_whole_tables = m->Tl;
    info->d_pred = m->pred; info->d_loss = m->loss; info->d_error_flag = m->err_flag;
    info->d_dense = m->x_dense; info->dense_ld = m->x_ld; info->d_labels = m->labels;
    info->n_bottom = (int32_t)m->bot.size(); info->n_top = (int32_t)m->topl.size();
    return DLRM_OK;
}

extern "C" int dlrm_model_local_table(const dlrm_model *m, int32_t local_table, int32_t *global_table, int64_t *first_pool_row,
                                      int64_t *rows, int64_t *first_table_row) {
    DLRM_REQUIRE(m != nullptr && local_table >= 0 && local_table < m->Tl + m->Trw, "local table out of range");
    if (global_table) *global_table = local_table < m->Tl ? m->my_tables[local_table] : m->rw_tables[local_table - m->Tl];
    if (first_pool_row) *first_pool_row = m->tro[local_table];
    if (rows) *rows = m->local_rows[local_table];
    if (first_table_row) *first_table_row = m->shard_begin_h[local_table];
    return DLRM_OK;
}

extern "C" int dlrm_model_feature_order(const dlrm_model *m, int32_t *order) {
    DLRM_REQUIRE(m != nullptr && order != nullptr, "NULL argument");
    for (int i = 0; i < m->T; i++) order[i] = m->feature_order[i];
    return DLRM_OK;
}

static const Layer *find_layer(const dlrm_model *m, int32_t stack, int32_t layer) {
    const std::vector<Layer> &st = stack == 0 ? m->bot : m->topl;
    if (stack < 0 || stack > 1 || layer < 0 || layer >= (int)st.size()) return nullptr;
    return &st[layer];
}

extern "C" int dlrm_model_layer(const dlrm_model *m, int32_t stack, int32_t layer, int32_t *N, int32_t *K, int64_t *w_offset,
                                int64_t *w_ld, int64_t *b_offset) {
    DLRM_REQUIRE(m != nullptr, "model is NULL");
    const Layer *l = find_layer(m, stack, layer);
    DLRM_REQUIRE(l != nullptr, "no layer %d in stack %d", layer, stack);
    if (N) *N = l->N;
    if (K) *K = l->K;
    if (w_offset) *w_offset = l->w_off;
    if (w_ld) *w_ld = l->w_ld;
    if (b_offset) *b_offset = l->b_off;
    return DLRM_OK;
}

extern "C" int dlrm_model_set_layer(dlrm_model *m, int32_t stack, int32_t layer, const float *h_w, const float *h_b) {
    DLRM_REQUIRE(m != nullptr && h_w != nullptr && h_b != nullptr, "NULL argument");
    const Layer *l = find_layer(m, stack, layer);
    DLRM_REQUIRE(l != nullptr, "no layer %d in stack %d", layer, stack);
    DLRM_HIP(hipMemcpy2D(m->params + l->w_off, sizeof(float) * l->w_ld, h_w, sizeof(float) * l->K, sizeof(float) * l->K, l->N,
                         hipMemcpyHostToDevice));
    DLRM_HIP(hipMemcpy(m->params + l->b_off, h_b, sizeof(float) * l->N, hipMemcpyHostToDevice));
    return DLRM_OK;
}

extern "C" int dlrm_model_get_layer(const dlrm_model *m, int32_t stack, int32_t layer, float *h_w, float *h_b) {
    DLRM_REQUIRE(m != nullptr && h_w != nullptr && h_b != nullptr, "NULL argument");
    const Layer *l = find_layer(m, stack, layer);
    DLRM_REQUIRE(l != nullptr, "no layer %d in stack %d", layer, stack);
    DLRM_HIP(hipDeviceSynchronize());
    DLRM_HIP(hipMemcpy2D(h_w, sizeof(float) * l->K, m->params + l->w_off, sizeof(float) * l->w_ld, sizeof(float) * l->K, l->N,
                         hipMemcpyDeviceToHost));
    DLRM_HIP(hipMemcpy(h_b, m->params + l->b_off, sizeof(float) * l->N, hipMemcpyDeviceToHost));
    return DLRM_OK;
}

// Xavier-style normal init of every layer from one seed (the same on every rank: replicas start identical);
// the reference binds unscaled randn weights (src/dlrmtest.cpp:110-113)
extern "C" int dlrm_model_init_dense(dlrm_model *m, uint64_t seed) {
    DLRM_REQUIRE(m != nullptr, "model is NULL");
    uint64_t s = splitmix64(seed ^ 0x6d6c705f696e6974ull);
    auto uni = [&]() { s = splitmix64(s); return (float)(s >> 40) * (1.0f / 16777216.0f); };
    auto nrm = [&]() { const float u1 = uni() + 1e-7f, u2 = uni(); return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2); };
    for (int stack = 0; stack < 2; stack++) {
        const std::vector<Layer> &st = stack == 0 ? m->bot : m->topl;
        for (size_t i = 0; i < st.size(); i++) {
            const Layer &l = st[i];
            std::vector<float> w((size_t)l.N * l.K), b(l.N);
            const float sw = std::sqrt(2.0f / (float)(l.K + l.N)), sb = std::sqrt(1.0f / (float)l.N);
            for (auto &x : w) x = nrm() * sw;
            for (auto &x : b) x = nrm() * sb;
            if (int rc = dlrm_model_set_layer(m, stack, (int)i, w.data(), b.data())) return rc;
        }
    }
    return DLRM_OK;
}

extern "C" int dlrm_model_load_batch(dlrm_model *m, const float *d_dense, int64_t dense_ld, const float *d_labels,
                                     dlrm_stream_t stream) {
    DLRM_REQUIRE(m != nullptr && d_dense != nullptr && d_labels != nullptr, "NULL argument");
    DLRM_REQUIRE(dense_ld >= m->cfg.dense_in, "dense_ld too small");
    // both copies in one launch (the labels as a [B, 1] block): a small step is a chain of dependent launches
    const dlrm_copy_item items[2] = {{m->x_dense, m->x_ld, d_dense, dense_ld, m->B, m->cfg.dense_in},
                                     {m->labels, 1, d_labels, 1, m->B, 1}};
    return dlrm_copy_2d_many(items, 2, stream);
}

static int ensure_sparse_ws(dlrm_model *m, const void *idx, const void *off, int64_t n_idx) {
    for (int rw = 0; rw < 2; rw++) {
        if (rw ? !(m->Trw && m->sharded) : !m->Tl) continue;
        dlrm_embbag e = table_desc(m, rw != 0, idx, off, n_idx);
        size_t need = 0;
        if (int rc = dlrm_embbag_backward_workspace_bytes(&e, &need)) return rc;
        size_t &have = rw ? m->ws_sparse_rw_bytes : m->ws_sparse_bytes;
        void *&ws = rw ? m->ws_sparse_rw : m->ws_sparse;
        if (need > have) {
            // grows with the first batch of a new size only (a steady loop never allocates)
            DLRM_HIP(hipDeviceSynchronize());
            char *p = nullptr;
            if (int rc = dalloc(m, &p, need, false)) return rc;
            ws = p; have = need;
        }
    }
    return DLRM_OK;
}

extern "C" int dlrm_model_forward(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices,
                                  dlrm_stream_t stream) {
    DLRM_REQUIRE(m != nullptr, "model is NULL");
    DLRM_REQUIRE((d_indices != nullptr || n_indices == 0) && d_offsets != nullptr && n_indices >= 0, "bad batch");
    return forward_pass(m, d_indices, d_offsets, n_indices, false, as_stream(stream));
}

static int train_step_impl(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices, hipStream_t main);

// cfg.graph_steps: the step's launches, stream forks and joins are captured once per set of batch buffers (a host
// that cycles a few resident buffers, or double-buffers its input) and replayed with one hipGraphLaunch.  At small
// batches a step is ~60 launches of 3-20 us on four streams and the time between them is what the step consists of.
static int graphed_step(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices, hipStream_t main) {
    constexpr size_t kMaxGraphs = 8;
    m->graph_clock++;
    for (auto &g : m->graphs)
        if (g.idx == d_indices && g.off == d_offsets && g.n == n_indices && g.stream == main) {
            g.used = m->graph_clock;
            DLRM_HIP(hipGraphLaunch(g.exec, main));
            return DLRM_OK;
        }
    if (m->graphs.size() >= kMaxGraphs) {      // forget the set that was used longest ago
        size_t old = 0;
        for (size_t i = 1; i < m->graphs.size(); i++)
            if (m->graphs[i].used < m->graphs[old].used) old = i;
        DLRM_HIP(hipStreamSynchronize(m->graphs[old].stream));
        (void)hipGraphExecDestroy(m->graphs[old].exec);
        (void)hipGraphDestroy(m->graphs[old].graph);
        m->graphs.erase(m->graphs.begin() + old);
    }
    dlrm_model::Captured g{d_indices, d_offsets, n_indices, main, nullptr, nullptr, m->graph_clock};
    DLRM_HIP(hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal));
    const int rc = train_step_impl(m, d_indices, d_offsets, n_indices, main);
    const hipError_t e = hipStreamEndCapture(main, &g.graph);
    if (rc) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    DLRM_HIP(e);
    DLRM_HIP(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    m->graphs.push_back(g);
    DLRM_HIP(hipGraphLaunch(g.exec, main));
    return DLRM_OK;
}

extern "C" int dlrm_model_train_step(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices,
                                     dlrm_stream_t stream) {
    DLRM_REQUIRE(m != nullptr, "model is NULL");
    DLRM_REQUIRE((d_indices != nullptr || n_indices == 0) && d_offsets != nullptr && n_indices >= 0, "bad batch");
    hipStream_t main = as_stream(stream);
    if (int rc = ensure_sparse_ws(m, d_indices, d_offsets, n_indices)) return rc;
    // (a step that brackets operators with timing events, and the exchanges of a sharded step, are launched directly)
    if (m->cfg.graph_steps && !m->sharded && m->time_mask == 0 && main != nullptr)
        return graphed_step(m, d_indices, d_offsets, n_indices, main);
    return train_step_impl(m, d_indices, d_offsets, n_indices, main);
}

static int train_step_impl(dlrm_model *m, const void *d_indices, const void *d_offsets, int64_t n_indices, hipStream_t main) {
    hipStream_t emb = m->s_emb ? m->s_emb : main;
    const dlrm_model_config &c = m->cfg;
    if (int rc = forward_pass(m, d_indices, d_offsets, n_indices, true, main)) return rc;
    // loss (means over the GLOBAL batch, so the ranks' gradients add): the gradient on the critical path, the
    // value -- one workgroup's fixed-order sum, which nothing downstream reads -- beside the backward pass
    hipStream_t side = m->s_prep ? m->s_prep : main;
    if (m->fuse_last) {
        const size_t n = m->topl.size();
        const Layer &l = m->topl[n - 1];
        const bool bce = c.loss == DLRM_LOSS_BCE;
        float *dx_max = m->f16x2 && big_layer(m, m->topl[n - 2]) ? out_slot(m, m->topl[n - 2].s_dz) : nullptr;
        if (int rc = timed(m, DLRM_TIME_LINEAR_FORWARD_SKINNY, main, [&] {
                return gemv_forward_loss_backward(m->top_acts[n - 2].p, m->top_acts[n - 2].ld, m->params + l.w_off,
                                                  m->params + l.b_off, m->labels, m->Bg, bce, m->pred, m->dpred, m->top_dz[n - 2].p,
                                                  m->top_dz[n - 2].ld, (int)m->B, l.K, c.final_relu, bce ? 0 : c.final_relu, dx_max,
                                                  main);
            }))
            return rc;
    }
    if (int rc = wait_for(m, side, main)) return rc;    // the predictions are there
    for (int half = m->fuse_last ? 1 : 0; half < 2; half++) {
        float *lo = half ? m->loss : nullptr, *gr = half ? nullptr : m->dpred;
        hipStream_t s = half ? side : main;
        if (c.loss == DLRM_LOSS_BCE) {
            if (int rc = dlrm_loss_bce_logits(m->pred, m->labels, m->B, m->Bg, lo, gr, s)) return rc;
        } else {
            if (int rc = dlrm_loss_mse(m->pred, m->labels, m->B, m->Bg, lo, gr, c.final_relu, s)) return rc;
        }
    }
    // top MLP backward, interaction backward
    if (m->dot) {
        if (int rc = stack_backward(m, m->topl, m->inter, m->inter_ld, m->top_acts, m->dpred, 1, m->top_dz, m->dinter, m->inter_ld, main, m->fuse_last))
            return rc;
        if (int rc = timed(m, DLRM_TIME_INTERACT_DOT_BACKWARD, main, [&] {
                float *dx0_max = big_layer(m, m->bot.back()) ? out_slot(m, m->bot.back().s_dz) : nullptr;
                if (m->views)   // gradients straight into the send blocks of the backward exchange
                    return interact_dot_backward_views_max(m->x_views.data(), m->dinter, m->inter_ld, m->B, m->F, m->D,
                                                           m->dx_views.data(), 1, dx0_max, main);
                if (m->gather && m->gather_bwd) {     // the tables are as the forward pass read them: their update comes later
                    dlrm_embbag e = table_desc(m, false, d_indices, d_offsets, n_indices);
                    return interact_dot_backward_gather_max(&e, m->feat, m->feat_ld, m->dinter, m->inter_ld, m->dfeat, 1, dx0_max, main);
                }
                return interact_dot_backward_max(m->feat, m->dinter, m->inter_ld, m->B, m->F, m->D, m->dfeat, 1, dx0_max, main);
            }))
            return rc;
    } else {
        if (int rc = stack_backward(m, m->topl, m->inter, m->inter_ld, m->top_acts, m->dpred, 1, m->top_dz, nullptr, 0, main, m->fuse_last)) return rc;
        // the bottom part of the concatenation is a ReLU output, the pooled part is not
        const int nb = m->bottom.back();
        const float *dz0 = m->topl.size() > 1 ? m->top_dz[0].p : m->dpred;
        const int64_t dz0_ld = m->topl.size() > 1 ? m->top_dz[0].ld : 1;
        if (int rc = linear_dgrad(m, m->topl[0], 0, nb, dz0, dz0_ld, m->dfeat, m->feat_ld, m->feat, m->feat_ld, main)) return rc;
        if (int rc = linear_dgrad(m, m->topl[0], nb, m->top_in - nb, dz0, dz0_ld, m->dfeat + nb, m->feat_ld, nullptr, 0, main)) return rc;
    }
    // with a communicator of its own the all-reduce of the top MLP's gradients (most of the dense parameters) starts
    // here, on its own stream, beside the exchange, the sparse update and the bottom MLP's backward pass
    const bool early_reduce = m->sharded && c.comm_dense != nullptr && m->s_reduce != nullptr;
    const int64_t top_off = m->topl.front().w_off;
    if (early_reduce) {
        if (int rc = wait_for(m, m->s_reduce, m->s_wgrad ? m->s_wgrad : main)) return rc;    // the top stack's weight gradients are there
        if (int rc = timed(m, DLRM_TIME_COMM_ALL_REDUCE, m->s_reduce, [&] {
                return dlrm_comm_all_reduce_sum(c.comm_dense, m->grads + top_off, m->n_params - top_off, m->s_reduce);
            }))
            return rc;
    }
    // sparse rows (exchange + fused update) on the embedding stream beside the bottom MLP's backward
    // Without a stream of its own the sparse update goes behind the bottom MLP's data gradients: ahead of them it would
    // only delay the events the weight-gradient stream waits for (small batches: that stream's chain is the step's tail).
    static const bool emb_last_ok = [] { const char *e = getenv("DLRM_EMB_LAST"); return !(e && e[0] == '0'); }();
    // (a captured step keeps the old place: replayed from a graph this order measured slower, 0.40 against 0.36 ms)
    const bool emb_last = emb == main && m->s_wgrad != nullptr && emb_last_ok && !m->cfg.graph_steps;
    if (!emb_last) {
        if (int rc = wait_for(m, emb, main)) return rc;     // dfeat is complete
        if (int rc = emb_backward(m, d_indices, d_offsets, n_indices, emb)) return rc;
    }
    if (int rc = stack_backward(m, m->bot, m->x_dense, m->x_ld, m->bot_acts, m->dfeat, m->feat_ld, m->bot_dz, nullptr, 0, main)) return rc;
    if (emb_last)
        if (int rc = emb_backward(m, d_indices, d_offsets, n_indices, main)) return rc;
    if (m->s_wgrad)
        if (int rc = wait_for(m, main, m->s_wgrad)) return rc;   // every dense gradient is there
    if (early_reduce) {
        if (int rc = wait_for(m, m->s_reduce, main)) return rc;
        if (int rc = timed(m, DLRM_TIME_COMM_ALL_REDUCE, m->s_reduce,
                           [&] { return dlrm_comm_all_reduce_sum(c.comm_dense, m->grads, top_off, m->s_reduce); }))
            return rc;
        if (int rc = wait_for(m, main, m->s_reduce)) return rc;
    } else if (m->sharded) {
        dlrm_comm *cd = c.comm_dense ? c.comm_dense : c.comm;
        if (int rc = timed(m, DLRM_TIME_COMM_ALL_REDUCE, main, [&] { return dlrm_comm_all_reduce_sum(cd, m->grads, m->n_params, main); }))
            return rc;
    }
    // (the dense update does not wait for the sparse one: in a sharded step the backward exchange and the rows' update
    //  end after the dense gradients are complete)
    if (int rc = dlrm_sgd_step(m->params, m->grads, m->n_params, c.lr, main)) return rc;
    if (int rc = wait_for(m, main, emb)) return rc;      // rows are updated before anyone reuses dfeat
    return wait_for(m, main, side);                      // the loss value is there before anyone overwrites the predictions
}

extern "C" int dlrm_model_enable_timing(dlrm_model *m, uint32_t class_mask, int32_t max_records) {
    DLRM_REQUIRE(m != nullptr && max_records >= 0, "bad argument");
    DLRM_HIP(hipDeviceSynchronize());
    m->time_mask = class_mask;
    m->n_trecs = 0;
    while ((int32_t)m->trecs.size() < max_records) {
        dlrm_model::TimeRec r{};
        DLRM_HIP(hipEventCreate(&r.e0));
        DLRM_HIP(hipEventCreate(&r.e1));
        m->trecs.push_back(r);
    }
    return DLRM_OK;
}

extern "C" int dlrm_model_collect_timing(dlrm_model *m, double *h_ms_sum, int64_t *h_launches) {
    DLRM_REQUIRE(m != nullptr && h_ms_sum != nullptr && h_launches != nullptr, "NULL argument");
    DLRM_HIP(hipDeviceSynchronize());
    for (int c = 0; c < DLRM_TIME_NUM_CLASSES; c++) { h_ms_sum[c] = 0.0; h_launches[c] = 0; }
    for (size_t i = 0; i < m->n_trecs; i++) {
        float ms = 0.f;
        DLRM_HIP(hipEventElapsedTime(&ms, m->trecs[i].e0, m->trecs[i].e1));
        h_ms_sum[m->trecs[i].cls] += ms;
        h_launches[m->trecs[i].cls] += 1;
    }
    m->n_trecs = 0;
    return DLRM_OK;
}
