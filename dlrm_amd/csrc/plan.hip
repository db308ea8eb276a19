// Recorded step: the host describes its launches once (descriptors are copied), then replays any
// range of them with one call.  The reference re-builds modules and an autograd graph per batch
// (src/dlrmtest.cpp:473-545); here the sequence of a train step is fixed, so the per-batch host work
// is one call that patches the batch's index / offset pointers and one call per replayed range.
#include <functional>
#include <vector>

#include "common.h"

struct dlrm_plan {
    struct Entry {
        int op;
        std::function<int(dlrm_plan *, hipStream_t)> run;
    };
    std::vector<Entry> entries;
    // the batch: EmbeddingBag entries address it through element offsets recorded with them
    const void *d_indices = nullptr;
    const void *d_offsets = nullptr;
    int64_t n_indices = 0;
    // optional timing: event pairs per entry and replay, summed on request
    int timing_runs = 0;                 // capacity (replays of an entry between two collects)
    std::vector<hipEvent_t> events;      // [entry][run][2]
    std::vector<int> used;               // replays recorded per entry since the last collect
};

// the dlrm_plan_add_* functions return an entry number, so their failures are -1, not a status
#define PLAN_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            ::dlrm::set_error(DLRM_ERR_INVALID, __VA_ARGS__);    \
            return -1;                                           \
        }                                                        \
    } while (0)

namespace {

int idx_size(int32_t idx_dtype) { return idx_dtype == DLRM_I64 ? 8 : 4; }

int add(dlrm_plan *p, int op, std::function<int(dlrm_plan *, hipStream_t)> fn) {
    p->entries.push_back({op, std::move(fn)});
    return (int)p->entries.size() - 1;
}

void drop_events(dlrm_plan *p) {
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    p->used.clear();
    p->timing_runs = 0;
}

}  // namespace

using namespace dlrm;

extern "C" int dlrm_plan_create(dlrm_plan **out) {
    DLRM_REQUIRE(out != nullptr, "out is NULL");
    *out = new dlrm_plan();
    return DLRM_OK;
}

extern "C" int dlrm_plan_destroy(dlrm_plan *p) {
    if (p) {
        drop_events(p);
        delete p;
    }
    return DLRM_OK;
}

extern "C" int dlrm_plan_size(const dlrm_plan *p) { return p ? (int)p->entries.size() : 0; }

extern "C" int dlrm_plan_op(const dlrm_plan *p, int entry) {
    return (p && entry >= 0 && entry < (int)p->entries.size()) ? p->entries[entry].op : -1;
}

extern "C" int dlrm_plan_set_batch(dlrm_plan *p, const void *d_indices, const void *d_offsets, int64_t n_indices) {
    DLRM_REQUIRE(p != nullptr, "plan is NULL");
    p->d_indices = d_indices; p->d_offsets = d_offsets; p->n_indices = n_indices;
    return DLRM_OK;
}

extern "C" int dlrm_plan_add_embbag_forward(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first, float *d_out,
                                            int64_t out_ld) {
    PLAN_REQUIRE(p != nullptr && e != nullptr && offsets_first >= 0, "bad arguments");
    const dlrm_embbag d = *e;
    return add(p, DLRM_OP_EMBBAG_FORWARD, [d, offsets_first, d_out, out_ld](dlrm_plan *pl, hipStream_t s) {
        dlrm_embbag x = d;
        x.d_indices = pl->d_indices;
        x.d_offsets = static_cast<const char *>(pl->d_offsets) + offsets_first * idx_size(d.idx_dtype);
        x.n_indices = pl->n_indices;
        return dlrm_embbag_forward(&x, d_out, out_ld, s);
    });
}

extern "C" int dlrm_plan_add_embbag_backward_update(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first,
                                                    const float *d_grad, int64_t grad_ld, const dlrm_sparse_opt *opt,
                                                    void *d_workspace, size_t workspace_bytes) {
    PLAN_REQUIRE(p != nullptr && e != nullptr && opt != nullptr && offsets_first >= 0, "bad arguments");
    const dlrm_embbag d = *e;
    const dlrm_sparse_opt o = *opt;
    return add(p, DLRM_OP_EMBBAG_BACKWARD_UPDATE,
               [d, o, offsets_first, d_grad, grad_ld, d_workspace, workspace_bytes](dlrm_plan *pl, hipStream_t s) {
                   dlrm_embbag x = d;
                   x.d_indices = pl->d_indices;
                   x.d_offsets = static_cast<const char *>(pl->d_offsets) + offsets_first * idx_size(d.idx_dtype);
                   x.n_indices = pl->n_indices;
                   return dlrm_embbag_backward_update(&x, d_grad, grad_ld, &o, d_workspace, workspace_bytes, s);
               });
}

extern "C" int dlrm_plan_add_embbag_backward_prepare(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first,
                                                     void *d_workspace, size_t workspace_bytes) {
    PLAN_REQUIRE(p != nullptr && e != nullptr && offsets_first >= 0, "bad arguments");
    const dlrm_embbag d = *e;
    return add(p, DLRM_OP_EMBBAG_BACKWARD_PREPARE,
               [d, offsets_first, d_workspace, workspace_bytes](dlrm_plan *pl, hipStream_t s) {
                   dlrm_embbag x = d;
                   x.d_indices = pl->d_indices;
                   x.d_offsets = static_cast<const char *>(pl->d_offsets) + offsets_first * idx_size(d.idx_dtype);
                   x.n_indices = pl->n_indices;
                   return dlrm_embbag_backward_prepare(&x, d_workspace, workspace_bytes, s);
               });
}

extern "C" int dlrm_plan_add_embbag_backward_apply(dlrm_plan *p, const dlrm_embbag *e, int64_t offsets_first,
                                                   const float *d_grad, int64_t grad_ld, const dlrm_sparse_opt *opt,
                                                   void *d_workspace, size_t workspace_bytes) {
    PLAN_REQUIRE(p != nullptr && e != nullptr && opt != nullptr && offsets_first >= 0, "bad arguments");
    const dlrm_embbag d = *e;
    const dlrm_sparse_opt o = *opt;
    return add(p, DLRM_OP_EMBBAG_BACKWARD_APPLY,
               [d, o, offsets_first, d_grad, grad_ld, d_workspace, workspace_bytes](dlrm_plan *pl, hipStream_t s) {
                   dlrm_embbag x = d;
                   x.d_indices = pl->d_indices;
                   x.d_offsets = static_cast<const char *>(pl->d_offsets) + offsets_first * idx_size(d.idx_dtype);
                   x.n_indices = pl->n_indices;
                   return dlrm_embbag_backward_apply(&x, d_grad, grad_ld, &o, d_workspace, workspace_bytes, s);
               });
}

extern "C" int dlrm_plan_add_linear_forward(dlrm_plan *p, const dlrm_linear *l) {
    PLAN_REQUIRE(p != nullptr && l != nullptr, "bad arguments");
    const dlrm_linear d = *l;
    return add(p, DLRM_OP_LINEAR_FORWARD, [d](dlrm_plan *, hipStream_t s) { return dlrm_linear_forward(&d, s); });
}

extern "C" int dlrm_plan_add_linear_backward_data(dlrm_plan *p, const dlrm_linear_bwd *l) {
    PLAN_REQUIRE(p != nullptr && l != nullptr, "bad arguments");
    const dlrm_linear_bwd d = *l;
    return add(p, DLRM_OP_LINEAR_BACKWARD_DATA,
               [d](dlrm_plan *, hipStream_t s) { return dlrm_linear_backward_data(&d, s); });
}

extern "C" int dlrm_plan_add_linear_backward_weight(dlrm_plan *p, const dlrm_linear_bwd *l, void *d_workspace,
                                                    size_t workspace_bytes) {
    PLAN_REQUIRE(p != nullptr && l != nullptr, "bad arguments");
    const dlrm_linear_bwd d = *l;
    return add(p, DLRM_OP_LINEAR_BACKWARD_WEIGHT, [d, d_workspace, workspace_bytes](dlrm_plan *, hipStream_t s) {
        return dlrm_linear_backward_weight(&d, d_workspace, workspace_bytes, s);
    });
}

extern "C" int dlrm_plan_add_sgd_step(dlrm_plan *p, float *d_w, const float *d_g, int64_t n, float lr) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_SGD_STEP, [=](dlrm_plan *, hipStream_t s) { return dlrm_sgd_step(d_w, d_g, n, lr, s); });
}

extern "C" int dlrm_plan_add_interact_dot_forward(dlrm_plan *p, const float *d_x, int64_t B, int32_t F, int32_t D,
                                                  float *d_out, int64_t out_ld) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_INTERACT_DOT_FORWARD,
               [=](dlrm_plan *, hipStream_t s) { return dlrm_interact_dot_forward(d_x, B, F, D, d_out, out_ld, s); });
}

extern "C" int dlrm_plan_add_interact_dot_backward(dlrm_plan *p, const float *d_x, const float *d_dout, int64_t dout_ld,
                                                   int64_t B, int32_t F, int32_t D, float *d_dx,
                                                   int32_t relu_mask_feature0) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_INTERACT_DOT_BACKWARD, [=](dlrm_plan *, hipStream_t s) {
        return dlrm_interact_dot_backward(d_x, d_dout, dout_ld, B, F, D, d_dx, relu_mask_feature0, s);
    });
}

extern "C" int dlrm_plan_add_loss_mse(dlrm_plan *p, const float *d_pred, const float *d_target, int64_t n,
                                      int64_t mean_over, float *d_loss, float *d_dpred, int32_t relu_mask_pred) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_LOSS, [=](dlrm_plan *, hipStream_t s) {
        return dlrm_loss_mse(d_pred, d_target, n, mean_over, d_loss, d_dpred, relu_mask_pred, s);
    });
}

extern "C" int dlrm_plan_add_loss_bce_logits(dlrm_plan *p, const float *d_logit, const float *d_target, int64_t n,
                                             int64_t mean_over, float *d_loss, float *d_dlogit) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_LOSS, [=](dlrm_plan *, hipStream_t s) {
        return dlrm_loss_bce_logits(d_logit, d_target, n, mean_over, d_loss, d_dlogit, s);
    });
}

extern "C" int dlrm_plan_add_copy_2d(dlrm_plan *p, float *d_dst, int64_t dst_ld, const float *d_src, int64_t src_ld,
                                     int64_t rows, int64_t cols) {
    PLAN_REQUIRE(p != nullptr, "plan is NULL");
    return add(p, DLRM_OP_COPY_2D,
               [=](dlrm_plan *, hipStream_t s) { return dlrm_copy_2d(d_dst, dst_ld, d_src, src_ld, rows, cols, s); });
}

extern "C" int dlrm_plan_enable_timing(dlrm_plan *p, int32_t max_runs) {
    DLRM_REQUIRE(p != nullptr && max_runs >= 0, "bad arguments");
    drop_events(p);
    if (max_runs == 0) return DLRM_OK;
    const size_t n = p->entries.size() * (size_t)max_runs * 2;
    p->events.resize(n, nullptr);
    for (size_t i = 0; i < n; i++) DLRM_HIP(hipEventCreate(&p->events[i]));
    p->used.assign(p->entries.size(), 0);
    p->timing_runs = max_runs;
    return DLRM_OK;
}

extern "C" int dlrm_plan_run(dlrm_plan *p, int32_t first, int32_t last, dlrm_stream_t stream) {
    DLRM_REQUIRE(p != nullptr, "plan is NULL");
    DLRM_REQUIRE(first >= 0 && last <= (int)p->entries.size() && first <= last, "bad range [%d, %d) of %d entries",
                 first, last, (int)p->entries.size());
    hipStream_t s = as_stream(stream);
    for (int i = first; i < last; i++) {
        hipEvent_t *ev = nullptr;
        if (p->timing_runs > 0 && p->used[i] < p->timing_runs)
            ev = &p->events[((size_t)i * p->timing_runs + p->used[i]) * 2];
        if (ev) DLRM_HIP(hipEventRecord(ev[0], s));
        if (int rc = p->entries[i].run(p, s)) return rc;
        if (ev) {
            DLRM_HIP(hipEventRecord(ev[1], s));
            p->used[i]++;
        }
    }
    return DLRM_OK;
}

extern "C" int dlrm_plan_collect_timing(dlrm_plan *p, double *h_ms_sum, int32_t *h_runs) {
    DLRM_REQUIRE(p != nullptr && h_ms_sum != nullptr && h_runs != nullptr, "bad arguments");
    for (size_t i = 0; i < p->entries.size(); i++) {
        h_ms_sum[i] = 0.0;
        h_runs[i] = 0;
        if (p->timing_runs == 0) continue;
        for (int r = 0; r < p->used[i]; r++) {
            hipEvent_t *ev = &p->events[(i * p->timing_runs + r) * 2];
            DLRM_HIP(hipEventSynchronize(ev[1]));
            float ms = 0.f;
            DLRM_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
            h_ms_sum[i] += ms;
        }
        h_runs[i] = p->used[i];
        p->used[i] = 0;
    }
    return DLRM_OK;
}
