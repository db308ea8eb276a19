/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's DLRM hot path, used only as the checker
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
 * product path (dlrm_amd/) never calls into this file.
 *
 * Pinning:
 *   - orc_pagecache_* (the page-LRU embedding cache state machine) is pinned
 *     against the reference's own run artefacts
 *     dlrm_libtorch/build/LRU_BASE_RMC{1,2,3}_C2_B40_D200.csv: with the
 *     reference's synthetic ids (torch::manual_seed(0), src/dlrmtest.cpp:306-317)
 *     the per-batch totals of pages read and pages written reproduce those
 *     files exactly (tests/test_oracle_pagecache_csv.py, fixtures in
 *     tests/golden/).
 *   - the floating point operators restate libtorch CPU operator semantics at
 *     the reference's call sites; the reference has no test or fixture for
 *     them, so they are pinned against fixtures generated in this container
 *     with the same ATen CPU kernels (tests/golden/make_golden.py); row-wise Adagrad, for which libtorch has
 *     no operator, against torch.optim.Adagrad on gradients where the two rules coincide
 *     (tests/test_oracle_golden.py).
 *
 * Every accumulation is done in double and rounded once, so this file is a
 * "closest to exact" comparator for both the HIP path and libtorch.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_POOL_SUM 0
#define ORC_POOL_MEAN 1
#define ORC_DUP_ACCUMULATE 0
#define ORC_DUP_LAST_WINS 1

/* ---------------------------------------------------------------- bf16 -- */

static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* round-to-nearest-even, NaN kept quiet (what a plain cast does in libtorch's
 * c10::BFloat16 and in hardware v_cvt_pk_bf16_f32) */
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

void orc_f32_to_bf16(const float *src, uint16_t *dst, int64_t n) {
    for (int64_t i = 0; i < n; i++) dst[i] = f32_to_bf16(src[i]);
}
void orc_bf16_to_f32(const uint16_t *src, float *dst, int64_t n) {
    for (int64_t i = 0; i < n; i++) dst[i] = bf16_to_f32(src[i]);
}

/* ------------------------------------------------- table initialisation -- */

/*
 * Reference: DATA::initial() fills a table with rand()/(RAND_MAX+1.0), i.e.
 * uniform [0,1) (src/data_transfer.cpp:47-61).  glibc rand() is not
 * reproducible on a GPU, so both this oracle and the HIP initialiser use the
 * same counter based generator: value(i) = top 24 bits of
 * splitmix64(seed + i) scaled by 2^-24 -- uniform [0,1) on the same grid a
 * float rand()/2^31 lands on after rounding.
 */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void orc_table_init_uniform(float *w, int64_t n, uint64_t seed, int64_t first) {
    const uint64_t base = splitmix64(seed);
    for (int64_t i = 0; i < n; i++)
        w[i] = (float)(splitmix64(base + (uint64_t)(first + i)) >> 40) * (1.0f / 16777216.0f);
}

/* -------------------------------------------------- EmbeddingBag forward -- */

/*
 * Reference call site: emb::forward, src/dlrmtest.cpp:191-197 ->
 * torch::embedding_bag(weight, indices, offsets, mode).  Bag g pools rows
 * indices[offsets[g] .. offsets[g+1]) of its table; the last bag ends at
 * n_idx when include_last_offset == 0.  Mean divides by the bag length and
 * yields 0 for an empty bag.  Bags are numbered g = t*batch + b and the
 * pooled vector is written at out[b*out_ld + t*dim] -- the concatenated
 * [B, T*D] layout the reference builds with torch::cat (:196).
 * row_dtype 0: float rows, 1: bf16 rows (accumulated in wider precision).
 */
int orc_embbag_forward(const void *weight, int row_bf16, const int64_t *table_row_offsets,
                       int num_tables, int dim, const int64_t *indices, const int64_t *offsets,
                       int64_t batch, int64_t n_idx, int include_last_offset, int pooling,
                       float *out, int64_t out_ld) {
    const float *wf = (const float *)weight;
    const uint16_t *wh = (const uint16_t *)weight;
    double *acc = (double *)malloc(sizeof(double) * (size_t)dim);
    if (!acc) return 1;
    for (int t = 0; t < num_tables; t++) {
        int64_t base = table_row_offsets ? table_row_offsets[t] : 0;
        int64_t nrows = table_row_offsets ? table_row_offsets[t + 1] - base : INT64_MAX;
        for (int64_t b = 0; b < batch; b++) {
            int64_t g = (int64_t)t * batch + b;
            int64_t s = offsets[g];
            int64_t e = (g + 1 < (int64_t)num_tables * batch || include_last_offset) ? offsets[g + 1] : n_idx;
            for (int d = 0; d < dim; d++) acc[d] = 0.0;
            for (int64_t p = s; p < e; p++) {
                int64_t r = indices[p];
                if (r < 0 || r >= nrows) { free(acc); return 2; }
                int64_t row = base + r;
                if (row_bf16)
                    for (int d = 0; d < dim; d++) acc[d] += (double)bf16_to_f32(wh[row * dim + d]);
                else
                    for (int d = 0; d < dim; d++) acc[d] += (double)wf[row * dim + d];
            }
            double sc = 1.0;
            if (pooling == ORC_POOL_MEAN) sc = (e > s) ? 1.0 / (double)(e - s) : 0.0;
            float *o = out + b * out_ld + (int64_t)t * dim;
            for (int d = 0; d < dim; d++) o[d] = (float)(acc[d] * sc);
        }
    }
    free(acc);
    return 0;
}

/* ------------------------------- sparse backward + fused row-wise update -- */

/*
 * Reference: autograd of embedding_bag (dense index_add of g/L, K10) followed
 * by torch::optim::SGD::step (src/dlrmtest.cpp:505-506,:533) and the row
 * write-back save_emb (:88-107).
 *
 * Canonical semantics (dup_mode ACCUMULATE): the gradient of a row is the sum
 * over every occurrence of that row in the batch of scale_g * grad[bag g]
 * (scale = 1 for sum, 1/len(g) for mean); each touched row is updated once.
 * As-shipped semantics (dup_mode LAST_WINS, src/dlrmtest.cpp:96-101): every
 * occurrence is updated independently from the same starting value and
 * written back in position order, so only the LAST occurrence's gradient
 * survives.
 *
 * optimizer 0: SGD              w -= lr * g
 * optimizer 1: row-wise Adagrad state[row] += mean_d(g_d^2);
 *                               w -= lr * g / (sqrt(state[row]) + eps)
 * grad is laid out like the forward output: grad[b*grad_ld + t*dim + d].
 */
typedef struct {
    int64_t key;
    int64_t pos;
    int64_t bag;
} orc_occ;

static int occ_cmp(const void *a, const void *b) {
    const orc_occ *x = (const orc_occ *)a, *y = (const orc_occ *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

int orc_embbag_backward_update(void *weight, int row_bf16, const int64_t *table_row_offsets,
                               int num_tables, int dim, const int64_t *indices,
                               const int64_t *offsets, int64_t batch, int64_t n_idx,
                               int include_last_offset, int pooling, const float *grad,
                               int64_t grad_ld, int optimizer, int dup_mode, float lr, float eps,
                               float *adagrad_state) {
    float *wf = (float *)weight;
    uint16_t *wh = (uint16_t *)weight;
    int64_t nb = (int64_t)num_tables * batch;
    orc_occ *occ = (orc_occ *)malloc(sizeof(orc_occ) * (size_t)(n_idx > 0 ? n_idx : 1));
    double *acc = (double *)malloc(sizeof(double) * (size_t)dim);
    int64_t *bag_len = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nb > 0 ? nb : 1));
    if (!occ || !acc || !bag_len) { free(occ); free(acc); free(bag_len); return 1; }
    int64_t n = 0;
    for (int64_t g = 0; g < nb; g++) {
        int t = (int)(g / batch);
        int64_t base = table_row_offsets ? table_row_offsets[t] : 0;
        int64_t s = offsets[g];
        int64_t e = (g + 1 < nb || include_last_offset) ? offsets[g + 1] : n_idx;
        bag_len[g] = e - s;
        for (int64_t p = s; p < e; p++) {
            occ[n].key = base + indices[p];
            occ[n].pos = p;
            occ[n].bag = g;
            n++;
        }
    }
    qsort(occ, (size_t)n, sizeof(orc_occ), occ_cmp);
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        while (j < n && occ[j].key == occ[i].key) j++;
        for (int d = 0; d < dim; d++) acc[d] = 0.0;
        int64_t first = (dup_mode == ORC_DUP_LAST_WINS) ? j - 1 : i;
        for (int64_t k = first; k < j; k++) {
            int64_t g = occ[k].bag;
            int t = (int)(g / batch);
            int64_t b = g % batch;
            /* the scale is applied in float, as autograd does (grad * (1/L)) */
            float sc = (pooling == ORC_POOL_MEAN) ? 1.0f / (float)bag_len[g] : 1.0f;
            const float *gp = grad + b * grad_ld + (int64_t)t * dim;
            for (int d = 0; d < dim; d++) acc[d] += (double)(gp[d] * sc);
        }
        int64_t row = occ[i].key;
        double denom = 1.0;
        if (optimizer == 1) {
            double ss = 0.0;
            for (int d = 0; d < dim; d++) { float gd = (float)acc[d]; ss += (double)gd * (double)gd; }
            float st = adagrad_state[row] + (float)(ss / (double)dim);
            adagrad_state[row] = st;
            denom = (double)(sqrtf(st) + eps);
        }
        for (int d = 0; d < dim; d++) {
            float gd = (float)acc[d];
            double step = (double)lr * (double)gd / denom;
            if (row_bf16) {
                float w = bf16_to_f32(wh[row * dim + d]);
                wh[row * dim + d] = f32_to_bf16((float)((double)w - step));
            } else {
                wf[row * dim + d] = (float)((double)wf[row * dim + d] - step);
            }
        }
        i = j;
    }
    free(occ); free(acc); free(bag_len);
    return 0;
}

/* --------------------------------------------------------------- dense -- */

/*
 * Reference: Net2::forward, src/dlrmtest.cpp:254-272 -- torch::nn::Linear
 * (y = x W^T + b, W is [N, K]) followed by torch::relu on every layer.
 */
void orc_linear_forward(const float *x, int64_t x_ld, const float *w, const float *bias, float *y,
                        int64_t y_ld, int64_t M, int N, int K, int relu) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++) {
        for (int n = 0; n < N; n++) {
            double a = bias ? (double)bias[n] : 0.0;
            const float *xr = x + m * x_ld, *wr = w + (int64_t)n * K;
            for (int k = 0; k < K; k++) a += (double)xr[k] * (double)wr[k];
            float v = (float)a;
            y[m * y_ld + n] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
}

/*
 * Backward of y = act(x W^T + b).  dy is the gradient w.r.t. the activation
 * output; when relu != 0 it is first masked with (y > 0) (ReLU backward as
 * autograd's threshold_backward does).  Outputs: dx [M,K] (optional), dw
 * [N,K], db [N] (optional).
 */
void orc_linear_backward(const float *x, int64_t x_ld, const float *w, const float *y, int64_t y_ld,
                         const float *dy, int64_t dy_ld, float *dx, int64_t dx_ld, float *dw,
                         float *db, int64_t M, int N, int K, int relu) {
    float *dz = (float *)malloc(sizeof(float) * (size_t)M * (size_t)N);
    for (int64_t m = 0; m < M; m++)
        for (int n = 0; n < N; n++) {
            float g = dy[m * dy_ld + n];
            if (relu && !(y[m * y_ld + n] > 0.0f)) g = 0.0f;
            dz[m * N + n] = g;
        }
    if (dx) {
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++)
            for (int k = 0; k < K; k++) {
                double a = 0.0;
                for (int n = 0; n < N; n++) a += (double)dz[m * N + n] * (double)w[(int64_t)n * K + k];
                dx[m * dx_ld + k] = (float)a;
            }
    }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        for (int k = 0; k < K; k++) {
            double a = 0.0;
            for (int64_t m = 0; m < M; m++) a += (double)dz[m * N + n] * (double)x[m * x_ld + k];
            dw[(int64_t)n * K + k] = (float)a;
        }
        if (db) {
            double a = 0.0;
            for (int64_t m = 0; m < M; m++) a += (double)dz[m * N + n];
            db[n] = (float)a;
        }
    }
    free(dz);
}

/* plain SGD on a dense tensor: torch::optim::SGD without momentum state
 * (the reference re-creates the optimizer every batch, so its momentum
 * buffer is always fresh; src/dlrmtest.cpp:499-506) */
void orc_sgd_step(float *w, const float *g, int64_t n, float lr) {
    for (int64_t i = 0; i < n; i++) w[i] = (float)((double)w[i] - (double)lr * (double)g[i]);
}

/* ---------------------------------------------------------- interaction -- */

/*
 * Feature interaction.
 * Concatenation is what the reference does (torch::cat, src/dlrmtest.cpp:265).
 * Pairwise dot interaction is absent from the reference; this restates the
 * standard DLRM formulation (bmm + strict lower triangle, row-major order
 * (i,j), j<i) followed by concatenation after the dense vector:
 *   x: [B, F, D] (feature 0 = bottom MLP output), out: [B, D + F(F-1)/2].
 */
void orc_interact_dot_forward(const float *x, int64_t B, int F, int D, float *out, int64_t out_ld) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        const float *xb = x + b * (int64_t)F * D;
        float *o = out + b * out_ld;
        for (int d = 0; d < D; d++) o[d] = xb[d];
        int c = D;
        for (int i = 1; i < F; i++)
            for (int j = 0; j < i; j++) {
                double a = 0.0;
                for (int d = 0; d < D; d++) a += (double)xb[i * D + d] * (double)xb[j * D + d];
                o[c++] = (float)a;
            }
    }
}

void orc_interact_dot_backward(const float *x, const float *dout, int64_t dout_ld, int64_t B, int F,
                               int D, float *dx) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        const float *xb = x + b * (int64_t)F * D;
        const float *go = dout + b * dout_ld;
        float *gx = dx + b * (int64_t)F * D;
        double *acc = (double *)calloc((size_t)F * (size_t)D, sizeof(double));
        for (int d = 0; d < D; d++) acc[d] = (double)go[d];
        int c = D;
        for (int i = 1; i < F; i++)
            for (int j = 0; j < i; j++) {
                double g = (double)go[c++];
                for (int d = 0; d < D; d++) {
                    acc[i * D + d] += g * (double)xb[j * D + d];
                    acc[j * D + d] += g * (double)xb[i * D + d];
                }
            }
        for (int k = 0; k < F * D; k++) gx[k] = (float)acc[k];
        free(acc);
    }
}

/* ----------------------------------------------------------------- loss -- */

/* torch::mse_loss(pred, target), mean reduction (src/dlrmtest.cpp:522);
 * dpred = 2 (pred - target) / n */
float orc_mse_loss(const float *pred, const float *target, int64_t n, float *dpred) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double d = (double)pred[i] - (double)target[i];
        s += d * d;
        if (dpred) dpred[i] = (float)(2.0 * d / (double)n);
    }
    return (float)(s / (double)n);
}

/* binary_cross_entropy_with_logits, mean reduction (the MLPerf-DLRM loss; not
 * in the reference): l = max(x,0) - x t + log1p(exp(-|x|)),
 * dx = (sigmoid(x) - t) / n */
float orc_bce_logits_loss(const float *logit, const float *target, int64_t n, float *dlogit) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double x = (double)logit[i], t = (double)target[i];
        s += (x > 0 ? x : 0) - x * t + log1p(exp(-fabs(x)));
        if (dlogit) dlogit[i] = (float)((1.0 / (1.0 + exp(-x)) - t) / (double)n);
    }
    return (float)(s / (double)n);
}

/* -------------------------------------------- page-LRU embedding cache -- */

/*
 * Restatement of class cache (include/data.h:242-298, src/emb_cache.cpp) and
 * of the policy driver reclaim_emb_cache (src/dlrmtest.cpp:397-412), with the
 * OCSSD replaced by "the backing store": only the state that decides WHICH
 * pages are resident, in what LRU order, and the counters are kept.
 *
 * A page holds vpo = page_bytes / (dim*4) consecutive rows of one table
 * (include/data.h:211).  Per table an LRU list of resident pages, MRU first
 * (emb_cache.cpp:317-352).  Physical placement (which PU a page lives on) is
 * out of scope, so the read/write page counters are kept per TABLE; the
 * reference indexes them by physical PU (emb_cache.cpp:435, :315), which
 * changes only how the same total is split over CSV rows.
 */
typedef struct {
    int64_t prev, next; /* list links, -1 = none; prev == -2 means "not resident" */
    int64_t slot;       /* cache slot holding the page, -1 when not resident */
} orc_node;

typedef struct orc_pagecache {
    int num_tables;
    int64_t rows_per_table;
    int dim;
    int vpo;           /* rows per page */
    int ws_opt;        /* pages evicted per Reclaim_vector_page call */
    int64_t pages_per_table;
    int64_t cache_size; /* capacity threshold in pages (cache::cache_size) */
    int64_t page_count; /* resident pages, all tables (cache::page_count) */
    orc_node *nodes;    /* [num_tables * pages_per_table] */
    int64_t *head;      /* MRU page per table */
    int64_t *tail;      /* LRU page per table */
    int64_t *lru_len;
    int64_t *read_pages;  /* write_buffer::read_page_count, per table */
    int64_t *write_pages; /* write_buffer::write_page_count, per table */
    int64_t *hit_ids;     /* lookups that found their page resident */
    int64_t *miss_ids;    /* lookups that did not */
    /* Slot pool (not in the reference, which keeps each page in its own heap
     * block): the HBM cache stores resident pages in n_slots fixed slots.
     * Free slots sit in a ring; a page read takes the slot at the ring head,
     * an evicted page returns its slot at the ring tail.  Reads are served in
     * read order, evictions in eviction order, so the page->slot map is a
     * deterministic function of the call sequence. */
    int64_t n_slots;
    int64_t *ring;
    int64_t ring_head, ring_tail; /* monotonically increasing; index mod n_slots */
    /* Write-back economy (not in the reference, which writes every victim back, emb_cache.cpp:259-305):
     * a page is dirty from the first update of one of its rows (orc_pagecache_mark_dirty) until it leaves;
     * written_back counts the victims that have to be copied to the store: all of them, or with dirty_only
     * the dirty ones. */
    unsigned char *dirty; /* [num_tables * pages_per_table] */
    int64_t *written_back;
    int dirty_only;
} orc_pagecache;

orc_pagecache *orc_pagecache_create(int num_tables, int64_t rows_per_table, int dim, int page_bytes,
                                    int ws_opt, int64_t cache_size) {
    orc_pagecache *c = (orc_pagecache *)calloc(1, sizeof(*c));
    c->num_tables = num_tables;
    c->rows_per_table = rows_per_table;
    c->dim = dim;
    c->vpo = page_bytes / (dim * 4);
    c->ws_opt = ws_opt;
    c->pages_per_table = (rows_per_table + c->vpo - 1) / c->vpo;
    c->cache_size = cache_size;
    int64_t n = (int64_t)num_tables * c->pages_per_table;
    c->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)n);
    for (int64_t i = 0; i < n; i++) { c->nodes[i].prev = -2; c->nodes[i].next = -1; c->nodes[i].slot = -1; }
    c->head = (int64_t *)malloc(sizeof(int64_t) * num_tables);
    c->tail = (int64_t *)malloc(sizeof(int64_t) * num_tables);
    for (int t = 0; t < num_tables; t++) c->head[t] = c->tail[t] = -1;
    c->lru_len = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->read_pages = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->write_pages = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->hit_ids = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->miss_ids = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->written_back = (int64_t *)calloc(num_tables, sizeof(int64_t));
    c->dirty = (unsigned char *)calloc((size_t)n, 1);
    return c;
}

void orc_pagecache_destroy(orc_pagecache *c) {
    if (!c) return;
    free(c->nodes); free(c->head); free(c->tail); free(c->lru_len);
    free(c->read_pages); free(c->write_pages); free(c->hit_ids); free(c->miss_ids);
    free(c->ring); free(c->dirty); free(c->written_back);
    free(c);
}

static void pc_unlink(orc_pagecache *c, int t, int64_t page) {
    orc_node *nd = c->nodes + (int64_t)t * c->pages_per_table;
    int64_t p = nd[page].prev, n = nd[page].next;
    if (p >= 0) nd[p].next = n; else c->head[t] = n;
    if (n >= 0) nd[n].prev = p; else c->tail[t] = p;
}

static void pc_push_front(orc_pagecache *c, int t, int64_t page) {
    orc_node *nd = c->nodes + (int64_t)t * c->pages_per_table;
    nd[page].prev = -1;
    nd[page].next = c->head[t];
    if (c->head[t] >= 0) nd[c->head[t]].prev = page; else c->tail[t] = page;
    c->head[t] = page;
}

/*
 * cache::get_emb_vectors(ids, t) = read() + lru_page()
 * (src/emb_cache.cpp:358-371).  ids are visited in the order given; the
 * reference driver sorts them first (src/dlrmtest.cpp:62-63).
 *  read(): ids whose page is not resident form the miss list (:410-414);
 *  consecutive entries on the same page are one device read (:426-438);
 *  read_page_count and page_count grow by the number of reads (:435,:441);
 *  the pages become resident (:502-512).  With unsorted ids a page that
 *  occurs twice non-consecutively in the miss list is read twice, as in the
 *  reference.
 *  lru_page(): every id, in order, moves its page to the MRU end (:317-352).
 * Returns the number of page reads; miss_pages_out (optional, capacity n)
 * receives the pages read, in read order.
 */
int64_t orc_pagecache_get(orc_pagecache *c, int t, const int64_t *ids, int64_t n,
                          int64_t *miss_pages_out) {
    orc_node *nd = c->nodes + (int64_t)t * c->pages_per_table;
    int64_t reads = 0, last = -1;
    int64_t *newly = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        int64_t page = ids[i] / c->vpo;
        if (nd[page].prev != -2) { c->hit_ids[t]++; continue; }
        c->miss_ids[t]++;
        if (reads > 0 && page == last) continue;
        if (miss_pages_out) miss_pages_out[reads] = page;
        newly[reads++] = page;
        last = page;
    }
    if (c->n_slots > 0) {
        if (c->ring_head + reads > c->ring_tail) { free(newly); return -1; } /* pool exhausted */
        for (int64_t i = 0; i < reads; i++)
            if (nd[newly[i]].slot < 0) nd[newly[i]].slot = c->ring[(c->ring_head++) % c->n_slots];
    }
    c->read_pages[t] += reads;
    c->page_count += reads;
    /* residency is set after the miss scan, as in the reference */
    for (int64_t i = 0; i < n; i++) {
        int64_t page = ids[i] / c->vpo;
        if (nd[page].prev != -2) pc_unlink(c, t, page); else c->lru_len[t]++;
        pc_push_front(c, t, page);
    }
    free(newly);
    return reads;
}

/*
 * cache::Reclaim_vector_page(t) (src/emb_cache.cpp:208-316): drop ws_opt
 * pages from the LRU end of table t and write them back;
 * page_count -= ws_opt and write_page_count += ws_opt unconditionally
 * (:312,:315).  The reference pops from an empty list when fewer than ws_opt
 * pages are resident (undefined behaviour); here only existing pages are
 * dropped, the counters still move by ws_opt.  evicted_out (optional,
 * capacity ws_opt) receives the dropped pages, LRU first; returns how many.
 */
int orc_pagecache_reclaim_table(orc_pagecache *c, int t, int64_t *evicted_out) {
    orc_node *nd = c->nodes + (int64_t)t * c->pages_per_table;
    int k = 0;
    for (; k < c->ws_opt && c->tail[t] >= 0; k++) {
        int64_t page = c->tail[t];
        pc_unlink(c, t, page);
        nd[page].prev = -2;
        nd[page].next = -1;
        {
            unsigned char *dy = c->dirty + (int64_t)t * c->pages_per_table + page;
            if (!c->dirty_only || *dy) c->written_back[t]++;
            *dy = 0;
        }
        if (c->n_slots > 0) {
            c->ring[(c->ring_tail++) % c->n_slots] = nd[page].slot;
            nd[page].slot = -1;
        }
        c->lru_len[t]--;
        if (evicted_out) evicted_out[k] = page;
    }
    c->page_count -= c->ws_opt;
    c->write_pages[t] += c->ws_opt;
    return k;
}

/* reclaim_emb_cache (src/dlrmtest.cpp:397-403): while over capacity, one
 * Reclaim_vector_page for every table.  Returns the number of passes. */
int64_t orc_pagecache_reclaim(orc_pagecache *c) {
    int64_t passes = 0;
    while (c->page_count > c->cache_size) {
        for (int t = 0; t < c->num_tables; t++) orc_pagecache_reclaim_table(c, t, NULL);
        passes++;
    }
    return passes;
}

/* give the cache a pool of n_slots page slots, all free, ring = 0,1,2,... */
void orc_pagecache_enable_slots(orc_pagecache *c, int64_t n_slots) {
    c->n_slots = n_slots;
    c->ring = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_slots);
    for (int64_t i = 0; i < n_slots; i++) c->ring[i] = i;
    c->ring_head = 0;
    c->ring_tail = n_slots;
}
int64_t orc_pagecache_slot_of(const orc_pagecache *c, int t, int64_t page) {
    return c->nodes[(int64_t)t * c->pages_per_table + page].slot;
}
/* slot map of table t for pages [0, n): -1 = not resident */
void orc_pagecache_slot_map(const orc_pagecache *c, int t, int64_t *out, int64_t n) {
    for (int64_t p = 0; p < n; p++) out[p] = c->nodes[(int64_t)t * c->pages_per_table + p].slot;
}
int64_t orc_pagecache_ring_head(const orc_pagecache *c) { return c->ring_head; }
int64_t orc_pagecache_ring_tail(const orc_pagecache *c) { return c->ring_tail; }

int64_t orc_pagecache_page_count(const orc_pagecache *c) { return c->page_count; }
void orc_pagecache_set_cache_size(orc_pagecache *c, int64_t s) { c->cache_size = s; }
int64_t orc_pagecache_cache_size(const orc_pagecache *c) { return c->cache_size; }
int orc_pagecache_vpo(const orc_pagecache *c) { return c->vpo; }
int orc_pagecache_is_resident(const orc_pagecache *c, int t, int64_t page) {
    return c->nodes[(int64_t)t * c->pages_per_table + page].prev != -2;
}
int64_t orc_pagecache_lru_len(const orc_pagecache *c, int t) { return c->lru_len[t]; }

/* LRU order of table t, MRU first; returns the number written (<= cap) */
int64_t orc_pagecache_lru_order(const orc_pagecache *c, int t, int64_t *out, int64_t cap) {
    const orc_node *nd = c->nodes + (int64_t)t * c->pages_per_table;
    int64_t k = 0;
    for (int64_t p = c->head[t]; p >= 0 && k < cap; p = nd[p].next) out[k++] = p;
    return k;
}

void orc_pagecache_set_dirty_only(orc_pagecache *c, int on) { c->dirty_only = on; }
/* rows `ids` of table t are updated: their (resident) pages differ from the store from now on */
void orc_pagecache_mark_dirty(orc_pagecache *c, int t, const int64_t *ids, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        int64_t page = ids[i] / c->vpo;
        if (c->nodes[(int64_t)t * c->pages_per_table + page].prev != -2)
            c->dirty[(int64_t)t * c->pages_per_table + page] = 1;
    }
}

/* counters: which 0 read_pages, 1 write_pages, 2 hit_ids, 3 miss_ids, 4 written_back;
 * reset != 0 zeroes them after the copy (cache::read_page / write_page,
 * include/data.h:274-285) */
void orc_pagecache_counters(orc_pagecache *c, int which, int64_t *out, int reset) {
    int64_t *src = which == 0 ? c->read_pages : which == 1 ? c->write_pages
                 : which == 2 ? c->hit_ids : which == 3 ? c->miss_ids : c->written_back;
    for (int t = 0; t < c->num_tables; t++) { out[t] = src[t]; if (reset) src[t] = 0; }
}
