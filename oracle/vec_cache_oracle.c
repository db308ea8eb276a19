/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's second cache policy, class cache2
 * (dlrm_libtorch/include/data2.h:219-287, src/emb_cache2.cpp), as its driver
 * src/dlrmtest2.cpp executes it: residency per embedding vector, a per-table
 * list that is FIFO in effect (lru_page counts hits but never moves an entry,
 * emb_cache2.cpp:638-659), eviction of the oldest entry into a per-table
 * buffer that is written out as ws_opt pages once it holds ws_opt * vpo vectors
 * (:397-633) -- vectors evicted together land in one new page and come back
 * together with one page read (the reverse table, :866-875).  Flash placement
 * below the page identity (chunks, write pointers, sector states, GC) is not
 * modelled: only which vectors share a page matters for the counters.
 *
 * Pinning: with the reference's synthetic ids (torch::manual_seed(0),
 * src/dlrmtest2.cpp:371-381) the per-batch totals of pages read and written
 * reproduce dlrm_libtorch/build/LRU_RMC{1,2,3}_C2_B40_D200.csv exactly
 * (tests/test_oracle_veccache_csv.py, fixture in tests/golden/).
 *
 * Used only by tests/; the product path (dlrm_amd/) never calls into this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t rows, n_init_pages;
    uint8_t *resident;   /* data[v % vpo] != NULL of the vector's home page */
    uint8_t *listed;     /* vector is in lru_main[t].hash_table */
    int32_t *cnt;        /* the list entry's .second: hits since insertion */
    int32_t *loc;        /* transfer_PA: page that holds the vector (init pages 0..n_init-1, then written pages) */
    /* lru_main[t].lru: front = newest.  Stored as a growing array, back = fifo[fifo_back] */
    int32_t *fifo;
    int64_t fifo_back, fifo_front, fifo_cap;   /* entries [fifo_back, fifo_front) */
    /* write_buf[t].reclaim */
    int32_t *rc_row, *rc_cnt;
    int64_t rc_n;
    /* write_buf[t].reverse_table for written pages: page (n_init + k) holds log_rows[k*vpo .. +vpo) */
    int32_t *log_rows;
    int64_t n_log_pages, log_cap;
    int64_t read_pages, write_pages, hit_ids, miss_ids;
    /* scratch for read(): stamp per page, grown with the log */
    int64_t *stamp;
    int64_t stamp_cap;
} vc_table;

typedef struct orc_veccache {
    int T, dim, vpo, ws;
    int64_t cache_size;   /* vectors per table (src/dlrmtest2.cpp:53, :468) */
    int64_t page_count;   /* only ever decremented by the reference (emb_cache2.cpp:614) */
    int64_t call;         /* stamp of the current read() */
    vc_table *t;
} orc_veccache;

orc_veccache *orc_veccache_create(int num_tables, int64_t rows_per_table, int dim, int page_bytes, int ws_opt,
                                  int64_t cache_size) {
    orc_veccache *c = calloc(1, sizeof(*c));
    c->T = num_tables; c->dim = dim; c->ws = ws_opt; c->cache_size = cache_size;
    c->vpo = page_bytes / (dim * (int)sizeof(float));   /* include/data2.h:161 */
    c->t = calloc((size_t)num_tables, sizeof(vc_table));
    for (int i = 0; i < num_tables; i++) {
        vc_table *t = &c->t[i];
        t->rows = rows_per_table;
        t->n_init_pages = (rows_per_table + c->vpo - 1) / c->vpo;
        t->resident = calloc((size_t)rows_per_table, 1);
        t->listed = calloc((size_t)rows_per_table, 1);
        t->cnt = calloc((size_t)rows_per_table, sizeof(int32_t));
        t->loc = malloc((size_t)rows_per_table * sizeof(int32_t));
        /* transfer_PA_from_init (src/emb_init2.cpp:792-806): chunk and sector of the home page */
        for (int64_t v = 0; v < rows_per_table; v++) t->loc[v] = (int32_t)(v / c->vpo);
        t->fifo_cap = 1024;
        t->fifo = malloc((size_t)t->fifo_cap * sizeof(int32_t));
        t->rc_row = malloc((size_t)ws_opt * c->vpo * sizeof(int32_t));
        t->rc_cnt = malloc((size_t)ws_opt * c->vpo * sizeof(int32_t));
        t->stamp_cap = t->n_init_pages;
        t->stamp = calloc((size_t)t->stamp_cap, sizeof(int64_t));
    }
    return c;
}

void orc_veccache_destroy(orc_veccache *c) {
    for (int i = 0; i < c->T; i++) {
        vc_table *t = &c->t[i];
        free(t->resident); free(t->listed); free(t->cnt); free(t->loc); free(t->fifo);
        free(t->rc_row); free(t->rc_cnt); free(t->log_rows); free(t->stamp);
    }
    free(c->t);
    free(c);
}

static void fifo_push_front(vc_table *t, int32_t v) {
    if (t->fifo_front == t->fifo_cap) {
        const int64_t n = t->fifo_front - t->fifo_back;
        if (t->fifo_back >= n && t->fifo_back > 0) {          /* reuse the space the back has left */
            memmove(t->fifo, t->fifo + t->fifo_back, (size_t)n * sizeof(int32_t));
        } else {
            int32_t *nf = malloc((size_t)t->fifo_cap * 2 * sizeof(int32_t));
            memcpy(nf, t->fifo + t->fifo_back, (size_t)n * sizeof(int32_t));
            free(t->fifo);
            t->fifo = nf;
            t->fifo_cap *= 2;
        }
        t->fifo_back = 0;
        t->fifo_front = n;
    }
    t->fifo[t->fifo_front++] = v;
}

/*
 * cache2::get_emb_vectors (emb_cache2.cpp:686-699) = read() (:729-892) + lru_page() (:638-659).
 * ids are taken in the caller's order (create_emb of dlrmtest2 does not sort them,
 * src/dlrmtest2.cpp:63-67, :128).  pages_out (optional, n entries) receives the pages read, in read
 * order.  Returns their number.
 */
int64_t orc_veccache_get(orc_veccache *c, int ti, const int64_t *ids, int64_t n, int64_t *pages_out) {
    vc_table *t = &c->t[ti];
    const int vpo = c->vpo;
    const int64_t call = ++c->call;
    int64_t lookup = 0;
    /* :749-762 the vectors whose data pointer is NULL, duplicates included; :775-793 one read per
     * distinct physical page, in order of first appearance */
    int64_t *pages = malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) {
        const int64_t v = ids[i];
        if (t->resident[v]) { t->hit_ids++; continue; }
        t->miss_ids++;
        const int64_t k = t->loc[v];
        if (t->stamp[k] == call) continue;
        t->stamp[k] = call;
        pages[lookup++] = k;
    }
    t->read_pages += lookup;   /* :802 */
    /* :865-886 a written page installs the vectors its reverse-table entry lists (wherever they live
     * now); an initial page installs the whole home page */
    for (int64_t q = 0; q < lookup; q++) {
        const int64_t k = pages[q];
        if (k >= t->n_init_pages) {
            const int32_t *r = t->log_rows + (k - t->n_init_pages) * vpo;
            for (int j = 0; j < vpo; j++) t->resident[r[j]] = 1;
        } else {
            for (int64_t v = k * vpo; v < (k + 1) * vpo && v < t->rows; v++) t->resident[v] = 1;
        }
        if (pages_out) pages_out[q] = k;
    }
    free(pages);
    /* lru_page: a listed vector counts the hit and stays where it is; a new one goes to the front */
    for (int64_t i = 0; i < n; i++) {
        const int64_t v = ids[i];
        if (t->listed[v]) {
            t->cnt[v]++;
        } else {
            t->listed[v] = 1;
            t->cnt[v] = 0;
            fifo_push_front(t, (int32_t)v);
        }
    }
    return lookup;
}

/*
 * One call of cache2::Reclaim_vector_page (emb_cache2.cpp:397-633): the oldest list entry leaves,
 * its vector moves into the reclaim buffer; unlisted resident vectors of its home page are dropped;
 * a full buffer (ws_opt * vpo vectors) becomes ws_opt new pages.  Returns the pages written (0 or ws_opt).
 */
int orc_veccache_reclaim_table(orc_veccache *c, int ti) {
    vc_table *t = &c->t[ti];
    const int vpo = c->vpo;
    const int64_t W = (int64_t)c->ws * vpo;
    if (t->fifo_front == t->fifo_back) return 0;
    const int64_t vec = t->fifo[t->fifo_back];                  /* :413-414, :429 */
    const int64_t start = (vec / vpo) * vpo;                    /* :431 */
    for (int64_t i = start; i < start + vpo && i < t->rows; i++) {
        if (i == vec) {
            int dup = 0;                                        /* :440-450 already waiting in the buffer */
            for (int64_t q = 0; q < t->rc_n; q++) dup |= t->rc_row[q] == (int32_t)i;
            if (dup) continue;
            t->rc_row[t->rc_n] = (int32_t)vec;                  /* :452 */
            t->rc_cnt[t->rc_n] = t->cnt[vec];
            t->rc_n++;
            t->resident[i] = 0;                                 /* :467-485 copied to the write buffer, pointer cleared */
            if (t->rc_n == W) break;                            /* :486-489 */
        } else if (t->resident[i] && !t->listed[i]) {           /* :492-496 */
            t->resident[i] = 0;
        }
    }
    t->listed[vec] = 0;                                         /* :501-502 */
    t->fifo_back++;
    if (t->rc_n < W) return 0;
    /* :507-629.  hot_page_list is a copy of an empty list (:406) and hot_page_count > 0
     * (src/dlrmtest2.cpp:539), so every page takes the first branch (:559-568): the vectors' hot-table
     * entries and the page's reverse-table entry are set; transfer_PA prefers the hot table (:714-717) */
    if (t->n_log_pages + c->ws > t->log_cap) {
        t->log_cap = t->log_cap ? t->log_cap * 2 : 1024;
        t->log_rows = realloc(t->log_rows, (size_t)t->log_cap * vpo * sizeof(int32_t));
    }
    if (t->n_init_pages + t->n_log_pages + c->ws > t->stamp_cap) {
        const int64_t nc = (t->n_init_pages + t->n_log_pages + c->ws) * 2;
        t->stamp = realloc(t->stamp, (size_t)nc * sizeof(int64_t));
        memset(t->stamp + t->stamp_cap, 0, (size_t)(nc - t->stamp_cap) * sizeof(int64_t));
        t->stamp_cap = nc;
    }
    for (int i = 0; i < c->ws; i++) {
        const int64_t page = t->n_init_pages + t->n_log_pages;
        for (int j = 0; j < vpo; j++) {
            const int32_t v = t->rc_row[(int64_t)i * vpo + j];
            t->loc[v] = (int32_t)page;
            t->log_rows[t->n_log_pages * vpo + j] = v;
        }
        t->n_log_pages++;
    }
    t->write_pages += c->ws;                                    /* :613 */
    c->page_count -= c->ws;                                     /* :614 */
    t->rc_n = 0;                                                /* :627 */
    return c->ws;
}

/* reclaim_emb_cache (src/dlrmtest2.cpp:462-481): per table, while the list is longer than cache_size */
int64_t orc_veccache_reclaim(orc_veccache *c) {
    int64_t calls = 0;
    for (int i = 0; i < c->T; i++)
        while (c->t[i].fifo_front - c->t[i].fifo_back > c->cache_size) {
            orc_veccache_reclaim_table(c, i);
            calls++;
        }
    return calls;
}

/* save_emb (src/dlrmtest2.cpp:169-177): a vector of the batch that reclaim has dropped in the meantime
 * gets a fresh buffer -- resident again, but not listed */
void orc_veccache_save(orc_veccache *c, int ti, const int64_t *ids, int64_t n) {
    vc_table *t = &c->t[ti];
    for (int64_t i = 0; i < n; i++) t->resident[ids[i]] = 1;
}

int orc_veccache_vpo(const orc_veccache *c) { return c->vpo; }
int64_t orc_veccache_page_count(const orc_veccache *c) { return c->page_count; }
int64_t orc_veccache_list_len(const orc_veccache *c, int t) { return c->t[t].fifo_front - c->t[t].fifo_back; }
int64_t orc_veccache_pending(const orc_veccache *c, int t) { return c->t[t].rc_n; }
int64_t orc_veccache_written_pages(const orc_veccache *c, int t) { return c->t[t].n_log_pages; }

/* which: 0 read_pages, 1 write_pages, 2 hit_ids, 3 miss_ids; out[T]; reset zeroes read/write like
 * cache2::read_page / write_page (include/data2.h:263-274) */
void orc_veccache_counters(orc_veccache *c, int which, int64_t *out, int reset) {
    for (int i = 0; i < c->T; i++) {
        vc_table *t = &c->t[i];
        int64_t *p = which == 0 ? &t->read_pages : which == 1 ? &t->write_pages : which == 2 ? &t->hit_ids : &t->miss_ids;
        out[i] = *p;
        if (reset && which < 2) *p = 0;
    }
}

/* per-vector state of table t: resident, listed (0/1), cnt, loc (page) -- any of the outputs may be NULL */
void orc_veccache_export(const orc_veccache *c, int ti, uint8_t *resident, uint8_t *listed, int32_t *cnt, int32_t *loc) {
    const vc_table *t = &c->t[ti];
    if (resident) memcpy(resident, t->resident, (size_t)t->rows);
    if (listed) memcpy(listed, t->listed, (size_t)t->rows);
    if (cnt) memcpy(cnt, t->cnt, (size_t)t->rows * sizeof(int32_t));
    if (loc) memcpy(loc, t->loc, (size_t)t->rows * sizeof(int32_t));
}

/* the list of table t, oldest entry first; returns the number written (<= cap) */
int64_t orc_veccache_fifo_order(const orc_veccache *c, int ti, int64_t *out, int64_t cap) {
    const vc_table *t = &c->t[ti];
    int64_t n = 0;
    for (int64_t q = t->fifo_back; q < t->fifo_front && n < cap; q++) out[n++] = t->fifo[q];
    return n;
}

/* vectors waiting in the reclaim buffer of table t, in eviction order */
int64_t orc_veccache_pending_rows(const orc_veccache *c, int ti, int64_t *out, int64_t cap) {
    const vc_table *t = &c->t[ti];
    int64_t n = 0;
    for (int64_t q = 0; q < t->rc_n && n < cap; q++) out[n++] = t->rc_row[q];
    return n;
}
