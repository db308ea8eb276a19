// C++ host side of the drop-in: the reference's model interface
// (dlrm_libtorch/src/dlrmtest.cpp + include/data.h) re-hosted on the C ABI of
// libdlrm_amd.so.  Same names, same argument meaning, same counter semantics;
// tensors are device buffers instead of libtorch CPU tensors.
//
//   reference                                   here
//   class cache (include/data.h:242-298)        dlrm_amd::cache
//   class cache2 (include/data2.h:219-287)      dlrm_amd::cache2   (the policy of src/dlrmtest2.cpp)
//   struct emb  (src/dlrmtest.cpp:144-210)      dlrm_amd::emb
//   struct Net2 (src/dlrmtest.cpp:217-278)      dlrm_amd::Net2
//   create_emb / save_emb / update_parameter    folded into emb (no mini-table is materialised:
//     (src/dlrmtest.cpp:53-107, :336-355)        rows are pooled and updated inside the cache's
//                                                slot pool) -- emb::update_parameter()
//   reclaim_emb_cache (src/dlrmtest.cpp:397)    dlrm_amd::reclaim_emb_cache
//
// Header only; link with -ldlrm_amd -lamdhip64.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/dlrm_amd.h"

namespace dlrm_amd {

inline void check(int rc, const char *what) {
    if (rc != DLRM_OK) throw std::runtime_error(std::string(what) + ": " + dlrm_last_error());
}
// The structs of the C ABI are passed by pointer: a library built from another revision of the
// header would read them with another layout.  Every adapter object checks this once.
inline void check_abi() {
    static const bool ok = [] {
        if (dlrm_version() != DLRM_ABI_VERSION)
            throw std::runtime_error("libdlrm_amd.so has ABI version " + std::to_string(dlrm_version()) +
                                     ", this program was compiled against " + std::to_string(DLRM_ABI_VERSION) +
                                     ": rebuild it");
        return true;
    }();
    (void)ok;
}
inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

template <typename T>
struct DeviceArray {
    T *p = nullptr;
    size_t n = 0;
    DeviceArray() = default;
    explicit DeviceArray(size_t count) { resize(count); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    DeviceArray(DeviceArray &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ~DeviceArray() { if (p) (void)hipFree(p); }
    void resize(size_t count) {
        if (count <= n) return;
        if (p) (void)hipFree(p);
        hip_check(hipMalloc(reinterpret_cast<void **>(&p), sizeof(T) * count), "hipMalloc");
        n = count;
    }
    void upload(const T *h, size_t count) {
        resize(count);
        hip_check(hipMemcpy(p, h, sizeof(T) * count, hipMemcpyHostToDevice), "hipMemcpy H2D");
    }
    void upload(const std::vector<T> &h) { upload(h.data(), h.size()); }
    std::vector<T> download(size_t count) const {
        std::vector<T> h(count);
        hip_check(hipMemcpy(h.data(), p, sizeof(T) * count, hipMemcpyDeviceToHost), "hipMemcpy D2H");
        return h;
    }
    void zero() { if (p) hip_check(hipMemset(p, 0, sizeof(T) * n), "hipMemset"); }
};

// include/data.h:31-35
struct write_buffer {
    int read_page_count = 0;
    int write_page_count = 0;
};

// class cache (include/data.h:242-298): HBM page cache over pinned host DRAM.
class cache {
public:
    // cache(table_num, size, dimention) (include/data.h:250) plus the two sizes the reference takes
    // from globals / the device: the capacity threshold in pages (cache_size, src/dlrmtest.cpp:47)
    // and how many pages one batch can bring in on top of it (slot headroom).
    cache(int table_num, int size, int dimention, int64_t cache_pages, int64_t headroom_pages)
        : cache_size((int)cache_pages), page_count(0), T_(table_num), rows_(size), dim_(dimention) {
        check_abi();
        vector_page_offset = 4096 / (dimention * (int)sizeof(float));   // include/data.h:211
        write_buf = new write_buffer[table_num];
        stores_.resize(table_num, nullptr);
        std::vector<int64_t> rows(table_num, size);
        for (int t = 0; t < table_num; t++)
            check(dlrm_host_alloc(&stores_[t], (size_t)size * dimention * sizeof(float)), "dlrm_host_alloc");
        dlrm_cache_config cfg{};
        cfg.num_tables = table_num; cfg.dim = dimention; cfg.row_dtype = DLRM_F32;
        cfg.rows_per_page = vector_page_offset; cfg.evict_group = 8;   // nvm_dev_get_ws_opt() of the reference's device
        cfg.table_rows = rows.data(); cfg.h_table_store = stores_.data();
        cfg.num_slots = cache_pages + headroom_pages; cfg.cache_size = cache_pages; cfg.d_slot_pool = nullptr;
        check(dlrm_cache_create(&cfg, &h_), "dlrm_cache_create");
        void *pool = nullptr;
        check(dlrm_cache_storage(h_, &pool, &pool_rows_), "dlrm_cache_storage");
        pool_ = static_cast<float *>(pool);
        ids_dev_.resize(table_num);
        rows_dev_.resize(table_num);
        read_base_.assign(table_num, 0);
        write_base_.assign(table_num, 0);
    }
    ~cache() {
        (void)hipDeviceSynchronize();
        dlrm_cache_destroy(h_);
        for (void *p : stores_) dlrm_host_free(p);
        delete[] write_buf;
    }
    cache(const cache &) = delete;

    // embedding_table_init::write() (include/data.h:217): uniform [0,1) tables in the backing store
    void write(uint64_t seed = 0) {
        for (int t = 0; t < T_; t++)
            check(dlrm_table_init_uniform(stores_[t], DLRM_F32, (int64_t)rows_ * dim_, (seed << 20) + 7919ull * (t + 1), 0,
                                          nullptr), "dlrm_table_init_uniform");
        hip_check(hipDeviceSynchronize(), "sync");
    }

    // cache::get_emb_vectors (include/data.h:247): instead of host row pointers, returns the device
    // array holding, for every id, its row inside the slot pool (valid until the next call for the
    // same table).  Pages are faulted in, counted and touched exactly as in the reference.
    const int *get_emb_vectors(std::vector<int> &vectors, int table_ID) {
        std::vector<int64_t> offs(T_ + 1, 0);
        for (int t = table_ID + 1; t <= T_; t++) offs[t] = (int64_t)vectors.size();
        ids_dev_[table_ID].upload(vectors);
        rows_dev_[table_ID].resize(vectors.size());
        check(dlrm_cache_lookup(h_, ids_dev_[table_ID].p, DLRM_I32, offs.data(), rows_dev_[table_ID].p, nullptr),
              "dlrm_cache_lookup");
        refresh();
        return rows_dev_[table_ID].p;
    }

    // reclaim_emb_cache's loop (src/dlrmtest.cpp:399-403) runs inside the library
    void reclaim() {
        check(dlrm_cache_set_cache_size(h_, cache_size, nullptr), "dlrm_cache_set_cache_size");
        check(dlrm_cache_reclaim(h_, nullptr), "dlrm_cache_reclaim");
        refresh();
    }
    // cache::Reclaim_vector_page (include/data.h:267): the ws_opt least recently used pages of ONE table leave
    void Reclaim_vector_page(int table_ID) {
        check(dlrm_cache_reclaim_table(h_, table_ID, nullptr), "dlrm_cache_reclaim_table");
        refresh();
    }

    int get_pu() const { return 8; }   // geo->l.npugrp of the reference's device (CSV labels only)
    void read_page(int i) { reset_one(i, true); }     // include/data.h:274-279
    void write_page(int i) { reset_one(i, false); }   // include/data.h:280-285
    void dump_chunk_sector_stats(const std::string &) {}   // flash wear accounting: not applicable

    float *pool() const { return pool_; }               // replaces get_vec() (include/data.h:268)
    int64_t pool_rows() const { return pool_rows_; }
    int tables() const { return T_; }
    int dim() const { return dim_; }
    static constexpr bool sorts_ids = true;             // create_emb sorts a batch's ids (src/dlrmtest.cpp:62-63)
    static constexpr int counter_stride = DLRM_CACHE_NUM_COUNTERS;
    void restore(std::vector<int> &, int) {}            // pages of the batch are never dropped by its own reclaim
    void flush() { check(dlrm_cache_flush(h_, nullptr), "dlrm_cache_flush"); hip_check(hipDeviceSynchronize(), "sync"); }
    const float *store(int t) const { return static_cast<const float *>(stores_[t]); }
    int64_t error_flags() {
        dlrm_cache_state_t st;
        check(dlrm_cache_state(h_, &st, nullptr), "dlrm_cache_state");
        return st.error_flags;
    }

    // Cumulative per-table counters {read, write, hit, miss, evicted, resident} as of the last
    // refresh: what the reference's CSV cannot say (its rows only carry pages read / written).
    const std::vector<int64_t> &counters() {
        refresh();
        return last_;
    }

    // public state the reference driver reads (include/data.h:288-291)
    int cache_size;
    int page_count;
    int vector_page_offset;
    write_buffer *write_buf;

private:
    void refresh() {
        dlrm_cache_state_t st;
        check(dlrm_cache_state(h_, &st, nullptr), "dlrm_cache_state");
        page_count = (int)st.page_count;
        std::vector<int64_t> c((size_t)T_ * counter_stride);
        check(dlrm_cache_counters(h_, c.data(), 0, nullptr), "dlrm_cache_counters");
        for (int t = 0; t < T_; t++) {
            write_buf[t].read_page_count = (int)(c[t * counter_stride + 0] - read_base_[t]);
            write_buf[t].write_page_count = (int)(c[t * counter_stride + 1] - write_base_[t]);
        }
        last_ = c;
    }
    void reset_one(int i, bool read) {
        if (last_.empty()) refresh();
        if (read) { read_base_[i] = last_[i * counter_stride + 0]; write_buf[i].read_page_count = 0; }
        else { write_base_[i] = last_[i * counter_stride + 1]; write_buf[i].write_page_count = 0; }
    }

    dlrm_cache *h_ = nullptr;
    int T_, rows_, dim_;
    std::vector<void *> stores_;
    float *pool_ = nullptr;
    int64_t pool_rows_ = 0;
    std::vector<DeviceArray<int>> ids_dev_, rows_dev_;
    std::vector<int64_t> last_;
    std::vector<int64_t> read_base_, write_base_;   // [table] counter values at the last read_page / write_page
};

// reclaim_emb_cache (src/dlrmtest.cpp:397-412): evict down to cache_size, log the write counters
inline void reclaim_emb_cache(cache &c, std::ofstream *csv, double now) {
    c.reclaim();
    for (int i = 0; i < c.tables(); i++) {
        if (csv) *csv << now << ";" << i / c.get_pu() << ";" << i % c.get_pu() << ";" << ";" << c.write_buf[i].write_page_count << "\n";
        c.write_page(i);
    }
}

// class cache2 (include/data2.h:219-287): the vector-granular policy -- rows are resident one by one, a table
// lists the rows that were asked for (oldest first), evicted rows are repacked into new pages of the store.
class cache2 {
public:
    // cache2(table_num, size, dimention) (include/data2.h:228) plus the sizes the reference takes from globals
    // or leaves unbounded: listed rows per table (cache_size, src/dlrmtest2.cpp:53), rows of the HBM pool
    // (listed rows + rows that come along with a page) and pages a table may write.
    cache2(int table_num, int size, int dimention, int64_t cache_rows, int64_t num_slots, int64_t log_pages)
        : cache_size((int)cache_rows), page_count(0), hot_page_count(0), T_(table_num), rows_(size), dim_(dimention) {
        check_abi();
        vector_page_offset = 4096 / (dimention * (int)sizeof(float));   // include/data2.h:161
        write_buf = new write_buffer[table_num];
        stores_.resize(table_num, nullptr);
        std::vector<int64_t> rows(table_num, size);
        const int64_t n_init = ((int64_t)size + vector_page_offset - 1) / vector_page_offset;
        for (int t = 0; t < table_num; t++)
            check(dlrm_host_alloc(&stores_[t], (size_t)(n_init + log_pages) * 4096), "dlrm_host_alloc");
        dlrm_vcache_config cfg{};
        cfg.num_tables = table_num; cfg.dim = dimention; cfg.rows_per_page = vector_page_offset;
        cfg.evict_group = 8;   // nvm_dev_get_ws_opt() of the reference's device
        cfg.table_rows = rows.data(); cfg.h_table_store = stores_.data();
        cfg.log_pages_per_table = log_pages; cfg.num_slots = num_slots; cfg.cache_size = cache_rows; cfg.d_slot_pool = nullptr;
        check(dlrm_vcache_create(&cfg, &h_), "dlrm_vcache_create");
        void *pool = nullptr;
        check(dlrm_vcache_storage(h_, &pool, &pool_rows_), "dlrm_vcache_storage");
        pool_ = static_cast<float *>(pool);
        ids_dev_.resize(table_num);
        rows_dev_.resize(table_num);
        read_base_.assign(table_num, 0);
        write_base_.assign(table_num, 0);
    }
    ~cache2() {
        (void)hipDeviceSynchronize();
        dlrm_vcache_destroy(h_);
        for (void *p : stores_) dlrm_host_free(p);
        delete[] write_buf;
    }
    cache2(const cache2 &) = delete;

    // embedding_table_init2::write(): uniform [0,1) tables in the backing store
    void write(uint64_t seed = 0) {
        for (int t = 0; t < T_; t++)
            check(dlrm_table_init_uniform(stores_[t], DLRM_F32, (int64_t)rows_ * dim_, (seed << 20) + 7919ull * (t + 1), 0,
                                          nullptr), "dlrm_table_init_uniform");
        hip_check(hipDeviceSynchronize(), "sync");
    }

    // cache2::get_emb_vectors (include/data2.h:224): the device array holding, for every id (in the caller's
    // order), its row inside the slot pool
    const int *get_emb_vectors(std::vector<int> &vectors, int table_ID) {
        const std::vector<int64_t> offs = one_table(table_ID, vectors.size());
        ids_dev_[table_ID].upload(vectors);
        rows_dev_[table_ID].resize(vectors.size());
        check(dlrm_vcache_lookup(h_, ids_dev_[table_ID].p, DLRM_I32, offs.data(), rows_dev_[table_ID].p, nullptr),
              "dlrm_vcache_lookup");
        refresh();
        return rows_dev_[table_ID].p;
    }
    // save_emb's side effect (src/dlrmtest2.cpp:169-177): rows of the batch that reclaim dropped are resident again
    void restore(std::vector<int> &vectors, int table_ID) {
        const std::vector<int64_t> offs = one_table(table_ID, vectors.size());
        ids_dev_[table_ID].upload(vectors);
        check(dlrm_vcache_restore(h_, ids_dev_[table_ID].p, DLRM_I32, offs.data(), nullptr, nullptr), "dlrm_vcache_restore");
    }
    // reclaim_emb_cache's loops (src/dlrmtest2.cpp:465-470) run inside the library
    void reclaim() {
        check(dlrm_vcache_set_cache_size(h_, cache_size, nullptr), "dlrm_vcache_set_cache_size");
        check(dlrm_vcache_reclaim(h_, nullptr), "dlrm_vcache_reclaim");
        refresh();
    }
    void Reclaim_vector_page(int) { throw std::runtime_error("use reclaim_emb_cache(cache2&)"); }

    int get_pu() const { return 8; }
    void read_page(int i) { reset_one(i, true); }     // include/data2.h:263-268
    void write_page(int i) { reset_one(i, false); }   // include/data2.h:269-274
    void dump_chunk_sector_stats(const std::string &) {}

    float *pool() const { return pool_; }
    int64_t pool_rows() const { return pool_rows_; }
    int tables() const { return T_; }
    int dim() const { return dim_; }
    static constexpr bool sorts_ids = false;            // create_emb of dlrmtest2 keeps the request order (:63-67, :128)
    static constexpr int counter_stride = DLRM_VCACHE_NUM_COUNTERS;
    void flush() { hip_check(hipDeviceSynchronize(), "sync"); }   // rows are read back through read_rows
    // current values of rows [0, n) of one table, wherever they live
    std::vector<float> read_rows(int table, int n) {
        std::vector<int> ids(n);
        for (int i = 0; i < n; i++) ids[i] = i;
        DeviceArray<int> d_ids; d_ids.upload(ids);
        DeviceArray<float> out((size_t)n * dim_);
        const std::vector<int64_t> offs = one_table(table, (size_t)n);
        check(dlrm_vcache_read_rows(h_, d_ids.p, DLRM_I32, offs.data(), out.p, nullptr, nullptr), "dlrm_vcache_read_rows");
        return out.download((size_t)n * dim_);
    }
    int64_t error_flags() {
        dlrm_vcache_state_t st;
        check(dlrm_vcache_state(h_, &st, nullptr), "dlrm_vcache_state");
        return st.error_flags;
    }
    // cumulative per-table counters, DLRM_VCACHE_NUM_COUNTERS per table (the first six as in class cache)
    const std::vector<int64_t> &counters() {
        refresh();
        return last_;
    }

    // public state the reference driver reads or sets (include/data2.h:256-258, :281)
    int cache_size;
    int page_count;
    int hot_page_count;      // set by the driver (src/dlrmtest2.cpp:539); the classification it feeds is inert as shipped
    int vector_page_offset;
    write_buffer *write_buf;

private:
    std::vector<int64_t> one_table(int table_ID, size_t n) const {
        std::vector<int64_t> offs(T_ + 1, 0);
        for (int t = table_ID + 1; t <= T_; t++) offs[t] = (int64_t)n;
        return offs;
    }
    void refresh() {
        dlrm_vcache_state_t st;
        check(dlrm_vcache_state(h_, &st, nullptr), "dlrm_vcache_state");
        page_count = (int)st.page_count;
        std::vector<int64_t> c((size_t)T_ * counter_stride);
        check(dlrm_vcache_counters(h_, c.data(), 0, nullptr), "dlrm_vcache_counters");
        for (int t = 0; t < T_; t++) {
            write_buf[t].read_page_count = (int)(c[t * counter_stride + 0] - read_base_[t]);
            write_buf[t].write_page_count = (int)(c[t * counter_stride + 1] - write_base_[t]);
        }
        last_ = c;
    }
    void reset_one(int i, bool read) {
        if (last_.empty()) refresh();
        if (read) { read_base_[i] = last_[i * counter_stride + 0]; write_buf[i].read_page_count = 0; }
        else { write_base_[i] = last_[i * counter_stride + 1]; write_buf[i].write_page_count = 0; }
    }

    dlrm_vcache *h_ = nullptr;
    int T_, rows_, dim_;
    std::vector<void *> stores_;
    float *pool_ = nullptr;
    int64_t pool_rows_ = 0;
    std::vector<DeviceArray<int>> ids_dev_, rows_dev_;
    std::vector<int64_t> last_;
    std::vector<int64_t> read_base_, write_base_;   // [table] counter values at the last read_page / write_page
};

// reclaim_emb_cache (src/dlrmtest2.cpp:462-481)
inline void reclaim_emb_cache(cache2 &c, std::ofstream *csv, double now) {
    c.reclaim();
    for (int i = 0; i < c.tables(); i++) {
        if (csv) *csv << now << ";" << i / c.get_pu() << ";" << i % c.get_pu() << ";" << ";" << c.write_buf[i].write_page_count << "\n";
        c.write_page(i);
    }
}

// struct emb (src/dlrmtest.cpp:144-210; src/dlrmtest2.cpp:215-290): the EmbeddingBags of one batch, over either cache.
template <class Cache>
struct emb_t {
    // x: the batch as the reference lays it out, [batch][dense_feature + T*lookup_num] floats with the
    // ids float-encoded after the dense features (src/dlrmtest.cpp:150, :309-313)
    emb_t(const float *x, int batch, int dense_feature, int lookup_num, Cache &c, std::ofstream *csv, double now)
        : c_(c), B_(batch), L_(lookup_num) {
        const int T = c.tables(), row = dense_feature + T * lookup_num;
        std::vector<int> all_rows((size_t)T * batch * lookup_num);
        for (int t = 0; t < T; t++) {
            // create_emb: float -> int, sort, gather through the cache (src/dlrmtest.cpp:60-66)
            std::vector<int> ids((size_t)batch * lookup_num);
            for (int b = 0; b < batch; b++)
                for (int l = 0; l < lookup_num; l++) ids[(size_t)b * lookup_num + l] = (int)x[(size_t)b * row + dense_feature + t * lookup_num + l];
            if (Cache::sorts_ids) std::sort(ids.begin(), ids.end());
            const int *d_rows = c.get_emb_vectors(ids, t);
            ids_.push_back(ids);
            hip_check(hipMemcpy(all_rows.data() + (size_t)t * batch * lookup_num, d_rows, sizeof(int) * ids.size(),
                                hipMemcpyDeviceToHost), "rows D2H");
        }
        for (int i = 0; i < T; i++) {   // :162-167
            if (csv) *csv << now << ";" << i / c.get_pu() << ";" << i % c.get_pu() << ";" << c.write_buf[i].read_page_count << "\n";
            c.read_page(i);
        }
        rows_.upload(all_rows);
        // forward()'s lookup tensor is arange(B*L).view(B,L) per table (:172-187): bag b of table t
        // pools sorted positions [b*L, (b+1)*L)
        std::vector<int> off((size_t)T * batch + 1);
        for (size_t g = 0; g < off.size(); g++) off[g] = (int)(g * lookup_num);
        off_.upload(off);
        pooled_.resize((size_t)batch * T * c.dim());
    }

    dlrm_embbag desc(int first_table, int n_tables) const {
        dlrm_embbag e{};
        e.d_weight = c_.pool(); e.d_table_row_offsets = nullptr; e.total_rows = c_.pool_rows();
        e.num_tables = 1; e.dim = c_.dim(); e.row_dtype = DLRM_F32; e.idx_dtype = DLRM_I32;
        // every table indexes the same slot pool, so the T tables of the batch are one "table" with T*B bags
        e.d_indices = rows_.p + (size_t)first_table * B_ * L_;
        e.d_offsets = off_.p;   // offsets are relative to d_indices: bag g starts at g*L
        e.n_indices = (int64_t)n_tables * B_ * L_; e.batch = (int64_t)n_tables * B_;
        e.include_last_offset = 1;
        e.pooling = DLRM_POOL_MEAN;   // from_pretrained() ignores the kSum option of :79 and builds a mean-mode module (:80)
        return e;
    }

    // emb::forward (src/dlrmtest.cpp:170-206): [B, T*D] pooled vectors (device).  Bags are laid out
    // table-major in one T*B-bag call; out row stride D places them as [T][B][D], then the caller's
    // layout [B][T*D] is produced by one call per table writing at column t*D.
    float *forward(float *d_out, int64_t out_ld, int col0) {
        const int T = c_.tables();
        for (int t = 0; t < T; t++) {
            dlrm_embbag e = desc(t, 1);
            check(dlrm_embbag_forward(&e, d_out + col0 + (size_t)t * c_.dim(), out_ld, nullptr), "dlrm_embbag_forward");
        }
        return d_out;
    }

    // update_parameter -> save_emb (src/dlrmtest.cpp:88-107, :336-355) fused with the optimizer step
    // that precedes it: rows of `table` are updated in place inside the cache.  d_grad: gradient of
    // the pooled output of that table, [B][ld].  last_wins = the reference's duplicate semantics.
    void update_parameter(int table, const float *d_grad, int64_t ld, float lr, bool last_wins, DeviceArray<char> &ws) {
        dlrm_embbag e = desc(table, 1);
        dlrm_sparse_opt o{};
        o.optimizer = DLRM_OPT_SGD; o.dup_mode = last_wins ? DLRM_DUP_LAST_WINS : DLRM_DUP_ACCUMULATE; o.lr = lr;
        size_t need = 0;
        check(dlrm_embbag_backward_workspace_bytes(&e, &need), "workspace");
        ws.resize(need);
        check(dlrm_embbag_backward_update(&e, d_grad, ld, &o, ws.p, ws.n, nullptr), "dlrm_embbag_backward_update");
    }

    // the batch's rows that the reclaim after the lookups has dropped come back (cache2 only)
    void restore() {
        for (int t = 0; t < c_.tables(); t++) c_.restore(ids_[t], t);
    }

    Cache &c_;
    int B_, L_;
    std::vector<std::vector<int>> ids_;
    DeviceArray<int> rows_, off_;
    DeviceArray<float> pooled_;
};
using emb = emb_t<cache>;
using emb2 = emb_t<cache2>;

// struct Net2 (src/dlrmtest.cpp:217-278): bottom dense->128->32, concat with the pooled tables,
// (32 + T*D)->32->1, ReLU after every layer.
struct Net2 {
    int dense, T, D, cat;
    DeviceArray<float> W[4], b[4];
    Net2(int dense_feature, int tables, int dim) : dense(dense_feature), T(tables), D(dim), cat(32 + tables * dim) {
        check_abi();
    }
    // create_fc binds global randn weights (src/dlrmtest.cpp:110-140); here the caller supplies them
    void set_layer(int i, const std::vector<float> &w, const std::vector<float> &bias) { W[i].upload(w); b[i].upload(bias); }
    // Net2::forward(x, y) (src/dlrmtest.cpp:254-272): fc1, fc2 on the dense features, torch::cat with the pooled
    // embeddings, fc3, fc4, ReLU after every layer.  x: [B, dense] device floats; `pooled` writes the EmbeddingBag
    // outputs in place of the concatenation: it is called with (buffer, row stride, first column) and must fill
    // columns [32, 32 + T*D) of every row -- emb::forward does exactly that, so
    //     net2.forward(d_x, B, [&](float *cat, int64_t ld, int col0) { embedding.forward(cat, ld, col0); })
    // is the reference's  net2->forward(dense, embedding->forward(...))  without the copy.  Returns the [B]
    // predictions (device).  The activations stay available for the backward pass (h1, cat, h3, pred).
    template <class Pooled>
    const float *forward(const float *x, int batch, Pooled &&pooled) {
        h1.resize((size_t)batch * 128); catbuf.resize((size_t)batch * cat); h3.resize((size_t)batch * 32); pred.resize(batch);
        linear(x, dense, W[0].p, dense, b[0].p, h1.p, 128, batch, 128, dense);
        linear(h1.p, 128, W[1].p, 128, b[1].p, catbuf.p, cat, batch, 32, 128);
        pooled(catbuf.p, (int64_t)cat, 32);
        linear(catbuf.p, cat, W[2].p, cat, b[2].p, h3.p, 32, batch, 32, cat);
        linear(h3.p, 32, W[3].p, 32, b[3].p, pred.p, 1, batch, 1, 32);
        return pred.p;
    }
    DeviceArray<float> h1, catbuf, h3, pred;

    static void linear(const float *x, int64_t x_ld, const float *w, int64_t w_ld, const float *bias, float *y, int64_t y_ld,
                       int64_t M, int N, int K) {
        dlrm_linear l{};
        l.M = M; l.N = N; l.K = K; l.d_x = x; l.x_ld = x_ld; l.d_w = w; l.w_ld = w_ld; l.d_bias = bias; l.d_y = y; l.y_ld = y_ld;
        l.act = DLRM_ACT_RELU;
        check(dlrm_linear_forward(&l, nullptr), "dlrm_linear_forward");
    }
};

}  // namespace dlrm_amd
