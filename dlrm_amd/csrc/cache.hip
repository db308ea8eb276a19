// HBM-resident embedding page cache over a pinned host-DRAM backing store.
//
// Replaces class cache (dlrm_libtorch/include/data.h:242-298,
// src/emb_cache.cpp) and the policy driver reclaim_emb_cache
// (src/dlrmtest.cpp:397-412), with host DRAM standing in for the OCSSD.
//
// State (all in HBM): a direct page map (slot of every page of every table, or
// -1), the slot pool [num_slots][rows_per_page][dim] that the EmbeddingBag
// kernels read and update in place, per-slot (page, last-use epoch), a ring of
// free slots and per-table counters.  288 GB of HBM make a direct map cheaper
// than any hash: 4 bytes per page.
//
// Recency is kept per lookup call ("epoch"); inside one call pages count as
// touched in ascending page order.  That is exactly the order the reference's
// std::list ends up in, because its driver sorts the ids of a batch before
// calling get_emb_vectors (src/dlrmtest.cpp:62-63): LRU order == ascending
// (epoch, page).  So the victims of a reclaim are, per table, the smallest
// (epoch, page) keys -- found with an age histogram + a sort of the few
// candidates instead of a linked list walk.
//
// Determinism: missed pages are sorted, then take free slots in ring order;
// evicted slots return to the ring in eviction order (pass-major, table, LRU
// first).  The page->slot map and every counter are therefore pure functions
// of the call sequence and can be compared bit for bit with the CPU
// restatement.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace dlrm {

constexpr int kMaxCacheTables = 64;
constexpr int kAgeBins = 1024;
constexpr int kCacheThreads = 256;
// Workgroups of the kernels that move pages over the host link.  Their lanes spend microseconds
// waiting for each 16-byte piece; a grid that fills the chip would hold every wave slot for that long
// and starve the kernels of another stream (a prefetching lookup runs beside the MLP).  128 K lanes in
// flight keep the link busy and leave most wave slots free.
constexpr int kLinkBlocks = 512;
constexpr int kNumCounters = DLRM_CACHE_NUM_COUNTERS;  // read_pages, write_pages, hit_ids, miss_ids, evicted_pages,
                                                       // resident_pages, pages copied back to the store
enum { C_READ = 0, C_WRITE = 1, C_HIT = 2, C_MISS = 3, C_EVICT = 4, C_RESIDENT = 5, C_WRITTEN_BACK = 6 };
static_assert(kNumCounters == 7, "counter layout");
enum { S_HEAD = 0, S_TAIL, S_PAGE_COUNT, S_CACHE_SIZE, S_N_MISS, S_N_CAND, S_ERROR, S_EPOCH, S_PASSES, S_N_ASSIGNED,
       S_COUNT };
enum { ERR_BAD_ID = 1, ERR_POOL_EXHAUSTED = 2, ERR_EVICTED_CURRENT = 4 };

struct TableGeom {
    int T;
    int vpo;            // rows per page
    int ws;             // pages evicted per table per pass
    int dim;
    int esz;            // bytes per element
    int64_t page_bytes;
    int64_t rows[kMaxCacheTables];
    int64_t page_base[kMaxCacheTables + 1];  // first global page id of each table
    char *store[kMaxCacheTables];            // pinned host backing store per table (device accessible)
    float *state_store[kMaxCacheTables];     // one float per row beside it (optimizer state), or all NULL
    int has_state;
};

struct IdLayout {
    int64_t off[kMaxCacheTables + 1];  // ids of table t sit at positions [off[t], off[t+1])
};

struct CacheDev {
    int32_t *page_map;
    int32_t *slot_gp;
    uint32_t *slot_epoch;
    uint8_t *slot_dirty;   // rows of the slot were updated since it was filled / last written back
    int dirty_only;        // 1: only dirty pages are copied back to the store; 0: every victim (the reference)
    int32_t *ring;
    int64_t *scalars;
    int64_t *counters;   // [T][kNumCounters]
    int64_t *k_t;        // [T] victims per table of the current reclaim
    int32_t *thr_age;    // [T]
    uint32_t *hist;      // [T][kAgeBins]
    int32_t *miss, *miss_sorted;
    uint64_t *cand_key, *cand_key_sorted;
    int32_t *cand_val, *cand_val_sorted;
    char *data;
    float *state;          // [S * vpo] the rows' optimizer state, moved with their pages; NULL = none
    int64_t S;
    // asynchronous write-back (all NULL unless enabled): victims are parked here before their slots are released
    uint32_t *pending;     // one bit per page: parked by the last reclaim, its way to the store may not be over
    char *stage_data;      // [stage_cap] pages
    float *stage_state;    // [stage_cap * vpo]
    int32_t *stage_gp;     // [stage_cap] page of a parked entry, -1 = nothing to write
};

}  // namespace dlrm

struct dlrm_cache {
    dlrm::TableGeom g;
    dlrm::CacheDev d;
    int64_t total_pages;
    int64_t miss_cap;        // capacity of the miss buffers (ids per lookup call)
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int64_t *h_scalars;      // pinned staging for scalar read-back
    int64_t *h_reclaim;      // pinned copy of the scalars taken by reclaim_begin
    hipEvent_t reclaim_ev;   // recorded behind that copy
    bool reclaim_pending;
    bool owns_pool;
    bool owns_state;
    int protect_age;         // pages used in the last protect_age lookup calls are never victims
    int reclaim_table;       // table of a pending single-table reclaim, -1 = every table
    int64_t stage_cap;       // pages the parking area holds; 0 = victims are written back inside the reclaim
    hipStream_t wb_stream;   // carries parked pages to the store
    hipEvent_t wb_ready, wb_done;
    bool wb_recorded;        // wb_done has been recorded at least once
};

namespace dlrm {

__device__ inline int table_of_page(const TableGeom &g, int64_t gp) {
    int lo = 0, hi = g.T;  // page_base[lo] <= gp < page_base[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (gp >= g.page_base[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ inline int table_of_pos(const IdLayout &l, int T, int64_t i) {
    int lo = 0, hi = T;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (i >= l.off[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- lookup ----------------------------------------------------------------

// Wave-level aggregation of counter updates: lanes of a wave almost always work on the same table
// (ids and sorted pages are grouped by table), so one lane adds the wave's total instead of 64 lanes
// queueing on one address.  `valid` lanes carry table `t` and a 0/1 flag per counter.
__device__ inline void wave_add2(int64_t *counters, bool valid, int t, int ca, bool fa, int cb, bool fb) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int t0 = __shfl(t, leader, 64);
        const bool mine = valid && t == t0;
        const unsigned long long same = __ballot(mine);
        const int na = __popcll(__ballot(mine && fa)), nb = __popcll(__ballot(mine && fb));
        if (lane == leader) {
            if (na) atomicAdd((unsigned long long *)&counters[t0 * kNumCounters + ca], (unsigned long long)na);
            if (nb) atomicAdd((unsigned long long *)&counters[t0 * kNumCounters + cb], (unsigned long long)nb);
        }
        todo &= ~same;
    }
}

// position for every lane with `take` in a list whose length lives at *count (one atomic per wave)
__device__ inline int64_t wave_append(int64_t *count, bool take) {
    const int lane = threadIdx.x & 63;
    const unsigned long long w = __ballot(take);
    if (w == 0) return -1;
    const int leader = __ffsll((long long)w) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd((unsigned long long *)count, (unsigned long long)__popcll(w));
    base = __shfl(base, leader, 64);
    return take ? (int64_t)base + __popcll(w & ((1ull << lane) - 1ull)) : -1;
}

// cache::read() miss detection (src/emb_cache.cpp:410-438) + lru_page() touch (:317-352)
template <typename IDX>
__global__ __launch_bounds__(kCacheThreads) void probe_kernel(TableGeom g, CacheDev d, IdLayout l, const IDX *ids,
                                                               int64_t n) {
    const uint32_t epoch = (uint32_t)d.scalars[S_EPOCH];
    // every lane of a wave makes the same number of trips (the ballots below need the whole wave)
    for (int64_t base = (int64_t)blockIdx.x * kCacheThreads; base < n; base += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = base + threadIdx.x;
        bool valid = i < n, hit = false, claim = false;
        int t = 0;
        int64_t gp = 0;
        if (valid) {
            t = table_of_pos(l, g.T, i);
            const int64_t id = (int64_t)ids[i];
            if (id < 0 || id >= g.rows[t]) {
                atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_BAD_ID);
                valid = false;
            } else {
                gp = g.page_base[t] + id / g.vpo;
                const int32_t s = d.page_map[gp];
                hit = s >= 0;
                if (hit) d.slot_epoch[s] = epoch;   // every writer of this call stores the same value
                else claim = s == -1 && atomicCAS(&d.page_map[gp], -1, -2) == -1;   // first to see the page missing
            }
        }
        wave_add2(d.counters, valid, t, C_HIT, hit, C_MISS, !hit);
        const int64_t pos = wave_append(&d.scalars[S_N_MISS], claim);
        if (claim) d.miss[pos] = (int32_t)gp;
    }
}

// missed pages (ascending) take free slots in ring order (emb_cache.cpp:502-512 installs the pages)
__global__ __launch_bounds__(kCacheThreads) void assign_kernel(TableGeom g, CacheDev d) {
    const int64_t n_miss = d.scalars[S_N_MISS];
    const int64_t head = d.scalars[S_HEAD], avail = d.scalars[S_TAIL] - head;
    const uint32_t epoch = (uint32_t)d.scalars[S_EPOCH];
    for (int64_t base = (int64_t)blockIdx.x * kCacheThreads; base < n_miss; base += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n_miss;
        bool placed = false;
        int t = 0;
        if (valid) {
            const int64_t gp = d.miss_sorted[i];
            t = table_of_page(g, gp);
            if (i < avail) {
                const int32_t s = d.ring[(head + i) % d.S];
                d.page_map[gp] = s;
                d.slot_gp[s] = (int32_t)gp;
                d.slot_epoch[s] = epoch;
                d.slot_dirty[s] = 0;
                placed = true;
            } else {
                d.page_map[gp] = -1;
                atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_POOL_EXHAUSTED);
            }
        }
        wave_add2(d.counters, valid, t, C_READ, placed, C_RESIDENT, placed);
    }
}

// bytes of page gp that hold rows (the last page of a table may be partial)
__device__ inline int64_t valid_page_bytes(const TableGeom &g, int t, int64_t page) {
    const int64_t rows_left = g.rows[t] - page * g.vpo;
    const int64_t r = rows_left < g.vpo ? rows_left : g.vpo;
    return r * g.dim * g.esz;
}

// backing store -> slot: the device read replaces nvm_cmd_read + decode (emb_cache.cpp:451-497).
// One lane moves 16 bytes; a page is covered by consecutive lanes (coalesced on both sides).
// phase -1: every missed page; 0: those whose page is not on its way to the store; 1: those that are (launched
// once that write-back has ended)
__device__ inline bool fill_now(const CacheDev &d, int64_t gp, int phase) {
    if (phase < 0) return true;
    const bool pending = (d.pending[gp >> 5] >> (gp & 31)) & 1u;
    return pending == (phase == 1);
}

__global__ __launch_bounds__(kCacheThreads) void fill_kernel(TableGeom g, CacheDev d, int phase) {
    const int64_t n_miss = d.scalars[S_N_MISS];
    const int64_t avail = d.scalars[S_TAIL] - d.scalars[S_HEAD];
    const int64_t n = n_miss < avail ? n_miss : avail;
    const int64_t chunks = g.page_bytes / 16;
    const int64_t total = n * chunks;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / chunks, c = w - i * chunks;
        const int64_t gp = d.miss_sorted[i];
        if (!fill_now(d, gp, phase)) continue;
        const int t = table_of_page(g, gp);
        const int64_t page = gp - g.page_base[t];
        if (c * 16 < valid_page_bytes(g, t, page)) {
            const int32_t s = d.page_map[gp];
            const uint4 v = *reinterpret_cast<const uint4 *>(g.store[t] + page * g.page_bytes + c * 16);
            *reinterpret_cast<uint4 *>(d.data + (int64_t)s * g.page_bytes + c * 16) = v;
        }
    }
    if (!g.has_state) return;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < n * g.vpo;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / g.vpo, r = w - i * g.vpo;
        const int64_t gp = d.miss_sorted[i];
        if (!fill_now(d, gp, phase)) continue;
        const int t = table_of_page(g, gp);
        const int64_t row = (gp - g.page_base[t]) * g.vpo + r;
        if (row < g.rows[t]) d.state[(int64_t)d.page_map[gp] * g.vpo + r] = g.state_store[t][row];
    }
}

// row index of every id inside the slot pool (what get_emb_vectors returns as row pointers, :364-367)
template <typename IDX>
__global__ __launch_bounds__(kCacheThreads) void translate_kernel(TableGeom g, CacheDev d, IdLayout l, const IDX *ids,
                                                                   int64_t n, IDX *rows_out) {
    for (int64_t i = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kCacheThreads) {
        const int t = table_of_pos(l, g.T, i);
        const int64_t id = (int64_t)ids[i];
        IDX r = (IDX)-1;
        if (id >= 0 && id < g.rows[t]) {
            const int32_t s = d.page_map[g.page_base[t] + id / g.vpo];
            if (s >= 0) r = (IDX)((int64_t)s * g.vpo + id % g.vpo);
        }
        rows_out[i] = r;
    }
}

__global__ void begin_lookup_kernel(CacheDev d) {
    d.scalars[S_EPOCH] += 1;   // every call is one step of recency
    d.scalars[S_N_MISS] = 0;
    d.scalars[S_N_ASSIGNED] = 0;
}

__global__ void reset_rw_counters_kernel(int64_t *cnt, int T) {
    const int t = threadIdx.x;
    if (t < T) { cnt[t * kNumCounters + C_READ] = 0; cnt[t * kNumCounters + C_WRITE] = 0; }
}

__global__ void finish_lookup_kernel(CacheDev d) {
    const int64_t n_miss = d.scalars[S_N_MISS];
    const int64_t avail = d.scalars[S_TAIL] - d.scalars[S_HEAD];
    const int64_t n = n_miss < avail ? n_miss : avail;
    d.scalars[S_HEAD] += n;
    d.scalars[S_PAGE_COUNT] += n;   // page_count += lookup (emb_cache.cpp:441)
    d.scalars[S_N_ASSIGNED] = n;
    d.scalars[S_N_MISS] = 0;
}

// ---- reclaim ---------------------------------------------------------------

// reclaim_emb_cache (src/dlrmtest.cpp:397-403): while (page_count > cache_size) one
// Reclaim_vector_page per table; each drops ws pages (emb_cache.cpp:223-238, :312)
// only_table >= 0: one Reclaim_vector_page(only_table) call, whatever page_count is (include/data.h:267)
__global__ void plan_reclaim_kernel(TableGeom g, CacheDev d, int only_table) {
    const int64_t pc = d.scalars[S_PAGE_COUNT], cs = d.scalars[S_CACHE_SIZE];
    const int64_t per_pass = (int64_t)g.T * g.ws;
    const int64_t passes = only_table >= 0 ? 1 : (pc > cs ? (pc - cs + per_pass - 1) / per_pass : 0);
    d.scalars[S_PASSES] = passes;
    d.scalars[S_N_CAND] = 0;
    for (int t = 0; t < g.T; t++) {
        const int64_t res = d.counters[t * kNumCounters + C_RESIDENT];
        const int64_t want = (only_table >= 0 && t != only_table) ? 0 : passes * g.ws;
        d.k_t[t] = want < res ? want : res;
    }
}

__global__ __launch_bounds__(kCacheThreads) void age_hist_kernel(TableGeom g, CacheDev d) {
    if (d.scalars[S_PASSES] == 0) return;
    // ages cluster in a few bins per table: count in LDS first, one global atomic per non-empty bin
    // and workgroup (tables beyond the LDS window go straight to global memory)
    constexpr int kLdsTables = 8;
    __shared__ unsigned int h[kLdsTables * kAgeBins];
    for (int i = threadIdx.x; i < kLdsTables * kAgeBins; i += kCacheThreads) h[i] = 0;
    __syncthreads();
    const uint32_t now = (uint32_t)d.scalars[S_EPOCH];
    for (int64_t s = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; s < d.S; s += (int64_t)gridDim.x * kCacheThreads) {
        const int32_t gp = d.slot_gp[s];
        if (gp < 0) continue;
        const int t = table_of_page(g, gp);
        uint32_t age = now - d.slot_epoch[s];
        if (age > kAgeBins - 1) age = kAgeBins - 1;
        if (t < kLdsTables) atomicAdd(&h[t * kAgeBins + age], 1u);
        else atomicAdd(&d.hist[t * kAgeBins + age], 1u);
    }
    __syncthreads();
    const int nt = g.T < kLdsTables ? g.T : kLdsTables;
    for (int i = threadIdx.x; i < nt * kAgeBins; i += kCacheThreads)
        if (h[i]) atomicAdd(&d.hist[i], h[i]);
}

// per table (one workgroup each): the youngest age a* >= protect such that the pages of age >= a* number at
// least k_t.  Pages younger than `protect` lookup calls are never victims; when fewer than k_t older pages
// exist, k_t shrinks to what there is.  Thread c owns the bins [4c, 4c+4).
__global__ __launch_bounds__(kAgeBins / 4) void threshold_kernel(TableGeom g, CacheDev d, int protect) {
    __shared__ unsigned tot[kAgeBins / 4];
    __shared__ int thr_s;
    __shared__ long long k_s;
    const int t = blockIdx.x, c = threadIdx.x;
    unsigned b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = 4 * c + j;
        b[j] = a >= protect ? d.hist[t * kAgeBins + a] : 0u;
        d.hist[t * kAgeBins + a] = 0;
    }
    tot[c] = b[0] + b[1] + b[2] + b[3];
    if (c == 0) { thr_s = kAgeBins; k_s = 0; }
    __syncthreads();
    const int64_t k = d.scalars[S_PASSES] == 0 ? 0 : d.k_t[t];
    int64_t above = 0;   // pages in bins beyond this thread's
    for (int cc = c + 1; cc < kAgeBins / 4; cc++) above += tot[cc];
    if (k > 0) {
        if (above < k && above + tot[c] >= k) {   // the threshold lies in this thread's bins
            int64_t cum = above;
            for (int j = 3; j >= 0; j--) {
                cum += b[j];
                if (cum >= k) { thr_s = 4 * c + j; break; }
            }
            k_s = k;
        }
        if (c == 0 && above + tot[0] < k) {       // fewer evictable pages than wanted: take them all
            k_s = above + tot[0];
            thr_s = k_s > 0 ? (protect > 0 ? protect : 0) : kAgeBins;
        }
    }
    __syncthreads();
    if (c == 0) {
        d.thr_age[t] = thr_s;
        d.k_t[t] = k_s;
    }
}

__global__ __launch_bounds__(kCacheThreads) void collect_candidates_kernel(TableGeom g, CacheDev d) {
    if (d.scalars[S_PASSES] == 0) return;
    const uint32_t now = (uint32_t)d.scalars[S_EPOCH];
    for (int64_t base = (int64_t)blockIdx.x * kCacheThreads; base < d.S; base += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t s = base + threadIdx.x;
        bool take = false;
        int t = 0;
        int32_t gp = -1;
        if (s < d.S) {
            gp = d.slot_gp[s];
            if (gp >= 0) {
                t = table_of_page(g, gp);
                uint32_t age = now - d.slot_epoch[s];
                if (age > kAgeBins - 1) age = kAgeBins - 1;
                take = (int)age >= d.thr_age[t];
            }
        }
        const int64_t pos = wave_append(&d.scalars[S_N_CAND], take);
        if (take) {
            // ascending key = table, then oldest first (the age saturates at 2^26 - 1 calls), then page id
            const uint32_t full_age = now - d.slot_epoch[s];
            const uint32_t inv_age = 0x3ffffffu - (full_age < 0x3ffffffu ? full_age : 0x3ffffffu);
            d.cand_key[pos] = ((uint64_t)t << 58) | ((uint64_t)inv_age << 32) | (uint32_t)gp;
            d.cand_val[pos] = (int32_t)s;
        }
    }
}

// rank of candidate i inside its table (candidates are sorted by table, epoch, page)
__device__ inline int64_t cand_rank(const uint64_t *keys, int64_t n, int64_t i, int t) {
    const uint64_t first_key = (uint64_t)t << 58;
    int64_t lo = 0, hi = i;  // lower bound of first_key in [0, i]
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < first_key) lo = mid + 1; else hi = mid;
    }
    return i - lo;
}

// Position of every candidate in eviction order (pass-major, then table, then LRU rank inside the
// pass), or -1 when it survives: candidate i of table t is a victim iff its rank r < k_t.
__global__ __launch_bounds__(kCacheThreads) void rank_victims_kernel(TableGeom g, CacheDev d, int64_t n_cand,
                                                                      int32_t *cand_pos) {
    for (int64_t i = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; i < n_cand;
         i += (int64_t)gridDim.x * kCacheThreads) {
        const uint64_t key = d.cand_key_sorted[i];
        const int t = (int)(key >> 58);
        const int64_t r = cand_rank(d.cand_key_sorted, n_cand, i, t);
        int64_t pos = -1;
        if (r < d.k_t[t]) {
            const int64_t p = r / g.ws;
            pos = r - p * g.ws;
            for (int u = 0; u < g.T; u++) {
                const int64_t ku = d.k_t[u];
                pos += ku < p * g.ws ? ku : p * g.ws;
                if (u < t) {
                    int64_t in_pass = ku - p * g.ws;
                    in_pass = in_pass < 0 ? 0 : (in_pass > g.ws ? g.ws : in_pass);
                    pos += in_pass;
                }
            }
        }
        cand_pos[i] = (int32_t)pos;
    }
}

// slot -> backing store for every victim: replaces the serialise + nvm_cmd_write of
// Reclaim_vector_page (emb_cache.cpp:259-272, :305); every evicted page is written (no dirty bit there)
__global__ __launch_bounds__(kCacheThreads) void writeback_victims_kernel(TableGeom g, CacheDev d, int64_t n_cand,
                                                                           const int32_t *cand_pos) {
    const int64_t chunks = g.page_bytes / 16;
    const int64_t total = n_cand * chunks;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / chunks, c = w - i * chunks;
        if (cand_pos[i] < 0) continue;
        const int32_t s = d.cand_val_sorted[i];
        if (d.dirty_only && !d.slot_dirty[s]) continue;   // the store already holds these rows
        const uint64_t key = d.cand_key_sorted[i];
        const int t = (int)(key >> 58);
        const int64_t gp = (int64_t)(uint32_t)key;
        const int64_t page = gp - g.page_base[t];
        if (c * 16 < valid_page_bytes(g, t, page)) {
            const uint4 v = *reinterpret_cast<const uint4 *>(d.data + (int64_t)s * g.page_bytes + c * 16);
            *reinterpret_cast<uint4 *>(g.store[t] + page * g.page_bytes + c * 16) = v;
        }
    }
    if (!g.has_state) return;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < n_cand * g.vpo;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / g.vpo, r = w - i * g.vpo;
        if (cand_pos[i] < 0) continue;
        const int32_t s = d.cand_val_sorted[i];
        if (d.dirty_only && !d.slot_dirty[s]) continue;
        const uint64_t key = d.cand_key_sorted[i];
        const int t = (int)(key >> 58);
        const int64_t row = ((int64_t)(uint32_t)key - g.page_base[t]) * g.vpo + r;
        if (row < g.rows[t]) g.state_store[t][row] = d.state[(int64_t)s * g.vpo + r];
    }
}

// Asynchronous write-back, first half (on the reclaim's stream, HBM to HBM): victim number `pos` is parked in entry
// `pos` of the parking area and its page is flagged as on its way, so its slot can be released at once.
__global__ __launch_bounds__(kCacheThreads) void park_victims_kernel(TableGeom g, CacheDev d, int64_t n_cand,
                                                                      const int32_t *cand_pos) {
    const int64_t chunks = g.page_bytes / 16;
    const int64_t total = n_cand * chunks;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / chunks, c = w - i * chunks;
        const int64_t pos = cand_pos[i];
        if (pos < 0) continue;
        const int32_t s = d.cand_val_sorted[i];
        if (d.dirty_only && !d.slot_dirty[s]) continue;   // the store already holds these rows (the entry stays -1)
        const int64_t gp = (int64_t)(uint32_t)d.cand_key_sorted[i];
        if (c == 0) {
            d.stage_gp[pos] = (int32_t)gp;
            atomicOr(&d.pending[gp >> 5], 1u << (gp & 31));
        }
        *reinterpret_cast<uint4 *>(d.stage_data + pos * g.page_bytes + c * 16) =
            *reinterpret_cast<const uint4 *>(d.data + (int64_t)s * g.page_bytes + c * 16);
    }
    if (!g.has_state) return;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < n_cand * g.vpo;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = w / g.vpo, r = w - i * g.vpo;
        const int64_t pos = cand_pos[i];
        if (pos < 0) continue;
        const int32_t s = d.cand_val_sorted[i];
        if (d.dirty_only && !d.slot_dirty[s]) continue;
        d.stage_state[pos * g.vpo + r] = d.state[(int64_t)s * g.vpo + r];
    }
}

// ... second half (on the cache's own stream, beside whatever the caller enqueues next): parked pages to the store
__global__ __launch_bounds__(kCacheThreads) void writeback_parked_kernel(TableGeom g, CacheDev d, int64_t n_parked) {
    const int64_t chunks = g.page_bytes / 16;
    const int64_t total = n_parked * chunks;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t pos = w / chunks, c = w - pos * chunks;
        const int64_t gp = d.stage_gp[pos];
        if (gp < 0) continue;
        const int t = table_of_page(g, gp);
        const int64_t page = gp - g.page_base[t];
        if (c * 16 < valid_page_bytes(g, t, page))
            *reinterpret_cast<uint4 *>(g.store[t] + page * g.page_bytes + c * 16) =
                *reinterpret_cast<const uint4 *>(d.stage_data + pos * g.page_bytes + c * 16);
    }
    if (!g.has_state) return;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < n_parked * g.vpo;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t pos = w / g.vpo, r = w - pos * g.vpo;
        const int64_t gp = d.stage_gp[pos];
        if (gp < 0) continue;
        const int t = table_of_page(g, gp);
        const int64_t row = (gp - g.page_base[t]) * g.vpo + r;
        if (row < g.rows[t]) g.state_store[t][row] = d.stage_state[pos * g.vpo + r];
    }
}

__global__ __launch_bounds__(kCacheThreads) void evict_kernel(TableGeom g, CacheDev d, int64_t n_cand,
                                                               const int32_t *cand_pos) {
    const uint32_t now = (uint32_t)d.scalars[S_EPOCH];
    const int64_t tail = d.scalars[S_TAIL];
    // every lane of a wave makes the same number of trips (wave_add2 ballots)
    for (int64_t base = (int64_t)blockIdx.x * kCacheThreads; base < n_cand; base += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t i = base + threadIdx.x;
        const int64_t pos = i < n_cand ? cand_pos[i] : -1;
        bool victim = pos >= 0, written = false;
        int t = 0;
        if (victim) {
            const uint64_t key = d.cand_key_sorted[i];
            const int32_t s = d.cand_val_sorted[i];
            const int64_t gp = (int64_t)(uint32_t)key;
            t = (int)(key >> 58);
            // a page used by the current call is being dropped: the caller's row indices into the
            // slot pool go stale (the reference would crash in save_emb on the freed page)
            if (d.slot_epoch[s] == now) atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_EVICTED_CURRENT);
            written = !d.dirty_only || d.slot_dirty[s];
            d.slot_dirty[s] = 0;
            d.ring[(tail + pos) % d.S] = s;
            d.page_map[gp] = -1;
            d.slot_gp[s] = -1;
        }
        wave_add2(d.counters, victim, t, C_WRITTEN_BACK, written, C_WRITTEN_BACK, false);
    }
}

__global__ void finish_reclaim_kernel(TableGeom g, CacheDev d, int only_table) {
    const int64_t passes = d.scalars[S_PASSES];
    int64_t freed = 0;
    for (int t = 0; t < g.T; t++) {
        const int64_t k = d.k_t[t];
        freed += k;
        d.counters[t * kNumCounters + C_EVICT] += k;
        d.counters[t * kNumCounters + C_RESIDENT] -= k;
        if (only_table < 0 || t == only_table) {
            d.counters[t * kNumCounters + C_WRITE] += passes * g.ws;   // write_page_count += ws_opt per call (:315)
            d.scalars[S_PAGE_COUNT] -= passes * g.ws;                   // page_count -= ws_opt per call (:312)
        }
    }
    d.scalars[S_TAIL] += freed;
    d.scalars[S_PASSES] = 0;
}

// rows of the slot pool that a training step updates: their pages differ from the store from now on
template <typename IDX>
__global__ __launch_bounds__(kCacheThreads) void mark_dirty_kernel(CacheDev d, int vpo, const IDX *rows, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t r = (int64_t)rows[i];
        if (r >= 0 && r / vpo < d.S) d.slot_dirty[r / vpo] = 1;
    }
}

__global__ __launch_bounds__(kCacheThreads) void flush_kernel(TableGeom g, CacheDev d) {
    const int64_t chunks = g.page_bytes / 16;
    const int64_t total = d.S * chunks;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t s = w / chunks, c = w - s * chunks;
        const int32_t gp = d.slot_gp[s];
        if (gp < 0) continue;
        if (d.dirty_only && !d.slot_dirty[s]) continue;
        const int t = table_of_page(g, gp);
        const int64_t page = gp - g.page_base[t];
        if (c * 16 < valid_page_bytes(g, t, page)) {
            const uint4 v = *reinterpret_cast<const uint4 *>(d.data + s * g.page_bytes + c * 16);
            *reinterpret_cast<uint4 *>(g.store[t] + page * g.page_bytes + c * 16) = v;
        }
    }
    if (!g.has_state) return;
    for (int64_t w = (int64_t)blockIdx.x * kCacheThreads + threadIdx.x; w < d.S * g.vpo;
         w += (int64_t)gridDim.x * kCacheThreads) {
        const int64_t s = w / g.vpo, r = w - s * g.vpo;
        const int32_t gp = d.slot_gp[s];
        if (gp < 0 || (d.dirty_only && !d.slot_dirty[s])) continue;
        const int t = table_of_page(g, gp);
        const int64_t row = ((int64_t)gp - g.page_base[t]) * g.vpo + r;
        if (row < g.rows[t]) g.state_store[t][row] = d.state[w];
    }
}

__global__ void clear_dirty_kernel(CacheDev d) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < d.S; s += (int64_t)gridDim.x * blockDim.x)
        d.slot_dirty[s] = 0;
}

__global__ void init_ring_kernel(CacheDev d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.S; i += (int64_t)gridDim.x * blockDim.x) {
        d.ring[i] = (int32_t)i;
        d.slot_gp[i] = -1;
        d.slot_epoch[i] = 0;
        d.slot_dirty[i] = 0;
    }
}

// workgroups of the kernels that move pages over the host link (tuning aids: DLRM_CACHE_FILL_BLOCKS, DLRM_CACHE_WB_BLOCKS)
static unsigned link_blocks(bool writeback) {
    static const int fill = [] { const char *e = getenv("DLRM_CACHE_FILL_BLOCKS"); return e ? atoi(e) : kLinkBlocks; }();
    static const int wb = [] { const char *e = getenv("DLRM_CACHE_WB_BLOCKS"); return e ? atoi(e) : kLinkBlocks; }();
    const int v = writeback ? wb : fill;
    return (unsigned)(v < 1 ? 1 : v > 4096 ? 4096 : v);
}

static unsigned grid_for(int64_t n) {
    const int64_t b = ceil_div(n > 0 ? n : 1, kCacheThreads);
    return (unsigned)(b < 4096 ? b : 4096);
}

template <typename T>
static int dev_alloc(T **p, size_t count) {
    DLRM_HIP(hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * (count ? count : 1)));
    return DLRM_OK;
}

static int read_scalars(dlrm_cache *c, hipStream_t s) {
    DLRM_HIP(hipMemcpyAsync(c->h_scalars, c->d.scalars, sizeof(int64_t) * S_COUNT, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}

}  // namespace dlrm

using namespace dlrm;

extern "C" int dlrm_host_alloc(void **h_ptr, size_t bytes) {
    DLRM_REQUIRE(h_ptr != nullptr, "h_ptr is NULL");
    DLRM_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable));
    return DLRM_OK;
}

extern "C" int dlrm_host_free(void *h_ptr) {
    if (h_ptr) DLRM_HIP(hipHostFree(h_ptr));
    return DLRM_OK;
}

static int ensure_lookup_buffers(dlrm_cache *c, int64_t n) {
    if (n <= c->miss_cap) return DLRM_OK;
    (void)hipFree(c->d.miss); (void)hipFree(c->d.miss_sorted);
    c->d.miss = c->d.miss_sorted = nullptr;
    c->miss_cap = 0;
    if (int rc = dev_alloc(&c->d.miss, (size_t)n)) return rc;
    if (int rc = dev_alloc(&c->d.miss_sorted, (size_t)n)) return rc;
    c->miss_cap = n;
    return DLRM_OK;
}

static int ensure_sort_tmp(dlrm_cache *c, size_t bytes) {
    if (bytes <= c->sort_tmp_bytes) return DLRM_OK;
    (void)hipFree(c->sort_tmp);
    c->sort_tmp = nullptr; c->sort_tmp_bytes = 0;
    DLRM_HIP(hipMalloc(&c->sort_tmp, bytes));
    c->sort_tmp_bytes = bytes;
    return DLRM_OK;
}

extern "C" int dlrm_cache_create(const dlrm_cache_config *cfg, dlrm_cache **out) {
    DLRM_REQUIRE(cfg != nullptr && out != nullptr, "NULL argument");
    DLRM_REQUIRE(cfg->num_tables >= 1 && cfg->num_tables <= kMaxCacheTables, "num_tables must be 1..%d", kMaxCacheTables);
    DLRM_REQUIRE(cfg->dim >= 1 && cfg->rows_per_page >= 1 && cfg->evict_group >= 1, "bad geometry");
    DLRM_REQUIRE(cfg->row_dtype == DLRM_F32 || cfg->row_dtype == DLRM_BF16, "row_dtype %d", cfg->row_dtype);
    DLRM_REQUIRE(cfg->table_rows != nullptr && cfg->h_table_store != nullptr, "table_rows / h_table_store is NULL");
    DLRM_REQUIRE(cfg->num_slots >= 1 && cfg->num_slots < (1ll << 31), "num_slots out of range");
    const int esz = cfg->row_dtype == DLRM_F32 ? 4 : 2;
    const int64_t page_bytes = (int64_t)cfg->rows_per_page * cfg->dim * esz;
    DLRM_REQUIRE(page_bytes % 16 == 0, "a page (%lld bytes) must be a multiple of 16 bytes", (long long)page_bytes);

    dlrm_cache *c = new dlrm_cache();
    memset(c, 0, sizeof(*c));
    TableGeom &g = c->g;
    g.T = cfg->num_tables; g.vpo = cfg->rows_per_page; g.ws = cfg->evict_group; g.dim = cfg->dim; g.esz = esz;
    g.page_bytes = page_bytes;
    g.page_base[0] = 0;
    for (int t = 0; t < g.T; t++) {
        if (cfg->table_rows[t] < 1 || cfg->h_table_store[t] == nullptr) {
            delete c;
            return set_error(DLRM_ERR_INVALID, "table %d: rows < 1 or NULL backing store", t);
        }
        g.rows[t] = cfg->table_rows[t];
        g.page_base[t + 1] = g.page_base[t] + ceil_div(g.rows[t], g.vpo);
        // pinned host memory is addressed by the device through its mapped alias
        void *dptr = nullptr;
        hipError_t e = hipHostGetDevicePointer(&dptr, cfg->h_table_store[t], 0);
        if (e != hipSuccess) {
            delete c;
            return set_error(DLRM_ERR_HIP, "table %d backing store is not pinned, device-accessible host memory: %s", t,
                             hipGetErrorString(e));
        }
        g.store[t] = static_cast<char *>(dptr);
        if (cfg->h_row_state_store) {
            void *sptr = nullptr;
            if (cfg->h_row_state_store[t] == nullptr ||
                hipHostGetDevicePointer(&sptr, cfg->h_row_state_store[t], 0) != hipSuccess) {
                delete c;
                return set_error(DLRM_ERR_INVALID, "table %d: row-state store is NULL or not pinned, device-accessible host memory", t);
            }
            g.state_store[t] = static_cast<float *>(sptr);
        }
    }
    g.has_state = cfg->h_row_state_store != nullptr;
    c->total_pages = g.page_base[g.T];
    if (c->total_pages >= 0x7f7f7f7fll) {
        delete c;
        return set_error(DLRM_ERR_INVALID, "too many pages (%lld) for 32-bit page ids", (long long)c->total_pages);
    }
    CacheDev &d = c->d;
    d.S = cfg->num_slots;
    d.dirty_only = cfg->writeback_dirty_only ? 1 : 0;
    c->reclaim_table = -1;
    int rc = DLRM_OK;
    if (!rc) rc = dev_alloc(&d.page_map, (size_t)c->total_pages);
    if (!rc) rc = dev_alloc(&d.slot_gp, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.slot_epoch, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.slot_dirty, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.ring, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.scalars, (size_t)S_COUNT);
    if (!rc) rc = dev_alloc(&d.counters, (size_t)g.T * kNumCounters);
    if (!rc) rc = dev_alloc(&d.k_t, (size_t)g.T);
    if (!rc) rc = dev_alloc(&d.thr_age, (size_t)g.T);
    if (!rc) rc = dev_alloc(&d.hist, (size_t)g.T * kAgeBins);
    if (!rc) rc = dev_alloc(&d.cand_key, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.cand_key_sorted, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.cand_val, (size_t)d.S);
    if (!rc) rc = dev_alloc(&d.cand_val_sorted, (size_t)d.S);
    if (cfg->d_slot_pool) {
        d.data = static_cast<char *>(cfg->d_slot_pool);
        if (reinterpret_cast<uintptr_t>(d.data) % 16 != 0) rc = set_error(DLRM_ERR_INVALID, "d_slot_pool must be 16-byte aligned");
    } else {
        c->owns_pool = true;
        if (!rc) rc = dev_alloc(&d.data, (size_t)d.S * (size_t)page_bytes);
    }
    if (g.has_state && cfg->d_row_state_pool) {
        d.state = cfg->d_row_state_pool;
    } else if (g.has_state) {
        c->owns_state = true;
        if (!rc) rc = dev_alloc(&d.state, (size_t)d.S * (size_t)g.vpo);
        if (!rc) (void)hipMemset(d.state, 0, sizeof(float) * (size_t)d.S * (size_t)g.vpo);
    }
    if (!rc && (hipHostMalloc(reinterpret_cast<void **>(&c->h_reclaim), sizeof(int64_t) * S_COUNT, 0) != hipSuccess ||
                hipEventCreateWithFlags(&c->reclaim_ev, hipEventDisableTiming) != hipSuccess))
        rc = set_error(DLRM_ERR_NOMEM, "reclaim staging allocation failed");
    if (!rc && hipHostMalloc(reinterpret_cast<void **>(&c->h_scalars), sizeof(int64_t) * S_COUNT, 0) != hipSuccess)
        rc = set_error(DLRM_ERR_NOMEM, "hipHostMalloc failed");
    if (!rc && cfg->max_lookup_ids > 0) {
        // everything a lookup of up to max_lookup_ids ids and a reclaim can need, so that no later call allocates
        rc = ensure_lookup_buffers(c, cfg->max_lookup_ids);
        size_t t1 = 0, t2 = 0;
        if (!rc && rocprim::radix_sort_keys(nullptr, t1, d.miss, d.miss_sorted, (size_t)cfg->max_lookup_ids, 0u, 31u,
                                            (hipStream_t) nullptr, false) != hipSuccess)
            rc = set_error(DLRM_ERR_HIP, "radix sort size query failed");
        if (!rc && rocprim::radix_sort_pairs(nullptr, t2, d.cand_key, d.cand_key_sorted, d.cand_val, d.cand_val_sorted,
                                             (size_t)d.S, 0u, 64u, (hipStream_t) nullptr, false) != hipSuccess)
            rc = set_error(DLRM_ERR_HIP, "radix sort size query failed");
        if (!rc) rc = ensure_sort_tmp(c, t1 > t2 ? t1 : t2);
    }
    if (rc) { dlrm_cache_destroy(c); return rc; }
    (void)hipMemset(d.page_map, 0xff, sizeof(int32_t) * (size_t)c->total_pages);
    (void)hipMemset(d.scalars, 0, sizeof(int64_t) * S_COUNT);
    (void)hipMemset(d.counters, 0, sizeof(int64_t) * g.T * kNumCounters);
    (void)hipMemset(d.hist, 0, sizeof(uint32_t) * g.T * kAgeBins);
    hipLaunchKernelGGL(init_ring_kernel, dim3(grid_for(d.S)), dim3(256), 0, 0, d);
    int64_t init[S_COUNT] = {0};
    init[S_TAIL] = d.S;
    init[S_CACHE_SIZE] = cfg->cache_size;
    (void)hipMemcpy(d.scalars, init, sizeof(init), hipMemcpyHostToDevice);
    if (hipDeviceSynchronize() != hipSuccess) {
        dlrm_cache_destroy(c);
        return set_error(DLRM_ERR_HIP, "cache initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = c;
    return DLRM_OK;
}

extern "C" int dlrm_cache_destroy(dlrm_cache *c) {
    if (!c) return DLRM_OK;
    CacheDev &d = c->d;
    (void)hipFree(d.page_map); (void)hipFree(d.slot_gp); (void)hipFree(d.slot_epoch); (void)hipFree(d.slot_dirty); (void)hipFree(d.ring); (void)hipFree(d.scalars);
    (void)hipFree(d.counters); (void)hipFree(d.k_t); (void)hipFree(d.thr_age); (void)hipFree(d.hist); (void)hipFree(d.miss); (void)hipFree(d.miss_sorted);
    (void)hipFree(d.cand_key); (void)hipFree(d.cand_key_sorted); (void)hipFree(d.cand_val); (void)hipFree(d.cand_val_sorted);
    if (c->wb_stream) {
        (void)hipStreamSynchronize(c->wb_stream);
        (void)hipStreamDestroy(c->wb_stream);
        (void)hipEventDestroy(c->wb_ready);
        (void)hipEventDestroy(c->wb_done);
    }
    (void)hipFree(d.pending); (void)hipFree(d.stage_data); (void)hipFree(d.stage_state); (void)hipFree(d.stage_gp);
    if (c->owns_pool) (void)hipFree(d.data);
    if (c->owns_state) (void)hipFree(d.state);
    (void)hipFree(c->sort_tmp);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->h_reclaim) (void)hipHostFree(c->h_reclaim);
    if (c->reclaim_ev) (void)hipEventDestroy(c->reclaim_ev);
    delete c;
    return DLRM_OK;
}

extern "C" int dlrm_cache_storage(dlrm_cache *c, void **d_data, int64_t *total_rows) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    if (d_data) *d_data = c->d.data;
    if (total_rows) *total_rows = c->d.S * c->g.vpo;
    return DLRM_OK;
}

extern "C" int dlrm_cache_row_state(dlrm_cache *c, float **d_state) {
    DLRM_REQUIRE(c != nullptr && d_state != nullptr, "NULL argument");
    *d_state = c->d.state;
    return DLRM_OK;
}

extern "C" int dlrm_cache_lookup(dlrm_cache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                                 void *d_rows_out, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_table_offsets != nullptr, "NULL argument");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    IdLayout l;
    for (int t = 0; t <= c->g.T; t++) {
        l.off[t] = h_table_offsets[t];
        DLRM_REQUIRE(t == 0 ? l.off[0] == 0 : l.off[t] >= l.off[t - 1], "table offsets must start at 0 and ascend");
    }
    const int64_t n = l.off[c->g.T];
    hipStream_t s = as_stream(stream);
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_ids != nullptr && d_rows_out != nullptr, "d_ids / d_rows_out is NULL");
    // (buffers grow outside of steady state only; a caller that must not allocate sizes
    //  its first call to the largest batch)
    if (int rc = ensure_lookup_buffers(c, n)) return rc;
    const int bits = 31;  // whole non-negative int32 range: the 0x7f7f7f7f fill of unused entries sorts last
    size_t tmp = 0;
    DLRM_HIP(rocprim::radix_sort_keys(nullptr, tmp, c->d.miss, c->d.miss_sorted, (size_t)n, 0u, (unsigned)bits, s, false));
    if (int rc = ensure_sort_tmp(c, tmp)) return rc;

    // a new epoch for this call; unused miss entries sort to the end
    DLRM_HIP(hipMemsetAsync(c->d.miss, 0x7f, sizeof(int32_t) * (size_t)n, s));
    hipLaunchKernelGGL(begin_lookup_kernel, dim3(1), dim3(1), 0, s, c->d);
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(probe_kernel<int64_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->g, c->d, l,
                           static_cast<const int64_t *>(d_ids), n);
    else
        hipLaunchKernelGGL(probe_kernel<int32_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->g, c->d, l,
                           static_cast<const int32_t *>(d_ids), n);
    DLRM_LAUNCH_CHECK();
    tmp = c->sort_tmp_bytes;
    DLRM_HIP(rocprim::radix_sort_keys(c->sort_tmp, tmp, c->d.miss, c->d.miss_sorted, (size_t)n, 0u, (unsigned)bits, s,
                                      false));
    hipLaunchKernelGGL(assign_kernel, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->g, c->d);
    if (c->stage_cap > 0 && c->wb_recorded) {
        // pages that are not on their way to the store cross the link now, beside that write-back; the few that are
        // (evicted by the last reclaim and wanted again already) once it has ended
        hipLaunchKernelGGL(fill_kernel, dim3(link_blocks(false)), dim3(kCacheThreads), 0, s, c->g, c->d, 0);
        DLRM_HIP(hipStreamWaitEvent(s, c->wb_done, 0));
        hipLaunchKernelGGL(fill_kernel, dim3(link_blocks(false)), dim3(kCacheThreads), 0, s, c->g, c->d, 1);
    } else {
        hipLaunchKernelGGL(fill_kernel, dim3(link_blocks(false)), dim3(kCacheThreads), 0, s, c->g, c->d, -1);
    }
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(translate_kernel<int64_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->g, c->d, l,
                           static_cast<const int64_t *>(d_ids), n, static_cast<int64_t *>(d_rows_out));
    else
        hipLaunchKernelGGL(translate_kernel<int32_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->g, c->d, l,
                           static_cast<const int32_t *>(d_ids), n, static_cast<int32_t *>(d_rows_out));
    hipLaunchKernelGGL(finish_lookup_kernel, dim3(1), dim3(1), 0, s, c->d);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_cache_reclaim_begin(dlrm_cache *c, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(!c->reclaim_pending, "dlrm_cache_reclaim_begin called twice without dlrm_cache_reclaim_finish");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(plan_reclaim_kernel, dim3(1), dim3(1), 0, s, c->g, c->d, c->reclaim_table);
    hipLaunchKernelGGL(age_hist_kernel, dim3(grid_for(c->d.S)), dim3(kCacheThreads), 0, s, c->g, c->d);
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)c->g.T), dim3(kAgeBins / 4), 0, s, c->g, c->d, c->protect_age);
    hipLaunchKernelGGL(collect_candidates_kernel, dim3(grid_for(c->d.S)), dim3(kCacheThreads), 0, s, c->g, c->d);
    DLRM_LAUNCH_CHECK();
    // the candidate count decides the size of the sort: one small read-back, waited for in _finish
    DLRM_HIP(hipMemcpyAsync(c->h_reclaim, c->d.scalars, sizeof(int64_t) * S_COUNT, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipEventRecord(c->reclaim_ev, s));
    c->reclaim_pending = true;
    return DLRM_OK;
}

extern "C" int dlrm_cache_reclaim_finish(dlrm_cache *c, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(c->reclaim_pending, "dlrm_cache_reclaim_finish without dlrm_cache_reclaim_begin");
    hipStream_t s = as_stream(stream);
    DLRM_HIP(hipEventSynchronize(c->reclaim_ev));
    c->reclaim_pending = false;
    const int64_t passes = c->h_reclaim[S_PASSES], n_cand = c->h_reclaim[S_N_CAND];
    if (passes == 0) return DLRM_OK;
    // victims of this reclaim: at most passes * evict_group per table
    const int64_t most = passes * c->g.ws * (c->reclaim_table >= 0 ? 1 : c->g.T);
    const bool park = c->stage_cap > 0 && n_cand > 0 && most <= c->stage_cap;
    if (c->wb_recorded) DLRM_HIP(hipStreamWaitEvent(s, c->wb_done, 0));   // the parking area and the flags are free again
    if (c->stage_cap > 0) DLRM_HIP(hipMemsetAsync(c->d.pending, 0, sizeof(uint32_t) * (size_t)ceil_div(c->total_pages, 32), s));
    if (park) DLRM_HIP(hipMemsetAsync(c->d.stage_gp, 0xff, sizeof(int32_t) * (size_t)most, s));
    if (n_cand > 0) {
        size_t tmp = 0;
        DLRM_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d.cand_key, c->d.cand_key_sorted, c->d.cand_val,
                                           c->d.cand_val_sorted, (size_t)n_cand, 0u, 64u, s, false));
        if (int rc = ensure_sort_tmp(c, tmp)) return rc;
        tmp = c->sort_tmp_bytes;
        DLRM_HIP(rocprim::radix_sort_pairs(c->sort_tmp, tmp, c->d.cand_key, c->d.cand_key_sorted, c->d.cand_val,
                                           c->d.cand_val_sorted, (size_t)n_cand, 0u, 64u, s, false));
        int32_t *cand_pos = c->d.cand_val;  // the unsorted value buffer is free once the sort is done
        hipLaunchKernelGGL(rank_victims_kernel, dim3(grid_for(n_cand)), dim3(kCacheThreads), 0, s, c->g, c->d, n_cand,
                           cand_pos);
        if (park)
            hipLaunchKernelGGL(park_victims_kernel, dim3(grid_for(n_cand * (c->g.page_bytes / 16))), dim3(kCacheThreads), 0, s,
                               c->g, c->d, n_cand, cand_pos);
        else
            hipLaunchKernelGGL(writeback_victims_kernel, dim3(link_blocks(true)), dim3(kCacheThreads), 0, s, c->g, c->d, n_cand,
                               cand_pos);
        hipLaunchKernelGGL(evict_kernel, dim3(grid_for(n_cand)), dim3(kCacheThreads), 0, s, c->g, c->d, n_cand,
                           cand_pos);
    }
    hipLaunchKernelGGL(finish_reclaim_kernel, dim3(1), dim3(1), 0, s, c->g, c->d, c->reclaim_table);
    DLRM_LAUNCH_CHECK();
    if (park) {
        DLRM_HIP(hipEventRecord(c->wb_ready, s));
        DLRM_HIP(hipStreamWaitEvent(c->wb_stream, c->wb_ready, 0));
        hipLaunchKernelGGL(writeback_parked_kernel, dim3(link_blocks(true)), dim3(kCacheThreads), 0, c->wb_stream, c->g, c->d, most);
        DLRM_LAUNCH_CHECK();
        DLRM_HIP(hipEventRecord(c->wb_done, c->wb_stream));
        c->wb_recorded = true;
    }
    return DLRM_OK;
}

extern "C" int dlrm_cache_set_async_writeback(dlrm_cache *c, int64_t max_pages, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(max_pages >= 0, "max_pages < 0");
    DLRM_REQUIRE(!c->reclaim_pending, "a reclaim is pending: call dlrm_cache_reclaim_finish first");
    hipStream_t s = as_stream(stream);
    if (c->wb_stream) DLRM_HIP(hipStreamSynchronize(c->wb_stream));   // nothing is on its way any more
    DLRM_HIP(hipStreamSynchronize(s));
    c->wb_recorded = false;
    (void)hipFree(c->d.pending); (void)hipFree(c->d.stage_data); (void)hipFree(c->d.stage_state); (void)hipFree(c->d.stage_gp);
    c->d.pending = nullptr; c->d.stage_data = nullptr; c->d.stage_state = nullptr; c->d.stage_gp = nullptr;
    c->stage_cap = 0;
    if (max_pages == 0) return DLRM_OK;
    if (!c->wb_stream) {
        DLRM_HIP(hipStreamCreateWithFlags(&c->wb_stream, hipStreamNonBlocking));
        DLRM_HIP(hipEventCreateWithFlags(&c->wb_ready, hipEventDisableTiming));
        DLRM_HIP(hipEventCreateWithFlags(&c->wb_done, hipEventDisableTiming));
    }
    const size_t words = (size_t)ceil_div(c->total_pages, 32);
    int rc = dev_alloc(&c->d.pending, words);
    if (!rc) rc = dev_alloc(&c->d.stage_data, (size_t)max_pages * (size_t)c->g.page_bytes);
    if (!rc) rc = dev_alloc(&c->d.stage_gp, (size_t)max_pages);
    if (!rc && c->g.has_state) rc = dev_alloc(&c->d.stage_state, (size_t)max_pages * (size_t)c->g.vpo);
    if (rc) {
        (void)hipFree(c->d.pending); (void)hipFree(c->d.stage_data); (void)hipFree(c->d.stage_state); (void)hipFree(c->d.stage_gp);
        c->d.pending = nullptr; c->d.stage_data = nullptr; c->d.stage_state = nullptr; c->d.stage_gp = nullptr;
        return rc;
    }
    DLRM_HIP(hipMemset(c->d.pending, 0, sizeof(uint32_t) * words));
    c->stage_cap = max_pages;
    return DLRM_OK;
}

extern "C" int dlrm_cache_reclaim(dlrm_cache *c, dlrm_stream_t stream) {
    if (int rc = dlrm_cache_reclaim_begin(c, stream)) return rc;
    return dlrm_cache_reclaim_finish(c, stream);
}

extern "C" int dlrm_cache_reclaim_table(dlrm_cache *c, int32_t table, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(table >= 0 && table < c->g.T, "table %d out of range", table);
    DLRM_REQUIRE(!c->reclaim_pending, "a reclaim is pending: call dlrm_cache_reclaim_finish first");
    c->reclaim_table = table;
    int rc = dlrm_cache_reclaim_begin(c, stream);
    if (!rc) rc = dlrm_cache_reclaim_finish(c, stream);
    c->reclaim_table = -1;
    return rc;
}

extern "C" int dlrm_cache_set_protect_age(dlrm_cache *c, int32_t age) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(age >= 0 && age < kAgeBins, "protect age must be 0..%d", kAgeBins - 1);
    c->protect_age = age;
    return DLRM_OK;
}

extern "C" int dlrm_cache_mark_dirty(dlrm_cache *c, const void *d_rows, int32_t idx_dtype, int64_t n, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    DLRM_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_rows != nullptr, "d_rows is NULL");
    hipStream_t s = as_stream(stream);
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(mark_dirty_kernel<int64_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->d, c->g.vpo,
                           static_cast<const int64_t *>(d_rows), n);
    else
        hipLaunchKernelGGL(mark_dirty_kernel<int32_t>, dim3(grid_for(n)), dim3(kCacheThreads), 0, s, c->d, c->g.vpo,
                           static_cast<const int32_t *>(d_rows), n);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_cache_flush(dlrm_cache *c, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    if (c->wb_recorded) DLRM_HIP(hipStreamWaitEvent(as_stream(stream), c->wb_done, 0));   // the store is whole behind this call
    hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(kCacheThreads), 0, as_stream(stream), c->g, c->d);
    hipLaunchKernelGGL(clear_dirty_kernel, dim3(grid_for(c->d.S)), dim3(256), 0, as_stream(stream), c->d);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_cache_counters(dlrm_cache *c, int64_t *h_out, int32_t reset_read_write, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_out != nullptr, "NULL argument");
    hipStream_t s = as_stream(stream);
    const size_t bytes = sizeof(int64_t) * c->g.T * kNumCounters;
    DLRM_HIP(hipMemcpyAsync(h_out, c->d.counters, bytes, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    if (reset_read_write) {
        // cache::read_page(i) / write_page(i) zero the two page counters after logging (include/data.h:274-285)
        hipLaunchKernelGGL(reset_rw_counters_kernel, dim3(1), dim3(64), 0, s, c->d.counters, c->g.T);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

extern "C" int dlrm_cache_state(dlrm_cache *c, dlrm_cache_state_t *h_state, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_state != nullptr, "NULL argument");
    if (int rc = read_scalars(c, as_stream(stream))) return rc;
    h_state->page_count = c->h_scalars[S_PAGE_COUNT];
    h_state->cache_size = c->h_scalars[S_CACHE_SIZE];
    h_state->ring_head = c->h_scalars[S_HEAD];
    h_state->ring_tail = c->h_scalars[S_TAIL];
    h_state->epoch = c->h_scalars[S_EPOCH];
    h_state->error_flags = c->h_scalars[S_ERROR];
    h_state->last_lookup_pages_read = c->h_scalars[S_N_ASSIGNED];
    h_state->total_pages = c->total_pages;
    h_state->num_slots = c->d.S;
    return DLRM_OK;
}

extern "C" int dlrm_cache_set_cache_size(dlrm_cache *c, int64_t cache_size, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    hipStream_t s = as_stream(stream);
    DLRM_HIP(hipMemcpyAsync(c->d.scalars + S_CACHE_SIZE, &cache_size, sizeof(int64_t), hipMemcpyHostToDevice, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}

extern "C" int dlrm_cache_export_page_map(dlrm_cache *c, int32_t table, int32_t *h_out, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_out != nullptr && table >= 0 && table < c->g.T, "bad argument");
    hipStream_t s = as_stream(stream);
    const int64_t n = c->g.page_base[table + 1] - c->g.page_base[table];
    DLRM_HIP(hipMemcpyAsync(h_out, c->d.page_map + c->g.page_base[table], sizeof(int32_t) * (size_t)n,
                            hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}
