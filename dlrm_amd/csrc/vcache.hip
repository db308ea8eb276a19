// Vector-granular HBM cache over a log-structured page store in pinned host DRAM.
//
// Replaces class cache2 (dlrm_libtorch/include/data2.h:219-287, src/emb_cache2.cpp) and the policy
// driver of src/dlrmtest2.cpp (reclaim_emb_cache :462-481, save_emb :148-183), with host DRAM standing in
// for the OCSSD:
//   * residency is kept per embedding vector ("row");
//   * every table has a list of the rows that were asked for; a hit counts (the frequency used for the
//     hot/cold classification) but never moves the entry, so the list is FIFO by first request
//     (lru_page, emb_cache2.cpp:638-659).  Here: per slot a key (lookup epoch, position of the first
//     occurrence in the call) and a hit count -- oldest key = back of the list;
//   * a table whose list is longer than cache_size evicts its oldest entries.  Evicted rows collect in a
//     per-table buffer of W = evict_group * rows_per_page rows; a full buffer becomes evict_group new pages
//     of the store (Reclaim_vector_page, :397-633): rows that left together sit in one page;
//   * a missing row is read with the whole page it lives in, and every row that page lists becomes
//     resident (read, :729-892): one 4 KB read over the host link brings back what was evicted together.
//     Rows that came along unasked are resident but not listed; they are dropped when a listed row of
//     their home page is evicted (:492-496).
// Pages are identified per table: initial pages 0..n_init-1 hold rows [p*vpo, (p+1)*vpo) at their natural
// place in the store, written pages follow at n_init + k.  Where the reference would hand back a stale
// copy (a page read that lists a row which has moved on, or a row still waiting in the buffer), this
// cache fetches the row from where its latest value is: the state machine is the reference's, the values
// are always current.
//
// All state lives in HBM (direct arrays indexed by row: 288 GB make 12 bytes per row cheaper than any hash).
// The state machine is a pure function of the call sequence; slot numbers are not (free slots are taken
// with atomics), so tests compare residency, list membership, counts, order and page placement.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace dlrm {
namespace vc {

constexpr int kMaxTables = 64;
constexpr int kThreads = 256;
constexpr int kGroup = 16;                     // lanes that move one row, 16 bytes each per trip
constexpr int kGroupsPerBlock = kThreads / kGroup;
constexpr int kLinkBlocks = 512;               // kernels that wait on the host link keep most wave slots free
constexpr int kCounters = DLRM_VCACHE_NUM_COUNTERS;
enum { C_READ = 0, C_WRITE, C_HIT, C_MISS, C_EVICT, C_RESIDENT, C_LISTED, C_DROPPED, C_PENDING, C_WRITTEN };
enum { S_HEAD = 0, S_TAIL, S_PAGE_COUNT, S_CACHE_SIZE, S_N_MISS, S_ERROR, S_EPOCH, S_N_LISTED, S_N_EVICT, S_N_CAND,
       S_COUNT };
enum { ERR_BAD_ID = 1, ERR_POOL_EXHAUSTED = 2, ERR_LOG_FULL = 8 };
constexpr uint32_t kAgeMax = 0x3ffffffu;       // 26 bits of age in the eviction sort key

struct Geom {
    int T, vpo, ws, dim, W;                    // W = rows per write buffer
    int64_t row_bytes, page_bytes, log_cap;    // log_cap: pages a table may write
    int64_t rows[kMaxTables];
    int64_t row_base[kMaxTables + 1];          // first global row of each table
    int64_t n_init[kMaxTables];                // initial pages of each table
    int64_t page_base[kMaxTables + 1];         // first global page (initial + written region) of each table
    char *store[kMaxTables];                   // pinned host store per table: (n_init + log_cap) pages
    float *state_store[kMaxTables];            // one float per row position of that store (optimizer state), or NULL
    int has_state;
};

struct IdLayout {
    int64_t off[kMaxTables + 1];
};

struct Dev {
    int32_t *row_slot;     // [rows] slot, -1 not resident, -2 being installed
    int32_t *row_loc;      // [rows] page of the table that holds the row's stored value
    int32_t *row_pend;     // [rows] position in the table's write buffer while the row waits there, else -1
    int32_t *slot_row;     // [S] global row
    uint64_t *slot_key;    // [S] 0 = not listed, else (epoch << 32) | first position
    int32_t *slot_cnt;     // [S] hits since the row was listed
    int32_t *slot_first;   // [S] scratch of a lookup: first position of the row in the call
    int32_t *slot_occ;     // [S] scratch of a lookup: occurrences of the row in the call
    int32_t *ring;         // [S] free slots
    int64_t *scalars;
    int64_t *counters;     // [T][kCounters]
    uint32_t *page_stamp;  // [pages] epoch of the lookup that last read the page
    int32_t *log_rows;     // [T][log_cap][vpo] rows of every written page (table-local ids)
    int32_t *pend_row;     // [T][W]
    int32_t *pend_cnt;     // [T][W]
    char *pend_data;       // [T][W][row_bytes]
    int64_t *pend_n;       // [T]
    int64_t *log_head;     // [T] pages written so far
    int64_t *k_t;          // [T] evictions of the current reclaim
    int64_t *cand_base;    // [T] first position of the table in the sorted candidates
    int64_t *ev_base;      // [T+1] evictions of the tables before t in the current reclaim
    int64_t *flushes;      // [T] buffers completed by the current reclaim
    int32_t *miss;         // pages to read (global ids)
    uint64_t *cand_key, *cand_key_sorted;
    int32_t *cand_val, *cand_val_sorted;
    char *data;            // slot pool [S][row_bytes]
    float *state;          // [S] optimizer state of the resident rows; NULL = rows carry none
    float *pend_state;     // [T][W]
    int64_t S;
};

}  // namespace vc
}  // namespace dlrm

struct dlrm_vcache {
    dlrm::vc::Geom g;
    dlrm::vc::Dev d;
    int64_t total_rows, total_pages;
    int64_t miss_cap;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int64_t *h_scalars;
    bool owns_pool;
    bool owns_state;
};

namespace dlrm {
namespace vc {

__device__ inline int table_of(const int64_t *base, int T, int64_t x) {   // base[lo] <= x < base[hi]
    int lo = 0, hi = T;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x >= base[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

// one atomic per wave and table instead of one per lane (lanes of a wave mostly work on one table)
__device__ inline void wave_count(int64_t *counters, bool valid, int t, int which, int n) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid && n != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int t0 = __shfl(t, leader, 64);
        const bool mine = valid && n != 0 && t == t0;
        const unsigned long long same = __ballot(mine);
        int v = mine ? n : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == leader) atomicAdd((unsigned long long *)&counters[t0 * kCounters + which], (unsigned long long)(long long)v);
        todo &= ~same;
    }
}

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

// where the latest stored value of row r of table t lives when it is not in a slot
__device__ inline const char *row_source(const Geom &g, const Dev &d, int t, int64_t r, int64_t gr) {
    const int32_t pi = d.row_pend[gr];
    if (pi >= 0) return d.pend_data + ((int64_t)t * g.W + pi) * g.row_bytes;
    const int64_t lp = d.row_loc[gr];
    int j = 0;
    if (lp < g.n_init[t]) {
        j = (int)(r % g.vpo);
    } else {
        const int32_t *lr = d.log_rows + ((int64_t)t * g.log_cap + (lp - g.n_init[t])) * g.vpo;
        for (int q = 0; q < g.vpo; q++)
            if (lr[q] == (int32_t)r) { j = q; break; }
    }
    return g.store[t] + lp * g.page_bytes + (int64_t)j * g.row_bytes;
}

// the float of optimizer state that belongs to the row stored at p (slot pool, write buffer or table t's store)
__device__ inline float *state_at(const Geom &g, const Dev &d, int t, const char *p) {
    if (p >= d.data && p < d.data + d.S * g.row_bytes) return d.state + (p - d.data) / g.row_bytes;
    const char *pend_end = d.pend_data + (int64_t)g.T * g.W * g.row_bytes;
    if (p >= d.pend_data && p < pend_end) return d.pend_state + (p - d.pend_data) / g.row_bytes;
    return g.state_store[t] + (p - g.store[t]) / g.row_bytes;
}

// copies a row of table t between two of the places rows live in; its optimizer state goes along
__device__ inline void copy_row(const Geom &g, const Dev &d, int t, char *dst, const char *src, int lig,
                                bool with_state = true) {
    const int chunks = (int)(g.row_bytes / 16);
    for (int c = lig; c < chunks; c += kGroup)
        *reinterpret_cast<uint4 *>(dst + 16 * c) = *reinterpret_cast<const uint4 *>(src + 16 * c);
    if (g.has_state && with_state && lig == 0) *state_at(g, d, t, dst) = *state_at(g, d, t, src);
}

// Ring traffic of a workgroup trip: the lanes that pop (or push) a slot number themselves in LDS, one lane moves
// the ring's head (or tail) for all of them.  One global atomic per trip instead of one per row: the ring's
// counters are single addresses that every workgroup of the chip hits.
struct RingTrip {
    int *cnt;           // LDS
    long long *base;    // LDS
    __device__ void begin() {
        if (threadIdx.x == 0) *cnt = 0;
        __syncthreads();
    }
    __device__ int take(bool want) { return want ? atomicAdd(cnt, 1) : -1; }
    __device__ int take_n(int n) { return n ? atomicAdd(cnt, n) : 0; }
    // after every lane has called take(): first ring position of this trip's slots (all threads must call)
    __device__ long long reserve(int64_t *counter) {
        __syncthreads();
        if (threadIdx.x == 0) *base = *cnt ? (long long)atomicAdd((unsigned long long *)counter, (unsigned long long)*cnt) : 0;
        __syncthreads();
        return *base;
    }
};

// Every group of kGroup lanes of the workgroup offers one row (t, r) -- or none (valid false) -- that becomes
// resident if it is not (unlisted, hit count 0).  All threads of the workgroup must call.  Returns true for the
// groups that installed their row.  tail: number of pushes to the free ring before this kernel.
__device__ inline bool install_rows(const Geom &g, const Dev &d, RingTrip &ring, bool valid, int t, int64_t r, int lig,
                                    int64_t tail) {
    const int64_t gr = valid ? g.row_base[t] + r : 0;
    ring.begin();
    const bool want = valid && lig == 0 && d.row_slot[gr] == -1 && atomicCAS(&d.row_slot[gr], -1, -2) == -1;
    const int my = ring.take(want);
    const long long first = ring.reserve(&d.scalars[S_HEAD]);
    int32_t s = -1;
    if (want) {
        const int64_t idx = first + my;
        if (idx < tail) {
            s = d.ring[idx % d.S];
        } else {
            atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_POOL_EXHAUSTED);
            d.row_slot[gr] = -1;
        }
    }
    s = __shfl(s, (threadIdx.x & 63) & ~(kGroup - 1), 64);
    if (s < 0) return false;
    copy_row(g, d, t, d.data + (int64_t)s * g.row_bytes, row_source(g, d, t, r, gr), lig);
    if (lig == 0) {
        d.slot_row[s] = (int32_t)gr;
        d.slot_key[s] = 0;
        d.slot_cnt[s] = 0;
        d.slot_first[s] = 0x7fffffff;
        d.slot_occ[s] = 0;
        d.row_slot[gr] = s;
    }
    return true;
}

// per-workgroup tallies of counter changes, added to the global counters once at the end
struct BlockTally {
    int *v;   // LDS [kMaxTables]
    __device__ void clear() {
        for (int i = threadIdx.x; i < kMaxTables; i += kThreads) v[i] = 0;
        __syncthreads();
    }
    __device__ void add(int t, int n) { atomicAdd(&v[t], n); }
    __device__ void flush(int64_t *counters, int T, int which) {
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += kThreads)
            if (v[t]) atomicAdd((unsigned long long *)&counters[t * kCounters + which], (unsigned long long)(long long)v[t]);
    }
};

// ---- lookup ----------------------------------------------------------------

__global__ void begin_lookup_kernel(Dev d) {
    d.scalars[S_EPOCH] += 1;
    d.scalars[S_N_MISS] = 0;
}

// read(): the rows that are not resident and, once each, the pages they live in (emb_cache2.cpp:749-802)
template <typename IDX>
__global__ __launch_bounds__(kThreads) void probe_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n) {
    const uint32_t epoch = (uint32_t)d.scalars[S_EPOCH];
    for (int64_t base = (int64_t)blockIdx.x * kThreads; base < n; base += (int64_t)gridDim.x * kThreads) {
        const int64_t i = base + threadIdx.x;
        bool valid = i < n, hit = false, claim = false;
        int t = 0;
        int64_t gp = 0;
        if (valid) {
            t = table_of(l.off, g.T, i);
            const int64_t id = (int64_t)ids[i];
            if (id < 0 || id >= g.rows[t]) {
                atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_BAD_ID);
                valid = false;
            } else {
                const int64_t gr = g.row_base[t] + id;
                hit = d.row_slot[gr] >= 0;
                if (!hit) {
                    gp = g.page_base[t] + d.row_loc[gr];
                    claim = atomicExch(&d.page_stamp[gp], epoch) != epoch;   // first miss of this call on the page
                }
            }
        }
        wave_count(d.counters, valid, t, C_HIT, hit ? 1 : 0);
        wave_count(d.counters, valid, t, C_MISS, hit ? 0 : 1);
        wave_count(d.counters, valid, t, C_READ, claim ? 1 : 0);
        const int64_t pos = wave_append(&d.scalars[S_N_MISS], claim);
        if (claim) d.miss[pos] = (int32_t)gp;
    }
}

// read(), second half (:865-886): every row a fetched page lists becomes resident
__global__ __launch_bounds__(kThreads) void fill_kernel(Geom g, Dev d) {
    __shared__ int tally_mem[kMaxTables];
    __shared__ int ring_cnt;
    __shared__ long long ring_base;
    BlockTally res{tally_mem};
    RingTrip ring{&ring_cnt, &ring_base};
    res.clear();
    const int64_t n_miss = d.scalars[S_N_MISS], tail = d.scalars[S_TAIL];
    const int64_t items = n_miss * g.vpo;
    const int lig = threadIdx.x % kGroup;
    const int64_t per_trip = (int64_t)gridDim.x * kGroupsPerBlock;
    const int64_t trips = (items + per_trip - 1) / per_trip;          // the same for every workgroup
    for (int64_t trip = 0; trip < trips; trip++) {
        const int64_t it = trip * per_trip + (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup;
        bool valid = it < items;
        int t = 0;
        int64_t r = 0;
        if (valid) {
            const int64_t q = it / g.vpo;
            const int j = (int)(it - q * g.vpo);
            const int64_t gp = d.miss[q];
            t = table_of(g.page_base, g.T, gp);
            const int64_t lp = gp - g.page_base[t];
            if (lp < g.n_init[t]) r = lp * g.vpo + j;
            else r = d.log_rows[((int64_t)t * g.log_cap + (lp - g.n_init[t])) * g.vpo + j];
            valid = r < g.rows[t];                          // tail of a table's last initial page
        }
        if (install_rows(g, d, ring, valid, t, r, lig, tail) && lig == 0) res.add(t, 1);
    }
    res.flush(d.counters, g.T, C_RESIDENT);
}

// lru_page (:638-659), pass 1: per row of the call its first position and its number of occurrences
template <typename IDX>
__global__ __launch_bounds__(kThreads) void occurrences_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int t = table_of(l.off, g.T, i);
        const int64_t id = (int64_t)ids[i];
        if (id < 0 || id >= g.rows[t]) continue;
        const int32_t s = d.row_slot[g.row_base[t] + id];
        if (s < 0) continue;                                // pool exhausted (flagged)
        atomicMin(&d.slot_first[s], (int32_t)i);
        atomicAdd(&d.slot_occ[s], 1);
    }
}

// lru_page, pass 2 (the first occurrence of a row acts for all of them): a listed row counts its hits, a new
// one enters the list with the key of this call; plus the slot of every id (the row pointers of :692-695)
template <typename IDX>
__global__ __launch_bounds__(kThreads) void list_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n,
                                                         IDX *rows_out) {
    const uint64_t epoch = (uint64_t)(uint32_t)d.scalars[S_EPOCH];
    for (int64_t base = (int64_t)blockIdx.x * kThreads; base < n; base += (int64_t)gridDim.x * kThreads) {
        const int64_t i = base + threadIdx.x;
        int t = 0;
        bool listed_now = false;
        if (i < n) {
            t = table_of(l.off, g.T, i);
            const int64_t id = (int64_t)ids[i];
            int32_t s = -1;
            if (id >= 0 && id < g.rows[t]) s = d.row_slot[g.row_base[t] + id];
            rows_out[i] = (IDX)(s >= 0 ? s : -1);
            if (s >= 0 && d.slot_first[s] == (int32_t)i) {
                const int occ = d.slot_occ[s];
                if (d.slot_key[s] != 0) {
                    d.slot_cnt[s] += occ;
                } else {
                    d.slot_key[s] = (epoch << 32) | (uint64_t)(uint32_t)i;
                    d.slot_cnt[s] = occ - 1;
                    listed_now = true;
                }
                d.slot_first[s] = 0x7fffffff;
                d.slot_occ[s] = 0;
            }
        }
        wave_count(d.counters, i < n, t, C_LISTED, listed_now ? 1 : 0);
    }
}

__global__ void finish_lookup_kernel(Dev d) {
    if (d.scalars[S_HEAD] > d.scalars[S_TAIL]) d.scalars[S_HEAD] = d.scalars[S_TAIL];   // after an exhausted pool
}

// save_emb (src/dlrmtest2.cpp:169-177): rows of the batch that reclaim dropped are resident again, unlisted
template <typename IDX>
__global__ __launch_bounds__(kThreads) void restore_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n) {
    __shared__ int tally_mem[kMaxTables];
    __shared__ int ring_cnt;
    __shared__ long long ring_base;
    BlockTally res{tally_mem};
    RingTrip ring{&ring_cnt, &ring_base};
    res.clear();
    const int64_t tail = d.scalars[S_TAIL];
    const int lig = threadIdx.x % kGroup;
    const int64_t per_trip = (int64_t)gridDim.x * kGroupsPerBlock;
    const int64_t trips = (n + per_trip - 1) / per_trip;
    for (int64_t trip = 0; trip < trips; trip++) {
        const int64_t i = trip * per_trip + (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup;
        bool valid = i < n;
        int t = 0;
        int64_t id = 0;
        if (valid) {
            t = table_of(l.off, g.T, i);
            id = (int64_t)ids[i];
            if (id < 0 || id >= g.rows[t]) {
                if (lig == 0) atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_BAD_ID);
                valid = false;
            }
        }
        if (install_rows(g, d, ring, valid, t, id, lig, tail) && lig == 0) res.add(t, 1);
    }
    res.flush(d.counters, g.T, C_RESIDENT);
}

template <typename IDX>
__global__ __launch_bounds__(kThreads) void translate_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n,
                                                              IDX *rows_out) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int t = table_of(l.off, g.T, i);
        const int64_t id = (int64_t)ids[i];
        int32_t s = -1;
        if (id >= 0 && id < g.rows[t]) s = d.row_slot[g.row_base[t] + id];
        rows_out[i] = (IDX)(s >= 0 ? s : -1);
    }
}

// current value of every asked row, wherever it lives (slot, write buffer or store); changes nothing
template <typename IDX>
__global__ __launch_bounds__(kThreads) void read_rows_kernel(Geom g, Dev d, IdLayout l, const IDX *ids, int64_t n,
                                                              char *out, float *state_out) {
    const int lig = threadIdx.x % kGroup;
    for (int64_t i = (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup; i < n;
         i += (int64_t)gridDim.x * kGroupsPerBlock) {
        const int t = table_of(l.off, g.T, i);
        const int64_t id = (int64_t)ids[i];
        if (id < 0 || id >= g.rows[t]) continue;
        const int64_t gr = g.row_base[t] + id;
        const int32_t s = d.row_slot[gr];
        const char *src = s >= 0 ? d.data + (int64_t)s * g.row_bytes : row_source(g, d, t, id, gr);
        copy_row(g, d, t, out + i * g.row_bytes, src, lig, false);
        if (g.has_state && state_out && lig == 0) state_out[i] = *state_at(g, d, t, src);
    }
}

// ---- reclaim ---------------------------------------------------------------

// reclaim_emb_cache (src/dlrmtest2.cpp:462-481): per table, evict while the list is longer than cache_size
__global__ void plan_reclaim_kernel(Geom g, Dev d) {
    const int64_t cap = d.scalars[S_CACHE_SIZE];
    int64_t n_listed = 0, n_evict = 0;
    for (int t = 0; t < g.T; t++) {
        const int64_t listed = d.counters[t * kCounters + C_LISTED];
        int64_t k = listed > cap ? listed - cap : 0;
        int64_t fl = (d.pend_n[t] + k) / g.W;
        if (d.log_head[t] + fl * g.ws > g.log_cap) {       // the store's written region is full: this table keeps its rows
            atomicOr((unsigned long long *)&d.scalars[S_ERROR], (unsigned long long)ERR_LOG_FULL);
            k = 0;
            fl = 0;
        }
        d.k_t[t] = k;
        d.flushes[t] = fl;
        d.cand_base[t] = n_listed;
        d.ev_base[t] = n_evict;
        n_listed += listed;
        n_evict += k;
    }
    d.ev_base[g.T] = n_evict;
    d.scalars[S_N_LISTED] = n_listed;
    d.scalars[S_N_EVICT] = n_evict;
    d.scalars[S_N_CAND] = 0;
}

// every listed slot with a key that sorts by (table, oldest first).  A workgroup takes kCollectPer slots per
// trip and reserves room for its listed ones with ONE atomic (the order of the candidates does not matter, they
// are sorted next): a counter that every wave bumps would serialise a scan of tens of millions of slots.
constexpr int kCollectPer = 16;   // slots per thread and trip

__global__ __launch_bounds__(kThreads) void collect_kernel(Geom g, Dev d) {
    __shared__ int wave_tot[kThreads / 64];
    __shared__ long long blk_base;
    const uint32_t now = (uint32_t)d.scalars[S_EPOCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per_trip = (int64_t)kThreads * kCollectPer;
    for (int64_t base = (int64_t)blockIdx.x * per_trip; base < d.S; base += (int64_t)gridDim.x * per_trip) {
        uint64_t key[kCollectPer];
        int mine = 0;
#pragma unroll
        for (int u = 0; u < kCollectPer; u++) {
            const int64_t s = base + threadIdx.x + (int64_t)u * kThreads;
            key[u] = s < d.S ? d.slot_key[s] : 0;
            mine += key[u] != 0;
        }
        // exclusive prefix of `mine` over the workgroup
        int incl = mine;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int o = __shfl_up(incl, m, 64);
            if (lane >= m) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = incl - mine, total = 0;
        for (int w = 0; w < kThreads / 64; w++) {
            if (w < wave) before += wave_tot[w];
            total += wave_tot[w];
        }
        if (threadIdx.x == 0)
            blk_base = total ? (long long)atomicAdd((unsigned long long *)&d.scalars[S_N_CAND], (unsigned long long)total) : 0;
        __syncthreads();
        int64_t pos = blk_base + before;
#pragma unroll
        for (int u = 0; u < kCollectPer; u++) {
            if (key[u] == 0) continue;
            const int64_t s = base + threadIdx.x + (int64_t)u * kThreads;
            const int t = table_of(g.row_base, g.T, d.slot_row[s]);
            const uint32_t age = now - (uint32_t)(key[u] >> 32);
            const uint64_t inv = kAgeMax - (age < kAgeMax ? age : kAgeMax);
            d.cand_key[pos] = ((uint64_t)t << 58) | (inv << 32) | (key[u] & 0xffffffffull);
            d.cand_val[pos] = (int32_t)s;
            pos++;
        }
        __syncthreads();   // blk_base and wave_tot are reused by the next trip
    }
}

// the rows that waited in a buffer which this reclaim completes go to their new page first (the buffer's
// first pend_n positions; the newcomers of the same buffer are written by evict_kernel)
__global__ __launch_bounds__(kThreads) void flush_staged_kernel(Geom g, Dev d) {
    const int lig = threadIdx.x % kGroup;
    const int64_t items = (int64_t)g.T * g.W;
    for (int64_t it = (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup; it < items;
         it += (int64_t)gridDim.x * kGroupsPerBlock) {
        const int t = (int)(it / g.W);
        const int w = (int)(it - (int64_t)t * g.W);
        if (d.flushes[t] < 1 || w >= d.pend_n[t]) continue;
        const int64_t r = d.pend_row[(int64_t)t * g.W + w];
        const int64_t k = d.log_head[t] + w / g.vpo;
        const int j = w % g.vpo;
        const int64_t lp = g.n_init[t] + k;
        copy_row(g, d, t, g.store[t] + lp * g.page_bytes + (int64_t)j * g.row_bytes,
                 d.pend_data + ((int64_t)t * g.W + w) * g.row_bytes, lig);
        if (lig == 0) {
            d.log_rows[((int64_t)t * g.log_cap + k) * g.vpo + j] = (int32_t)r;
            d.row_loc[g.row_base[t] + r] = (int32_t)lp;
            d.row_pend[g.row_base[t] + r] = -1;
        }
    }
}

// Reclaim_vector_page (emb_cache2.cpp:397-633) for every eviction of this reclaim at once: eviction j of the call is
// eviction e = j - ev_base[t] of its table, i.e. candidate cand_base[t] + e of the sorted list
constexpr int kMaxPush = 1 + (128 + kGroup - 1) / kGroup;   // slots a lane can free in one trip (rows_per_page <= 128)

__global__ __launch_bounds__(kThreads) void evict_kernel(Geom g, Dev d, int64_t n_evict) {
    __shared__ int tally_mem[2][kMaxTables];
    __shared__ int ring_cnt;
    __shared__ long long ring_base;
    BlockTally gone{tally_mem[0]}, dropped{tally_mem[1]};
    RingTrip ring{&ring_cnt, &ring_base};
    gone.clear();
    dropped.clear();
    const int lig = threadIdx.x % kGroup;
    const int64_t per_trip = (int64_t)gridDim.x * kGroupsPerBlock;
    const int64_t trips = (n_evict + per_trip - 1) / per_trip;       // the same for every workgroup
    for (int64_t trip = 0; trip < trips; trip++) {
        const int64_t jev = trip * per_trip + (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup;
        int32_t freed[kMaxPush];
        int n_freed = 0;
        ring.begin();
        if (jev < n_evict) {
            const int t = table_of(d.ev_base, g.T, jev);
            const int64_t e = jev - d.ev_base[t];
            const int64_t q = d.cand_base[t] + e;
            const int32_t s = d.cand_val_sorted[q];
            const int64_t gr = d.slot_row[s], r = gr - g.row_base[t];
            const int64_t idx = d.pend_n[t] + e;
            const int64_t grp = idx / g.W;
            const int w = (int)(idx - grp * g.W);
            const bool written = grp < d.flushes[t];           // its buffer fills up in this reclaim
            char *dst;
            if (written) {
                const int64_t k = d.log_head[t] + grp * g.ws + w / g.vpo;
                const int j = w % g.vpo;
                const int64_t lp = g.n_init[t] + k;
                dst = g.store[t] + lp * g.page_bytes + (int64_t)j * g.row_bytes;
                if (lig == 0) {
                    d.log_rows[((int64_t)t * g.log_cap + k) * g.vpo + j] = (int32_t)r;
                    d.row_loc[gr] = (int32_t)lp;
                    d.row_pend[gr] = -1;
                }
            } else {
                dst = d.pend_data + ((int64_t)t * g.W + w) * g.row_bytes;
                if (lig == 0) {
                    d.pend_row[(int64_t)t * g.W + w] = (int32_t)r;
                    d.pend_cnt[(int64_t)t * g.W + w] = d.slot_cnt[s];
                    d.row_pend[gr] = w;
                }
            }
            copy_row(g, d, t, dst, d.data + (int64_t)s * g.row_bytes, lig);
            if (lig == 0) {
                // the slot's key stays non-zero until clear_keys_kernel: evictions of neighbouring rows run beside
                // this one and take a zero key for "unlisted"
                d.row_slot[gr] = -1;
                freed[n_freed++] = s;
                gone.add(t, 1);
            }
            // unlisted residents of the row's home page are dropped (:492-496); the eviction that fills a buffer
            // leaves the loop early and only reaches the rows below it (:486-489)
            const int64_t hp0 = (r / g.vpo) * g.vpo;
            const bool fills = w == g.W - 1;
            for (int64_t i = hp0 + lig; i < hp0 + g.vpo && i < g.rows[t]; i += kGroup) {
                if (i == r || (fills && i > r)) continue;
                const int64_t gi = g.row_base[t] + i;
                const int32_t si = d.row_slot[gi];
                if (si >= 0 && d.slot_key[si] == 0 && atomicCAS(&d.row_slot[gi], si, -1) == si) {
                    freed[n_freed++] = si;
                    dropped.add(t, 1);
                }
            }
        }
        // the freed slots of the whole workgroup go back to the ring behind one move of its tail
        const int my = ring.take_n(n_freed);
        const long long first = ring.reserve(&d.scalars[S_TAIL]);
        for (int u = 0; u < n_freed; u++) d.ring[(first + my + u) % d.S] = freed[u];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.T; t += kThreads) {
        const long long a = tally_mem[0][t], b = tally_mem[1][t];
        if (a) {
            atomicAdd((unsigned long long *)&d.counters[t * kCounters + C_EVICT], (unsigned long long)a);
            atomicAdd((unsigned long long *)&d.counters[t * kCounters + C_LISTED], (unsigned long long)(-a));
        }
        if (b) atomicAdd((unsigned long long *)&d.counters[t * kCounters + C_DROPPED], (unsigned long long)b);
        if (a + b) atomicAdd((unsigned long long *)&d.counters[t * kCounters + C_RESIDENT], (unsigned long long)(-(a + b)));
    }
}

__global__ __launch_bounds__(kThreads) void clear_keys_kernel(Dev d, int64_t n_cand) {
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n_cand; q += (int64_t)gridDim.x * kThreads) {
        const int t = (int)(d.cand_key_sorted[q] >> 58);
        if (q - d.cand_base[t] < d.k_t[t]) d.slot_key[d.cand_val_sorted[q]] = 0;
    }
}

__global__ void finish_reclaim_kernel(Geom g, Dev d) {
    for (int t = threadIdx.x; t < g.T; t += blockDim.x) {
        const int64_t fl = d.flushes[t];
        d.pend_n[t] = (d.pend_n[t] + d.k_t[t]) % g.W;
        d.log_head[t] += fl * g.ws;
        d.counters[t * kCounters + C_WRITE] += fl * g.ws;              // write_page_count += ws_opt (:613)
        d.counters[t * kCounters + C_PENDING] = d.pend_n[t];
        d.counters[t * kCounters + C_WRITTEN] = d.log_head[t];
        if (fl) atomicAdd((unsigned long long *)&d.scalars[S_PAGE_COUNT], (unsigned long long)(-(fl * g.ws)));   // :614
    }
}

__global__ void reset_rw_kernel(int64_t *cnt, int T) {
    const int t = threadIdx.x;
    if (t < T) { cnt[t * kCounters + C_READ] = 0; cnt[t * kCounters + C_WRITE] = 0; }
}

__global__ void init_kernel(Geom g, Dev d, int64_t total_rows) {
    const int64_t n = total_rows > d.S ? total_rows : d.S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < d.S) { d.ring[i] = (int32_t)i; d.slot_first[i] = 0x7fffffff; }
        if (i < total_rows) {
            const int t = table_of(g.row_base, g.T, i);
            d.row_loc[i] = (int32_t)((i - g.row_base[t]) / g.vpo);   // transfer_PA_from_init (src/emb_init2.cpp:792-806)
        }
    }
}

static unsigned grid_for(int64_t n, int per_block = kThreads) {
    const int64_t b = ceil_div(n > 0 ? n : 1, per_block);
    return (unsigned)(b < 4096 ? b : 4096);
}

template <typename T>
static int dev_alloc(T **p, size_t count) {
    DLRM_HIP(hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * (count ? count : 1)));
    return DLRM_OK;
}

static int layout_of(const dlrm_vcache *c, const int64_t *h_table_offsets, IdLayout *l) {
    for (int t = 0; t <= c->g.T; t++) {
        l->off[t] = h_table_offsets[t];
        DLRM_REQUIRE(t == 0 ? l->off[0] == 0 : l->off[t] >= l->off[t - 1], "table offsets must start at 0 and ascend");
    }
    DLRM_REQUIRE(l->off[c->g.T] < (1ll << 31), "too many ids in one call");
    return DLRM_OK;
}

static int ensure_sort_tmp(dlrm_vcache *c, size_t bytes) {
    if (bytes <= c->sort_tmp_bytes) return DLRM_OK;
    (void)hipFree(c->sort_tmp);
    c->sort_tmp = nullptr; c->sort_tmp_bytes = 0;
    DLRM_HIP(hipMalloc(&c->sort_tmp, bytes));
    c->sort_tmp_bytes = bytes;
    return DLRM_OK;
}

}  // namespace vc
}  // namespace dlrm

using namespace dlrm;
using namespace dlrm::vc;

extern "C" int dlrm_vcache_destroy(dlrm_vcache *c) {
    if (!c) return DLRM_OK;
    Dev &d = c->d;
    (void)hipFree(d.row_slot); (void)hipFree(d.row_loc); (void)hipFree(d.row_pend); (void)hipFree(d.slot_row);
    (void)hipFree(d.slot_key); (void)hipFree(d.slot_cnt); (void)hipFree(d.slot_first); (void)hipFree(d.slot_occ);
    (void)hipFree(d.ring); (void)hipFree(d.scalars); (void)hipFree(d.counters); (void)hipFree(d.page_stamp);
    (void)hipFree(d.log_rows); (void)hipFree(d.pend_row); (void)hipFree(d.pend_cnt); (void)hipFree(d.pend_data);
    (void)hipFree(d.pend_n); (void)hipFree(d.log_head); (void)hipFree(d.k_t); (void)hipFree(d.cand_base); (void)hipFree(d.ev_base);
    (void)hipFree(d.flushes); (void)hipFree(d.miss); (void)hipFree(d.cand_key); (void)hipFree(d.cand_key_sorted);
    (void)hipFree(d.cand_val); (void)hipFree(d.cand_val_sorted);
    if (c->owns_pool) (void)hipFree(d.data);
    if (c->owns_state) (void)hipFree(d.state);
    (void)hipFree(d.pend_state);
    (void)hipFree(c->sort_tmp);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    delete c;
    return DLRM_OK;
}

extern "C" int dlrm_vcache_create(const dlrm_vcache_config *cfg, dlrm_vcache **out) {
    DLRM_REQUIRE(cfg != nullptr && out != nullptr, "NULL argument");
    DLRM_REQUIRE(cfg->num_tables >= 1 && cfg->num_tables <= kMaxTables, "num_tables must be 1..%d", kMaxTables);
    DLRM_REQUIRE(cfg->dim >= 1 && cfg->rows_per_page >= 1 && cfg->evict_group >= 1, "bad geometry");
    DLRM_REQUIRE(cfg->rows_per_page <= 128, "rows_per_page %d: at most 128", cfg->rows_per_page);
    DLRM_REQUIRE(cfg->table_rows != nullptr && cfg->h_table_store != nullptr, "table_rows / h_table_store is NULL");
    DLRM_REQUIRE(cfg->num_slots >= 1 && cfg->num_slots < (1ll << 31), "num_slots out of range");
    DLRM_REQUIRE(cfg->log_pages_per_table >= cfg->evict_group, "log_pages_per_table must hold at least one write");
    DLRM_REQUIRE(cfg->row_dtype == DLRM_F32 || cfg->row_dtype == DLRM_BF16, "row_dtype %d", cfg->row_dtype);
    const int64_t row_bytes = (int64_t)cfg->dim * (cfg->row_dtype == DLRM_F32 ? 4 : 2);
    DLRM_REQUIRE(row_bytes % 16 == 0, "a row (%lld bytes) must be a multiple of 16 bytes", (long long)row_bytes);
    const int64_t W = (int64_t)cfg->evict_group * cfg->rows_per_page;
    // a row that is evicted again while its first copy still waits in the write buffer takes a special path
    // in the reference (emb_cache2.cpp:440-450); a list of at least one buffer's length rules that out
    DLRM_REQUIRE(cfg->cache_size >= W, "cache_size (%lld rows per table) must be at least evict_group * rows_per_page = %lld",
                 (long long)cfg->cache_size, (long long)W);

    dlrm_vcache *c = new dlrm_vcache();
    memset(c, 0, sizeof(*c));
    Geom &g = c->g;
    g.T = cfg->num_tables; g.vpo = cfg->rows_per_page; g.ws = cfg->evict_group; g.dim = cfg->dim; g.W = (int)W;
    g.row_bytes = row_bytes; g.page_bytes = row_bytes * g.vpo; g.log_cap = cfg->log_pages_per_table;
    g.row_base[0] = 0; g.page_base[0] = 0;
    for (int t = 0; t < g.T; t++) {
        if (cfg->table_rows[t] < 1 || cfg->h_table_store[t] == nullptr) {
            delete c;
            return set_error(DLRM_ERR_INVALID, "table %d: rows < 1 or NULL backing store", t);
        }
        g.rows[t] = cfg->table_rows[t];
        g.n_init[t] = ceil_div(g.rows[t], g.vpo);
        g.row_base[t + 1] = g.row_base[t] + g.rows[t];
        g.page_base[t + 1] = g.page_base[t] + g.n_init[t] + g.log_cap;
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
    c->total_rows = g.row_base[g.T];
    c->total_pages = g.page_base[g.T];
    if (c->total_rows >= (1ll << 31) || c->total_pages >= (1ll << 31)) {
        delete c;
        return set_error(DLRM_ERR_INVALID, "too many rows (%lld) or pages (%lld) for 32-bit ids", (long long)c->total_rows,
                         (long long)c->total_pages);
    }
    Dev &d = c->d;
    d.S = cfg->num_slots;
    const size_t R = (size_t)c->total_rows, S = (size_t)d.S, T = (size_t)g.T;
    int rc = DLRM_OK;
    if (!rc) rc = dev_alloc(&d.row_slot, R);
    if (!rc) rc = dev_alloc(&d.row_loc, R);
    if (!rc) rc = dev_alloc(&d.row_pend, R);
    if (!rc) rc = dev_alloc(&d.slot_row, S);
    if (!rc) rc = dev_alloc(&d.slot_key, S);
    if (!rc) rc = dev_alloc(&d.slot_cnt, S);
    if (!rc) rc = dev_alloc(&d.slot_first, S);
    if (!rc) rc = dev_alloc(&d.slot_occ, S);
    if (!rc) rc = dev_alloc(&d.ring, S);
    if (!rc) rc = dev_alloc(&d.scalars, (size_t)S_COUNT);
    if (!rc) rc = dev_alloc(&d.counters, T * kCounters);
    if (!rc) rc = dev_alloc(&d.page_stamp, (size_t)c->total_pages);
    if (!rc) rc = dev_alloc(&d.log_rows, T * (size_t)g.log_cap * g.vpo);
    if (!rc) rc = dev_alloc(&d.pend_row, T * (size_t)W);
    if (!rc) rc = dev_alloc(&d.pend_cnt, T * (size_t)W);
    if (!rc) rc = dev_alloc(&d.pend_data, T * (size_t)W * (size_t)row_bytes);
    if (!rc) rc = dev_alloc(&d.pend_n, T);
    if (!rc) rc = dev_alloc(&d.log_head, T);
    if (!rc) rc = dev_alloc(&d.k_t, T);
    if (!rc) rc = dev_alloc(&d.cand_base, T);
    if (!rc) rc = dev_alloc(&d.ev_base, T + 1);
    if (!rc) rc = dev_alloc(&d.flushes, T);
    if (!rc) rc = dev_alloc(&d.cand_key, S);
    if (!rc) rc = dev_alloc(&d.cand_key_sorted, S);
    if (!rc) rc = dev_alloc(&d.cand_val, S);
    if (!rc) rc = dev_alloc(&d.cand_val_sorted, S);
    if (cfg->d_slot_pool) {
        d.data = static_cast<char *>(cfg->d_slot_pool);
        if (reinterpret_cast<uintptr_t>(d.data) % 16 != 0) rc = set_error(DLRM_ERR_INVALID, "d_slot_pool must be 16-byte aligned");
    } else {
        c->owns_pool = true;
        if (!rc) rc = dev_alloc(&d.data, S * (size_t)row_bytes);
    }
    if (g.has_state) {
        if (!rc) rc = dev_alloc(&d.pend_state, T * (size_t)W);
        if (cfg->d_row_state_pool) {
            d.state = cfg->d_row_state_pool;
        } else {
            c->owns_state = true;
            if (!rc) rc = dev_alloc(&d.state, S);
            if (!rc) (void)hipMemset(d.state, 0, sizeof(float) * S);
        }
    }
    if (!rc && hipHostMalloc(reinterpret_cast<void **>(&c->h_scalars), sizeof(int64_t) * S_COUNT, 0) != hipSuccess)
        rc = set_error(DLRM_ERR_NOMEM, "hipHostMalloc failed");
    if (rc) { dlrm_vcache_destroy(c); return rc; }
    (void)hipMemset(d.row_slot, 0xff, sizeof(int32_t) * R);
    (void)hipMemset(d.row_pend, 0xff, sizeof(int32_t) * R);
    (void)hipMemset(d.slot_key, 0, sizeof(uint64_t) * S);
    (void)hipMemset(d.slot_cnt, 0, sizeof(int32_t) * S);
    (void)hipMemset(d.slot_occ, 0, sizeof(int32_t) * S);
    (void)hipMemset(d.slot_row, 0, sizeof(int32_t) * S);
    (void)hipMemset(d.counters, 0, sizeof(int64_t) * T * kCounters);
    (void)hipMemset(d.page_stamp, 0, sizeof(uint32_t) * (size_t)c->total_pages);
    (void)hipMemset(d.pend_n, 0, sizeof(int64_t) * T);
    (void)hipMemset(d.log_head, 0, sizeof(int64_t) * T);
    int64_t init[S_COUNT] = {0};
    init[S_TAIL] = d.S;
    init[S_CACHE_SIZE] = cfg->cache_size;
    (void)hipMemcpy(d.scalars, init, sizeof(init), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, g, d, c->total_rows);
    if (hipDeviceSynchronize() != hipSuccess) {
        dlrm_vcache_destroy(c);
        return set_error(DLRM_ERR_HIP, "cache initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = c;
    return DLRM_OK;
}

extern "C" int dlrm_vcache_storage(dlrm_vcache *c, void **d_data, int64_t *total_rows) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    if (d_data) *d_data = c->d.data;
    if (total_rows) *total_rows = c->d.S;
    return DLRM_OK;
}

static int ensure_miss(dlrm_vcache *c, int64_t n) {
    if (n <= c->miss_cap) return DLRM_OK;
    (void)hipFree(c->d.miss);
    c->d.miss = nullptr;
    c->miss_cap = 0;
    if (int rc = dev_alloc(&c->d.miss, (size_t)n)) return rc;
    c->miss_cap = n;
    return DLRM_OK;
}

extern "C" int dlrm_vcache_lookup(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                                  void *d_rows_out, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_table_offsets != nullptr, "NULL argument");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    IdLayout l;
    if (int rc = layout_of(c, h_table_offsets, &l)) return rc;
    const int64_t n = l.off[c->g.T];
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_ids != nullptr && d_rows_out != nullptr, "d_ids / d_rows_out is NULL");
    if (int rc = ensure_miss(c, n)) return rc;
    hipStream_t s = as_stream(stream);
    const Geom &g = c->g;
    const Dev &d = c->d;
    hipLaunchKernelGGL(begin_lookup_kernel, dim3(1), dim3(1), 0, s, d);
    if (idx_dtype == DLRM_I64) {
        const int64_t *ids = static_cast<const int64_t *>(d_ids);
        hipLaunchKernelGGL(probe_kernel<int64_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n);
        hipLaunchKernelGGL(fill_kernel, dim3(kLinkBlocks), dim3(kThreads), 0, s, g, d);
        hipLaunchKernelGGL(occurrences_kernel<int64_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n);
        hipLaunchKernelGGL(list_kernel<int64_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n,
                           static_cast<int64_t *>(d_rows_out));
    } else {
        const int32_t *ids = static_cast<const int32_t *>(d_ids);
        hipLaunchKernelGGL(probe_kernel<int32_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n);
        hipLaunchKernelGGL(fill_kernel, dim3(kLinkBlocks), dim3(kThreads), 0, s, g, d);
        hipLaunchKernelGGL(occurrences_kernel<int32_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n);
        hipLaunchKernelGGL(list_kernel<int32_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n,
                           static_cast<int32_t *>(d_rows_out));
    }
    hipLaunchKernelGGL(finish_lookup_kernel, dim3(1), dim3(1), 0, s, d);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_vcache_restore(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                                   void *d_rows_out, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_table_offsets != nullptr, "NULL argument");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    IdLayout l;
    if (int rc = layout_of(c, h_table_offsets, &l)) return rc;
    const int64_t n = l.off[c->g.T];
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_ids != nullptr, "d_ids is NULL");
    hipStream_t s = as_stream(stream);
    const Geom &g = c->g;
    const Dev &d = c->d;
    if (idx_dtype == DLRM_I64) {
        const int64_t *ids = static_cast<const int64_t *>(d_ids);
        hipLaunchKernelGGL(restore_kernel<int64_t>, dim3(kLinkBlocks), dim3(kThreads), 0, s, g, d, l, ids, n);
        if (d_rows_out)
            hipLaunchKernelGGL(translate_kernel<int64_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n,
                               static_cast<int64_t *>(d_rows_out));
    } else {
        const int32_t *ids = static_cast<const int32_t *>(d_ids);
        hipLaunchKernelGGL(restore_kernel<int32_t>, dim3(kLinkBlocks), dim3(kThreads), 0, s, g, d, l, ids, n);
        if (d_rows_out)
            hipLaunchKernelGGL(translate_kernel<int32_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, g, d, l, ids, n,
                               static_cast<int32_t *>(d_rows_out));
    }
    hipLaunchKernelGGL(finish_lookup_kernel, dim3(1), dim3(1), 0, s, d);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_vcache_row_state(dlrm_vcache *c, float **d_state) {
    DLRM_REQUIRE(c != nullptr && d_state != nullptr, "NULL argument");
    *d_state = c->d.state;
    return DLRM_OK;
}

extern "C" int dlrm_vcache_read_rows(dlrm_vcache *c, const void *d_ids, int32_t idx_dtype, const int64_t *h_table_offsets,
                                     void *d_out, float *d_state_out, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_table_offsets != nullptr, "NULL argument");
    DLRM_REQUIRE(idx_dtype == DLRM_I32 || idx_dtype == DLRM_I64, "idx_dtype %d", idx_dtype);
    IdLayout l;
    if (int rc = layout_of(c, h_table_offsets, &l)) return rc;
    const int64_t n = l.off[c->g.T];
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_ids != nullptr && d_out != nullptr, "d_ids / d_out is NULL");
    DLRM_REQUIRE(reinterpret_cast<uintptr_t>(d_out) % 16 == 0, "d_out must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    if (idx_dtype == DLRM_I64)
        hipLaunchKernelGGL(read_rows_kernel<int64_t>, dim3(kLinkBlocks), dim3(kThreads), 0, s, c->g, c->d, l,
                           static_cast<const int64_t *>(d_ids), n, reinterpret_cast<char *>(d_out), d_state_out);
    else
        hipLaunchKernelGGL(read_rows_kernel<int32_t>, dim3(kLinkBlocks), dim3(kThreads), 0, s, c->g, c->d, l,
                           static_cast<const int32_t *>(d_ids), n, reinterpret_cast<char *>(d_out), d_state_out);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_vcache_reclaim(dlrm_vcache *c, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    hipStream_t s = as_stream(stream);
    const Geom &g = c->g;
    const Dev &d = c->d;
    hipLaunchKernelGGL(plan_reclaim_kernel, dim3(1), dim3(1), 0, s, g, d);
    DLRM_LAUNCH_CHECK();
    DLRM_HIP(hipMemcpyAsync(c->h_scalars, d.scalars, sizeof(int64_t) * S_COUNT, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    const int64_t n_listed = c->h_scalars[S_N_LISTED], n_evict = c->h_scalars[S_N_EVICT];
    if (n_evict == 0) return DLRM_OK;
    DLRM_REQUIRE(n_listed <= d.S, "internal: %lld listed rows in %lld slots", (long long)n_listed, (long long)d.S);
    hipLaunchKernelGGL(collect_kernel, dim3(grid_for(d.S, kThreads * kCollectPer)), dim3(kThreads), 0, s, g, d);
    size_t tmp = 0;
    DLRM_HIP(rocprim::radix_sort_pairs(nullptr, tmp, d.cand_key, d.cand_key_sorted, d.cand_val, d.cand_val_sorted,
                                       (size_t)n_listed, 0u, 64u, s, false));
    if (int rc = ensure_sort_tmp(c, tmp)) return rc;
    tmp = c->sort_tmp_bytes;
    DLRM_HIP(rocprim::radix_sort_pairs(c->sort_tmp, tmp, d.cand_key, d.cand_key_sorted, d.cand_val, d.cand_val_sorted,
                                       (size_t)n_listed, 0u, 64u, s, false));
    hipLaunchKernelGGL(flush_staged_kernel, dim3(grid_for((int64_t)g.T * g.W, kGroupsPerBlock)), dim3(kThreads), 0, s, g, d);
    hipLaunchKernelGGL(evict_kernel, dim3(kLinkBlocks), dim3(kThreads), 0, s, g, d, n_evict);
    hipLaunchKernelGGL(clear_keys_kernel, dim3(grid_for(n_listed)), dim3(kThreads), 0, s, d, n_listed);
    hipLaunchKernelGGL(finish_reclaim_kernel, dim3(1), dim3(64), 0, s, g, d);
    DLRM_LAUNCH_CHECK();
    return DLRM_OK;
}

extern "C" int dlrm_vcache_counters(dlrm_vcache *c, int64_t *h_out, int32_t reset_read_write, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_out != nullptr, "NULL argument");
    hipStream_t s = as_stream(stream);
    DLRM_HIP(hipMemcpyAsync(h_out, c->d.counters, sizeof(int64_t) * c->g.T * kCounters, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    if (reset_read_write) {
        // cache2::read_page(i) / write_page(i) zero the two page counters after logging (include/data2.h:263-274)
        hipLaunchKernelGGL(reset_rw_kernel, dim3(1), dim3(64), 0, s, c->d.counters, c->g.T);
        DLRM_LAUNCH_CHECK();
    }
    return DLRM_OK;
}

extern "C" int dlrm_vcache_state(dlrm_vcache *c, dlrm_vcache_state_t *h_state, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_state != nullptr, "NULL argument");
    hipStream_t s = as_stream(stream);
    DLRM_HIP(hipMemcpyAsync(c->h_scalars, c->d.scalars, sizeof(int64_t) * S_COUNT, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    h_state->page_count = c->h_scalars[S_PAGE_COUNT];
    h_state->cache_size = c->h_scalars[S_CACHE_SIZE];
    h_state->free_slots = c->h_scalars[S_TAIL] - c->h_scalars[S_HEAD];
    h_state->epoch = c->h_scalars[S_EPOCH];
    h_state->error_flags = c->h_scalars[S_ERROR];
    h_state->last_lookup_pages_read = c->h_scalars[S_N_MISS];
    h_state->num_slots = c->d.S;
    h_state->total_rows = c->total_rows;
    return DLRM_OK;
}

extern "C" int dlrm_vcache_set_cache_size(dlrm_vcache *c, int64_t cache_size, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    DLRM_REQUIRE(cache_size >= c->g.W, "cache_size must be at least evict_group * rows_per_page = %d", c->g.W);
    hipStream_t s = as_stream(stream);
    DLRM_HIP(hipMemcpyAsync(c->d.scalars + S_CACHE_SIZE, &cache_size, sizeof(int64_t), hipMemcpyHostToDevice, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}

extern "C" int dlrm_vcache_export_table(dlrm_vcache *c, int32_t table, int32_t *h_row_slot, int32_t *h_row_page,
                                        int32_t *h_row_pending, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && table >= 0 && table < c->g.T, "bad argument");
    hipStream_t s = as_stream(stream);
    const size_t bytes = sizeof(int32_t) * (size_t)c->g.rows[table];
    const int64_t b = c->g.row_base[table];
    if (h_row_slot) DLRM_HIP(hipMemcpyAsync(h_row_slot, c->d.row_slot + b, bytes, hipMemcpyDeviceToHost, s));
    if (h_row_page) DLRM_HIP(hipMemcpyAsync(h_row_page, c->d.row_loc + b, bytes, hipMemcpyDeviceToHost, s));
    if (h_row_pending) DLRM_HIP(hipMemcpyAsync(h_row_pending, c->d.row_pend + b, bytes, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}

extern "C" int dlrm_vcache_export_slots(dlrm_vcache *c, int32_t *h_slot_row, uint64_t *h_slot_key, int32_t *h_slot_count,
                                        dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr, "cache is NULL");
    hipStream_t s = as_stream(stream);
    const size_t S = (size_t)c->d.S;
    if (h_slot_row) DLRM_HIP(hipMemcpyAsync(h_slot_row, c->d.slot_row, sizeof(int32_t) * S, hipMemcpyDeviceToHost, s));
    if (h_slot_key) DLRM_HIP(hipMemcpyAsync(h_slot_key, c->d.slot_key, sizeof(uint64_t) * S, hipMemcpyDeviceToHost, s));
    if (h_slot_count) DLRM_HIP(hipMemcpyAsync(h_slot_count, c->d.slot_cnt, sizeof(int32_t) * S, hipMemcpyDeviceToHost, s));
    DLRM_HIP(hipStreamSynchronize(s));
    return DLRM_OK;
}
