// Collectives of the sharded step over RCCL (xGMI inside a node): thin wrappers so that a C or C++ host can
// run the exchanges of include/dlrm_amd.h's sharded layout without a Python runtime.  One communicator per
// process and GPU; every call is enqueued on the caller's stream.
//
// The reference runs on one host and has no counterpart; the layout follows its striping of tables over
// parallel units (include/data2.h:150, src/emb_cache2.cpp:701-709): whole tables per rank (all-to-all of the
// pooled vectors and of their gradients), or row ranges of a table per rank (reduce-scatter / all-gather).
//
// Rehearsal transport.  RCCL refuses two ranks on one device, so a box with one GPU cannot run more than one rank
// of a host that speaks dlrm_comm.  An id drawn by dlrm_comm_unique_id_rehearsal selects a second transport with
// the same calls and results: the ranks (processes of one machine, any devices) meet in a POSIX shared-memory
// segment; every call drains its stream, copies this rank's send buffer to its mailbox in the segment, meets the
// others at a barrier, builds its result from the mailboxes (sums in rank order) and copies it back.  It exists
// so that the packing, counts and call order of a multi-rank host can be checked on one GPU; it is never chosen
// by an id from dlrm_comm_unique_id, and it is as slow as it sounds.
//
// RCCL is opened at run time (dlopen) rather than linked: a process that already carries an RCCL (PyTorch
// ships one) keeps exactly one copy.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <mutex>

#include <time.h>

#include "common.h"

namespace dlrm {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_error[256] = "";

static void load_rccl() {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "cannot open librccl: %s", dlerror());
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(g_rccl.lib, name);
        if (!p) { ok = false; snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl lacks %s", name); }
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.ReduceScatter = reinterpret_cast<decltype(g_rccl.ReduceScatter)>(sym("ncclReduceScatter"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    if (!ok) g_rccl.lib = nullptr;
}

static int rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.lib) return set_error(DLRM_ERR_HIP, "%s", g_rccl_error);
    return DLRM_OK;
}

#define DLRM_RCCL(call)                                                                                       \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return ::dlrm::set_error(DLRM_ERR_HIP, "%s failed: %s", #call, ::dlrm::g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace dlrm

namespace dlrm {

constexpr char kShmMagic[8] = {'D', 'L', 'R', 'M', 's', 'h', 'm', '1'};
constexpr int kShmMaxRanks = 16;

struct ShmHeader {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> error;        // a rank that failed inside a call says so here before it goes on to the barriers
    std::atomic<int64_t> box_bytes;   // mailbox size of the rank that created the segment: every rank must agree
    int64_t counts[kShmMaxRanks][kShmMaxRanks];   // counts[r][p]: words rank r sends to peer p in the current call
};

struct ShmTransport {
    char name[64];
    ShmHeader *hdr = nullptr;
    char *base = nullptr;       // the mapping
    size_t map_bytes = 0, box_bytes = 0;
    float *box(int r) const { return reinterpret_cast<float *>(base + 4096 + (size_t)r * box_bytes); }
};

}  // namespace dlrm

struct dlrm_comm {
    ncclComm_t comm;
    int rank, world;
    dlrm::ShmTransport *shm;    // rehearsal transport, else NULL
};

using namespace dlrm;

// Every rank arrives, or the call fails: a peer that never comes (it died, or left the call early) must not leave
// the others spinning for ever.  DLRM_COMM_REHEARSAL_TIMEOUT_S (default 300).
static int shm_barrier(dlrm_comm *c) {
    static const double limit = [] { const char *e = getenv("DLRM_COMM_REHEARSAL_TIMEOUT_S"); return e ? atof(e) : 300.0; }();
    ShmHeader *h = c->shm->hdr;
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
    } else {
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0; h->generation.load(std::memory_order_acquire) == gen; spins++) {
            usleep(50);
            if ((spins & 1023) == 1023) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
                    h->error.store(1, std::memory_order_release);
                    return set_error(DLRM_ERR_HIP, "rehearsal transport: rank %d waited %.0f s for its peers", c->rank, limit);
                }
            }
        }
    }
    return DLRM_OK;
}

// what a call returns after its last barrier: its own status, or that a peer failed inside the same call
static int shm_result(dlrm_comm *c, int rc) {
    if (rc) return rc;
    if (c->shm->hdr->error.load(std::memory_order_acquire))
        return set_error(DLRM_ERR_HIP, "rehearsal transport: a peer of rank %d failed inside this call", c->rank);
    return DLRM_OK;
}

// this rank's n words into its mailbox (after everything queued on the stream before the call has run).  A failure
// is published in the header and returned, but the caller still goes through the call's barriers: its peers are
// waiting there.
static int shm_post(dlrm_comm *c, const float *d_send, int64_t n, hipStream_t s) {
    int rc = DLRM_OK;
    if ((size_t)n * 4 > c->shm->box_bytes)
        rc = set_error(DLRM_ERR_INVALID, "rehearsal transport: %lld words exceed the %zu-byte mailbox (set DLRM_COMM_REHEARSAL_MB)",
                       (long long)n, c->shm->box_bytes);
    else if (hipStreamSynchronize(s) != hipSuccess)
        rc = set_error(DLRM_ERR_HIP, "rehearsal transport: the stream failed before the exchange");
    else if (n && hipMemcpy(c->shm->box(c->rank), d_send, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = set_error(DLRM_ERR_HIP, "rehearsal transport: copy from the device failed");
    if (rc) c->shm->hdr->error.store(1, std::memory_order_release);
    return rc;
}

static_assert(DLRM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "communicator id size");

extern "C" int dlrm_comm_unique_id(uint8_t *id) {
    DLRM_REQUIRE(id != nullptr, "id is NULL");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    DLRM_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return DLRM_OK;
}

extern "C" int dlrm_comm_unique_id_rehearsal(uint8_t *id) {
    DLRM_REQUIRE(id != nullptr, "id is NULL");
    memset(id, 0, DLRM_COMM_ID_BYTES);
    memcpy(id, kShmMagic, sizeof(kShmMagic));
    static std::atomic<unsigned> serial{0};
    snprintf(reinterpret_cast<char *>(id) + 8, 64, "/dlrm_comm_%d_%u_%llx", (int)getpid(), serial.fetch_add(1),
             (unsigned long long)splitmix64((uint64_t)getpid() * 1000003ull + (uint64_t)clock()));
    return DLRM_OK;
}

static int shm_init(const uint8_t *id, int32_t rank, int32_t world, dlrm_comm **out) {
    DLRM_REQUIRE(world <= kShmMaxRanks, "rehearsal transport: at most %d ranks", kShmMaxRanks);
    ShmTransport *t = new ShmTransport();
    snprintf(t->name, sizeof(t->name), "%s", reinterpret_cast<const char *>(id) + 8);
    const char *mb = getenv("DLRM_COMM_REHEARSAL_MB");
    t->box_bytes = (size_t)(mb ? atol(mb) : 64) << 20;
    t->map_bytes = 4096 + (size_t)world * t->box_bytes;
    static_assert(sizeof(ShmHeader) <= 4096, "header page");
    const int fd = shm_open(t->name, O_CREAT | O_RDWR, 0600);     // a new segment is zero-filled: counters start at 0
    if (fd < 0 || ftruncate(fd, (off_t)t->map_bytes) != 0) {
        if (fd >= 0) close(fd);
        delete t;
        return set_error(DLRM_ERR_HIP, "rehearsal transport: cannot create shared memory %s", t->name);
    }
    void *p = mmap(nullptr, t->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete t; return set_error(DLRM_ERR_NOMEM, "rehearsal transport: mmap failed"); }
    t->base = static_cast<char *>(p);
    t->hdr = reinterpret_cast<ShmHeader *>(p);
    dlrm_comm *h = new dlrm_comm();
    h->comm = nullptr; h->rank = rank; h->world = world; h->shm = t;
    // the first rank to get here records its mailbox size; a rank configured otherwise would read other offsets
    int64_t none = 0;
    t->hdr->box_bytes.compare_exchange_strong(none, (int64_t)t->box_bytes);
    int rc = shm_barrier(h);                        // everyone has mapped it
    if (!rc && t->hdr->box_bytes.load() != (int64_t)t->box_bytes) {
        t->hdr->error.store(1, std::memory_order_release);
        rc = set_error(DLRM_ERR_INVALID, "rehearsal transport: rank %d has %zu-byte mailboxes, the segment was made with %lld "
                       "(DLRM_COMM_REHEARSAL_MB differs between the ranks)", rank, t->box_bytes, (long long)t->hdr->box_bytes.load());
    }
    if (rank == 0) shm_unlink(t->name);             // the mappings keep it alive; nothing is left behind
    const int rc2 = shm_barrier(h);
    rc = shm_result(h, rc ? rc : rc2);
    if (rc) { munmap(t->base, t->map_bytes); delete t; delete h; return rc; }
    *out = h;
    return DLRM_OK;
}

extern "C" int dlrm_comm_init(const uint8_t *id, int32_t rank, int32_t world, dlrm_comm **out) {
    DLRM_REQUIRE(id != nullptr && out != nullptr, "NULL argument");
    DLRM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d of %d", rank, world);
    if (memcmp(id, kShmMagic, sizeof(kShmMagic)) == 0) return shm_init(id, rank, world, out);
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    DLRM_RCCL(g_rccl.CommInitRank(&c, world, u, rank));
    dlrm_comm *h = new dlrm_comm();
    h->comm = c; h->rank = rank; h->world = world; h->shm = nullptr;
    *out = h;
    return DLRM_OK;
}

extern "C" int dlrm_comm_destroy(dlrm_comm *c) {
    if (!c) return DLRM_OK;
    if (c->shm) {
        munmap(c->shm->base, c->shm->map_bytes);
        delete c->shm;
        delete c;
        return DLRM_OK;
    }
    const ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return set_error(DLRM_ERR_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return DLRM_OK;
}

extern "C" int dlrm_comm_all_reduce_sum(dlrm_comm *c, float *d_buf, int64_t n, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && n >= 0, "bad argument");
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_buf != nullptr, "d_buf is NULL");
    if (c->shm) {
        int rc = shm_post(c, d_buf, n, as_stream(stream));
        if (int b = shm_barrier(c)) return b;
        float *sum = nullptr;
        if (!rc && !c->shm->hdr->error.load()) {
            sum = new float[(size_t)n];
            for (int64_t i = 0; i < n; i++) {
                float a = 0.f;
                for (int p = 0; p < c->world; p++) a += c->shm->box(p)[i];
                sum[i] = a;
            }
        }
        if (int b = shm_barrier(c)) { delete[] sum; return b; }    // everyone has read every mailbox
        if (sum && hipMemcpy(d_buf, sum, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = set_error(DLRM_ERR_HIP, "rehearsal transport: copy to the device failed");
        delete[] sum;
        return shm_result(c, rc);
    }
    DLRM_RCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat, ncclSum, c->comm, as_stream(stream)));
    return DLRM_OK;
}

extern "C" int dlrm_comm_all_to_all(dlrm_comm *c, const float *d_send, const int64_t *h_send_counts, float *d_recv,
                                    const int64_t *h_recv_counts, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && h_send_counts != nullptr && h_recv_counts != nullptr, "NULL argument");
    hipStream_t s = as_stream(stream);
    int64_t so = 0, ro = 0;
    for (int p = 0; p < c->world; p++)
        DLRM_REQUIRE(h_send_counts[p] >= 0 && h_recv_counts[p] >= 0, "negative count for peer %d", p);
    if (c->shm) {
        int64_t total = 0;
        for (int p = 0; p < c->world; p++) {
            c->shm->hdr->counts[c->rank][p] = h_send_counts[p];
            total += h_send_counts[p];
        }
        int rc = shm_post(c, d_send, total, s);
        if (int b = shm_barrier(c)) return b;
        if (!rc && c->shm->hdr->error.load()) rc = shm_result(c, DLRM_OK);
        for (int p = 0; p < c->world && !rc; p++) {
            int64_t first = 0;
            for (int q = 0; q < c->rank; q++) first += c->shm->hdr->counts[p][q];
            if (c->shm->hdr->counts[p][c->rank] != h_recv_counts[p])
                rc = set_error(DLRM_ERR_INVALID, "rank %d expects %lld words from rank %d, which sends %lld", c->rank,
                               (long long)h_recv_counts[p], p, (long long)c->shm->hdr->counts[p][c->rank]);
            else if (h_recv_counts[p] &&
                     hipMemcpy(d_recv + ro, c->shm->box(p) + first, (size_t)h_recv_counts[p] * 4, hipMemcpyHostToDevice) != hipSuccess)
                rc = set_error(DLRM_ERR_HIP, "rehearsal transport: copy to the device failed");
            ro += h_recv_counts[p];
        }
        if (rc) c->shm->hdr->error.store(1, std::memory_order_release);
        if (int b = shm_barrier(c)) return b;
        return shm_result(c, rc);
    }
    DLRM_RCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->world; p++) {
        ncclResult_t r = ncclSuccess;
        if (h_send_counts[p]) r = g_rccl.Send(d_send + so, (size_t)h_send_counts[p], ncclFloat, p, c->comm, s);
        if (r == ncclSuccess && h_recv_counts[p]) r = g_rccl.Recv(d_recv + ro, (size_t)h_recv_counts[p], ncclFloat, p, c->comm, s);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return set_error(DLRM_ERR_HIP, "send/recv with peer %d failed: %s", p, g_rccl.GetErrorString(r));
        }
        so += h_send_counts[p];
        ro += h_recv_counts[p];
    }
    DLRM_RCCL(g_rccl.GroupEnd());
    return DLRM_OK;
}

extern "C" int dlrm_comm_reduce_scatter_sum(dlrm_comm *c, const float *d_send, float *d_recv, int64_t recv_count,
                                            dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && recv_count >= 0, "bad argument");
    if (recv_count == 0) return DLRM_OK;
    DLRM_REQUIRE(d_send != nullptr && d_recv != nullptr, "NULL buffer");
    if (c->shm) {
        int rc = shm_post(c, d_send, recv_count * c->world, as_stream(stream));
        if (int b = shm_barrier(c)) return b;
        float *sum = nullptr;
        if (!rc && !c->shm->hdr->error.load()) {
            sum = new float[(size_t)recv_count];
            for (int64_t i = 0; i < recv_count; i++) {
                float a = 0.f;
                for (int p = 0; p < c->world; p++) a += c->shm->box(p)[(int64_t)c->rank * recv_count + i];
                sum[i] = a;
            }
        }
        if (int b = shm_barrier(c)) { delete[] sum; return b; }
        if (sum && hipMemcpy(d_recv, sum, (size_t)recv_count * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = set_error(DLRM_ERR_HIP, "rehearsal transport: copy to the device failed");
        delete[] sum;
        return shm_result(c, rc);
    }
    DLRM_RCCL(g_rccl.ReduceScatter(d_send, d_recv, (size_t)recv_count, ncclFloat, ncclSum, c->comm, as_stream(stream)));
    return DLRM_OK;
}

extern "C" int dlrm_comm_all_gather(dlrm_comm *c, const float *d_send, float *d_recv, int64_t send_count,
                                    dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && send_count >= 0, "bad argument");
    if (send_count == 0) return DLRM_OK;
    DLRM_REQUIRE(d_send != nullptr && d_recv != nullptr, "NULL buffer");
    if (c->shm) {
        int rc = shm_post(c, d_send, send_count, as_stream(stream));
        if (int b = shm_barrier(c)) return b;
        if (!rc && !c->shm->hdr->error.load())
            for (int p = 0; p < c->world && !rc; p++)
                if (hipMemcpy(d_recv + (int64_t)p * send_count, c->shm->box(p), (size_t)send_count * 4, hipMemcpyHostToDevice) != hipSuccess)
                    rc = set_error(DLRM_ERR_HIP, "rehearsal transport: copy to the device failed");
        if (rc) c->shm->hdr->error.store(1, std::memory_order_release);
        if (int b = shm_barrier(c)) return b;
        return shm_result(c, rc);
    }
    DLRM_RCCL(g_rccl.AllGather(d_send, d_recv, (size_t)send_count, ncclFloat, c->comm, as_stream(stream)));
    return DLRM_OK;
}
