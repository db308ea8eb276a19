// Collectives of the sharded step over RCCL (xGMI inside a node): thin wrappers so that a C or C++ host can
// run the exchanges of include/dlrm_amd.h's sharded layout without a Python runtime.  One communicator per
// process and GPU; every call is enqueued on the caller's stream.
//
// The reference runs on one host and has no counterpart; the layout follows its striping of tables over
// parallel units (include/data2.h:150, src/emb_cache2.cpp:701-709): whole tables per rank (all-to-all of the
// pooled vectors and of their gradients), or row ranges of a table per rank (reduce-scatter / all-gather).
//
// RCCL is opened at run time (dlopen) rather than linked: a process that already carries an RCCL (PyTorch
// ships one) keeps exactly one copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

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

struct dlrm_comm {
    ncclComm_t comm;
    int rank, world;
};

using namespace dlrm;

static_assert(DLRM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "communicator id size");

extern "C" int dlrm_comm_unique_id(uint8_t *id) {
    DLRM_REQUIRE(id != nullptr, "id is NULL");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    DLRM_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return DLRM_OK;
}

extern "C" int dlrm_comm_init(const uint8_t *id, int32_t rank, int32_t world, dlrm_comm **out) {
    DLRM_REQUIRE(id != nullptr && out != nullptr, "NULL argument");
    DLRM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d of %d", rank, world);
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    DLRM_RCCL(g_rccl.CommInitRank(&c, world, u, rank));
    dlrm_comm *h = new dlrm_comm();
    h->comm = c; h->rank = rank; h->world = world;
    *out = h;
    return DLRM_OK;
}

extern "C" int dlrm_comm_destroy(dlrm_comm *c) {
    if (!c) return DLRM_OK;
    const ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return set_error(DLRM_ERR_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return DLRM_OK;
}

extern "C" int dlrm_comm_all_reduce_sum(dlrm_comm *c, float *d_buf, int64_t n, dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && n >= 0, "bad argument");
    if (n == 0) return DLRM_OK;
    DLRM_REQUIRE(d_buf != nullptr, "d_buf is NULL");
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
    DLRM_RCCL(g_rccl.ReduceScatter(d_send, d_recv, (size_t)recv_count, ncclFloat, ncclSum, c->comm, as_stream(stream)));
    return DLRM_OK;
}

extern "C" int dlrm_comm_all_gather(dlrm_comm *c, const float *d_send, float *d_recv, int64_t send_count,
                                    dlrm_stream_t stream) {
    DLRM_REQUIRE(c != nullptr && send_count >= 0, "bad argument");
    if (send_count == 0) return DLRM_OK;
    DLRM_REQUIRE(d_send != nullptr && d_recv != nullptr, "NULL buffer");
    DLRM_RCCL(g_rccl.AllGather(d_send, d_recv, (size_t)send_count, ncclFloat, c->comm, as_stream(stream)));
    return DLRM_OK;
}
