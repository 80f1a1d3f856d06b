// tb_comm.cpp — RCCL behind the C ABI (round 4).  The reference is shared-memory only (README.md:7); its GPU extension relies on plain indexing of device
// vectors (ext/CuThunderboltExt.jl:126-170).  Under a partition the one data-path exchange is the sum of interface partials with the neighbouring ranks, and
// the Krylov solvers add two scalar all-reduces per iteration: with these entries a host that has no GPU-aware message passing of its own — the
// Julia host the boundary is for — runs N > 1 through the library alone: it only has to carry the 128-byte communicator id from rank 0 to the other ranks.
//
// RCCL is opened at run time (dlopen of the library the process already has — a host framework's copy — or the system one), so libtbhip.so has no link-time
// dependency on it and a single-GPU user never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "tb_internal.h"

using namespace tb;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why; // why RCCL is unavailable: the dlopen error of the last candidate, captured when it happened
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("TB_RCCL_LIBRARY"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            if (const char *e = dlerror()) r.why = e; // read once, here: dlerror() clears itself and belongs to whichever dl* call came last
        }
        if (!r.handle) return;
        r.why = "symbols missing";
#define TB_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
        TB_SYM(GetUniqueId, "ncclGetUniqueId"); TB_SYM(CommInitRank, "ncclCommInitRank"); TB_SYM(CommDestroy, "ncclCommDestroy");
        TB_SYM(Send, "ncclSend"); TB_SYM(Recv, "ncclRecv"); TB_SYM(AllReduce, "ncclAllReduce");
        TB_SYM(GroupStart, "ncclGroupStart"); TB_SYM(GroupEnd, "ncclGroupEnd"); TB_SYM(GetErrorString, "ncclGetErrorString");
#undef TB_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllReduce && r.GroupStart && r.GroupEnd;
    });
    return r;
}

int need_rccl(const char *what)
{
    if (rccl().ok) return TB_OK;
    set_error("%s: RCCL is not available (librccl.so could not be opened: %s; set TB_RCCL_LIBRARY)", what, rccl().why.empty() ? "no candidate library" : rccl().why.c_str());
    return TB_ERR_UNSUPPORTED;
}

} // namespace

struct tb_comm {
    tb_device *dev = nullptr;   // used by the data-path entries only; tb_comm_destroy never touches it (host finalisers run in any order)
    int dev_id = 0;
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    hipStream_t xstream = nullptr; // queue of tb_comm_exchange_begin: the transfer runs beside what the device's stream does until tb_comm_exchange_end
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending = false;
};

#define TB_NCCL(call)                                                                                             \
    do {                                                                                                          \
        const ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) {                                                                                  \
            set_error("%s -> %s (%s:%d)", #call, rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error", __FILE__, __LINE__); \
            return TB_ERR_HIP;                                                                                    \
        }                                                                                                         \
    } while (0)

extern "C" {

int tb_comm_unique_id(void *id128)
{
    TB_REQUIRE(id128, "tb_comm_unique_id: NULL buffer");
    int rc = need_rccl("tb_comm_unique_id");
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == TB_COMM_ID_BYTES, "communicator id size");
    ncclUniqueId id;
    TB_NCCL(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return TB_OK;
}

int tb_comm_create(tb_device *dev, const void *id128, int rank, int world_size, tb_comm **out)
{
    TB_REQUIRE(dev && id128 && out && world_size >= 1 && rank >= 0 && rank < world_size, "tb_comm_create: bad argument");
    int rc = need_rccl("tb_comm_create");
    if (rc) return rc;
    TB_HIP(hipSetDevice(dev->id));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    auto c = new tb_comm();
    c->dev = dev; c->dev_id = dev->id; c->rank = rank; c->size = world_size;
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world_size, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
        delete c;
        return TB_ERR_HIP;
    }
    *out = c;
    return TB_OK;
}

int tb_comm_destroy(tb_comm *c)
{
    if (!c) return TB_OK;
    (void)hipSetDevice(c->dev_id);
    if (c->comm && rccl().ok) {
        (void)hipDeviceSynchronize(); // not the device object's stream: tb_device_destroy may already have run
        (void)rccl().CommDestroy(c->comm);
    }
    for (hipEvent_t e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
    return TB_OK;
}

int tb_comm_rank_size(tb_comm *c, int *rank, int *size)
{
    TB_REQUIRE(c, "tb_comm_rank_size: NULL communicator");
    if (rank) *rank = c->rank;
    if (size) *size = c->size;
    return TB_OK;
}

// grouped send / receive with every listed neighbour on `stream`; an error inside the group still closes it (an open group would silently queue
// every later RCCL call of the thread)
static int exchange_on(tb_comm *c, hipStream_t stream, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv)
{
    TB_NCCL(rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < n_peers && bad == ncclSuccess; ++k) {
        if (counts[k] == 0) continue;
        bad = rccl().Send(d_send[k], (size_t)counts[k], ncclDouble, peers[k], c->comm, stream);
        if (bad == ncclSuccess) bad = rccl().Recv(d_recv[k], (size_t)counts[k], ncclDouble, peers[k], c->comm, stream);
    }
    const ncclResult_t end = rccl().GroupEnd();
    if (bad == ncclSuccess) bad = end;
    if (bad != ncclSuccess) {
        set_error("tb_comm_exchange: ncclSend / ncclRecv / ncclGroupEnd -> %s", rccl().GetErrorString ? rccl().GetErrorString(bad) : "RCCL error");
        return TB_ERR_HIP;
    }
    return TB_OK;
}

static int check_exchange_args(const char *what, tb_comm *c, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv)
{
    TB_REQUIRE(c && n_peers >= 0 && (n_peers == 0 || (peers && counts && d_send && d_recv)), "%s: bad argument", what);
    for (int k = 0; k < n_peers; ++k)
        TB_REQUIRE(peers[k] >= 0 && peers[k] < c->size && counts[k] >= 0 && (counts[k] == 0 || (d_send[k] && d_recv[k])), "%s: bad neighbour %d", what, k);
    return TB_OK;
}

int tb_comm_exchange(tb_comm *c, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv)
{
    int rc = check_exchange_args("tb_comm_exchange", c, n_peers, peers, counts, d_send, d_recv);
    if (rc) return rc;
    if (n_peers == 0) return TB_OK;
    TB_HIP(hipSetDevice(c->dev->id));
    return exchange_on(c, c->dev->stream, n_peers, peers, counts, d_send, d_recv);
}

int tb_comm_exchange_begin(tb_comm *c, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv)
{
    int rc = check_exchange_args("tb_comm_exchange_begin", c, n_peers, peers, counts, d_send, d_recv);
    if (rc) return rc;
    TB_REQUIRE(!c->pending, "tb_comm_exchange_begin: the previous exchange of this communicator was not ended (tb_comm_exchange_end)");
    if (n_peers == 0) return TB_OK;
    TB_HIP(hipSetDevice(c->dev->id));
    if (!c->xstream) {
        TB_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
        for (hipEvent_t &e : c->ev) TB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    TB_HIP(hipEventRecord(c->ev[0], c->dev->stream));      // the send buffers are packed on the device's stream
    TB_HIP(hipStreamWaitEvent(c->xstream, c->ev[0], 0));
    rc = exchange_on(c, c->xstream, n_peers, peers, counts, d_send, d_recv);
    if (rc) return rc;
    TB_HIP(hipEventRecord(c->ev[1], c->xstream));
    c->pending = true;
    return TB_OK;
}

int tb_comm_exchange_end(tb_comm *c)
{
    TB_REQUIRE(c, "tb_comm_exchange_end: NULL communicator");
    if (!c->pending) return TB_OK;
    c->pending = false;
    TB_HIP(hipSetDevice(c->dev->id));
    TB_HIP(hipStreamWaitEvent(c->dev->stream, c->ev[1], 0)); // whatever the device's stream does next sees the received values
    return TB_OK;
}

int tb_comm_allreduce(tb_comm *c, double *d_buf, int64_t n, int op)
{
    TB_REQUIRE(c && n >= 0 && (d_buf || n == 0) && (op == TB_REDUCE_SUM || op == TB_REDUCE_MAX), "tb_comm_allreduce: bad argument");
    if (n == 0) return TB_OK;
    TB_HIP(hipSetDevice(c->dev->id));
    TB_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, op == TB_REDUCE_SUM ? ncclSum : ncclMax, c->comm, c->dev->stream));
    return TB_OK;
}

} // extern "C"
