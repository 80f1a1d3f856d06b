// mock_rccl.cpp — a TEST DOUBLE for the nine RCCL entry points tb_comm.cpp binds (TB_RCCL_LIBRARY points libtbhip at it), so that the C ABI's multi-rank
// path — communicator id carried by the host, tb_comm_create at world size > 1, grouped send / receive with real neighbour lists, all-reduces — runs with
// several processes on ONE GPU, where RCCL itself refuses ("Duplicate GPU detected").  Not a communication library: messages are staged through a POSIX
// shared-memory segment named by the 128-byte id, every call synchronises its stream and blocks on the host.  What it checks is OUR side of the calls
// (peers, counts, pointers, grouping, order); RCCL's own behaviour stays untested until a box has two GPUs.  Built by tests/test_rccl_world1.py with hipcc.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr size_t MAX_RANKS = 8, SLOT_BYTES = 1u << 20; // one message per (source, destination) pair in flight, ≤ 1 MiB (tests: interface planes of small boxes)

struct Slot {
    std::atomic<unsigned long long> posted, consumed;
    size_t bytes;
    unsigned char data[SLOT_BYTES];
};
struct Shared {
    std::atomic<int> barrier_count, barrier_sense;
    Slot box[MAX_RANKS][MAX_RANKS];  // [source][destination]
    unsigned char red[MAX_RANKS][SLOT_BYTES];
};

struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
    Shared *sh = nullptr;
    int rank = 0, size = 1, sense = 0;
    char name[64] = {0};
};

thread_local int group_depth = 0;
thread_local std::vector<Op> pending;
thread_local Comm *pending_comm = nullptr;

void barrier(Comm *c)
{
    c->sense ^= 1;
    if (c->sh->barrier_count.fetch_add(1) + 1 == c->size) {
        c->sh->barrier_count.store(0);
        c->sh->barrier_sense.store(c->sense);
    } else {
        while (c->sh->barrier_sense.load() != c->sense) usleep(50);
    }
}

ncclResult_t run(Comm *c, std::vector<Op> &ops)
{
    for (const Op &o : ops) // every send of the group first (buffered), then the receives: the neighbour does the same, nobody waits for an unposted message
        if (o.send) {
            if (o.bytes > SLOT_BYTES || o.peer < 0 || o.peer >= c->size) return ncclInvalidArgument;
            Slot &s = c->sh->box[c->rank][o.peer];
            while (s.posted.load() != s.consumed.load()) usleep(20);
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpy(s.data, o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            s.bytes = o.bytes;
            s.posted.fetch_add(1);
        }
    for (const Op &o : ops)
        if (!o.send) {
            if (o.peer < 0 || o.peer >= c->size) return ncclInvalidArgument;
            Slot &s = c->sh->box[o.peer][c->rank];
            while (s.posted.load() == s.consumed.load()) usleep(20);
            if (s.bytes != o.bytes) { fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from %d, got %zu\n", c->rank, o.bytes, o.peer, s.bytes); return ncclInvalidUsage; }
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpy(o.dst, s.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            s.consumed.fetch_add(1);
        }
    return ncclSuccess;
}

ncclResult_t submit(Comm *c, const Op &o)
{
    if (group_depth > 0) { pending.push_back(o); pending_comm = c; return ncclSuccess; }
    std::vector<Op> one(1, o);
    return run(c, one);
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tbmockrccl_%d_%ld", (int)getpid(), (long)random());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > (int)MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->rank = rank; c->size = nranks;
    strncpy(c->name, id.internal, sizeof c->name - 1);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    c->sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // (a fresh segment is zero-filled: counters start at 0)
    close(fd);
    if (c->sh == MAP_FAILED) { delete c; return ncclSystemError; }
    barrier(c); // like ncclCommInitRank: returns when every rank has joined
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (group_depth <= 0) return ncclInvalidUsage;
    if (--group_depth > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    if (!pending.empty()) r = run(pending_comm, pending);
    pending.clear();
    return r;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclDouble) return ncclInvalidArgument;
    return submit((Comm *)comm, Op{true, sendbuff, nullptr, count * sizeof(double), peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclDouble) return ncclInvalidArgument;
    return submit((Comm *)comm, Op{false, nullptr, recvbuff, count * sizeof(double), peer, stream});
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (datatype != ncclDouble || count * sizeof(double) > SLOT_BYTES || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->red[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    std::vector<double> acc(count);
    for (size_t i = 0; i < count; ++i) {
        double v = ((const double *)c->sh->red[0])[i];
        for (int r = 1; r < c->size; ++r) { const double w = ((const double *)c->sh->red[r])[i]; v = op == ncclSum ? v + w : (w > v ? w : v); } // rank order: the same bits on every rank
        acc[i] = v;
    }
    barrier(c); // every rank has read before anyone writes the next contribution
    if (hipMemcpy(recvbuff, acc.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock_rccl error"; }

} // extern "C"
