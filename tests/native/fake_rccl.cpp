// fake_rccl.cpp -- TEST DOUBLE for librccl (tests/test_gpu_exchange.py).  Not part of the product.
//
// The pool gives one GPU per box and RCCL refuses two ranks on one device, so the exchange half of
// the executor (rf_graph.cpp: exchange_rows on the comm stream, the interior/boundary split, the
// event edges) would never run before the driver's 8-GPU bench.  This library implements the eight
// RCCL entry points librfhip dlopens -- with the real signatures -- over POSIX shared memory, so
// several PROCESSES sharing GPU 0 can run the product's exchange-mode code unchanged:
//
//   ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd are queued; ncclGroupEnd waits for the
//   stream (everything the product ordered before the exchange), copies every send device -> mailbox,
//   then every receive mailbox -> device, synchronously.  One mailbox per ordered (src, dst) pair,
//   one message in flight per mailbox (sequence numbers), so neighbours cannot deadlock: all sends
//   are posted before any receive is waited for.
//
// It says nothing about RCCL itself (rf_comm_selftest covers the real library's ABI on one rank).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rf_rccl_abi.h"   // the function types the product calls through (checked against the real rccl.h by rccl_abi_check.cpp)

namespace {

constexpr size_t kBoxBytes = 8u << 20;   // payload capacity of one mailbox
constexpr int kMaxRanks = 8;

struct Mailbox {
    std::atomic<uint64_t> written;   // messages published
    std::atomic<uint64_t> read;      // messages consumed
    uint64_t size;
    char data[kBoxBytes];
};

struct Shared {
    std::atomic<int> attached;
    Mailbox box[kMaxRanks][kMaxRanks];   // [src][dst]
};

struct Comm {
    Shared* sh = nullptr;
    int rank = 0, world = 1;
    std::string name;
};

typedef rf::NcclId Id;   // the product's own view of ncclUniqueId

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int run_ops()
{
    // everything the caller enqueued before the exchange must have happened
    for (const Op& o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    for (const Op& o : g_ops) {
        if (!o.send) continue;
        if (o.bytes > kBoxBytes) return 2;
        Mailbox& m = o.comm->sh->box[o.comm->rank][o.peer];
        while (m.written.load(std::memory_order_acquire) != m.read.load(std::memory_order_acquire)) usleep(50);
        if (hipMemcpy(m.data, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        m.size = o.bytes;
        m.written.fetch_add(1, std::memory_order_release);
    }
    for (const Op& o : g_ops) {
        if (o.send) continue;
        Mailbox& m = o.comm->sh->box[o.peer][o.comm->rank];
        while (m.written.load(std::memory_order_acquire) == m.read.load(std::memory_order_acquire)) usleep(50);
        if (m.size != o.bytes) return 3;
        if (hipMemcpy(o.buf, m.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
        m.read.fetch_add(1, std::memory_order_release);
    }
    g_ops.clear();
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(Id* id)
{
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/rf_fake_rccl_%d_%ld", (int)getpid(), (long)random());
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    id.internal[sizeof(id.internal) - 1] = 0;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 5;
    if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); return 5; }   // new pages read as zero: every counter starts at 0
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 5;
    Comm* c = new Comm();
    c->sh = static_cast<Shared*>(p);
    c->rank = rank;
    c->world = nranks;
    c->name = id.internal;
    c->sh->attached.fetch_add(1);
    while (c->sh->attached.load() < nranks) usleep(100);   // the real call is collective too
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}

int ncclGroupStart()
{
    ++g_depth;
    return 0;
}

int ncclGroupEnd()
{
    if (--g_depth > 0) return 0;
    // FAKE_RCCL_FAIL=1: every exchange fails (bench.py's probe-and-agree path, tests/test_gpu_exchange.py)
    if (const char* e = std::getenv("FAKE_RCCL_FAIL")) {
        if (*e == '1') {
            g_ops.clear();
            return 5;
        }
    }
    return run_ops();
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (dtype != 0) return 4;   // the product only moves bytes (ncclInt8)
    g_ops.push_back({true, const_cast<void*>(buf), count, peer, static_cast<Comm*>(comm), stream});
    return g_depth > 0 ? 0 : run_ops();
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (dtype != 0) return 4;
    g_ops.push_back({false, buf, count, peer, static_cast<Comm*>(comm), stream});
    return g_depth > 0 ? 0 : run_ops();
}

const char* ncclGetErrorString(int r)
{
    switch (r) {
        case 0: return "success";
        case 1: return "fake rccl: HIP call failed";
        case 2: return "fake rccl: message larger than a mailbox";
        case 3: return "fake rccl: size mismatch between send and recv";
        case 4: return "fake rccl: invalid argument";
        default: return "fake rccl: shared memory";
    }
}

}  // extern "C"

// the double implements exactly the types the product binds
static_assert(std::is_same<decltype(&ncclGetUniqueId), rf::NcclGetUniqueIdFn>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), rf::NcclCommInitRankFn>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), rf::NcclCommDestroyFn>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclSend), rf::NcclSendFn>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), rf::NcclRecvFn>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclGroupStart), rf::NcclGroupFn>::value, "ncclGroupStart");
static_assert(std::is_same<decltype(&ncclGroupEnd), rf::NcclGroupFn>::value, "ncclGroupEnd");
static_assert(std::is_same<decltype(&ncclGetErrorString), rf::NcclGetErrorStringFn>::value, "ncclGetErrorString");
