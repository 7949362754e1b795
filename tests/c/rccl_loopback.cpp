// rccl_loopback.cpp -- TEST INFRASTRUCTURE: the ten RCCL entry points libparticlefilter.so binds at run time
// (pf_loc.hip, rccl()), implemented over POSIX shared memory with host staging, so that the in-library exchange of a
// sharded filter (pfx_loc_shard_deposit_range: group of ncclSend/ncclRecv to the neighbours, ncclAllGather of the
// chunk-sum rows, ncclAllReduce of the pose sums) can run with MORE THAN ONE RANK on a box with one GPU -- real RCCL
// refuses two ranks on one device.  Selected by PFX_RCCL_LIB=<this library>; never loaded otherwise, never shipped as
// part of the product.  Ranks are processes on one host; every call is host-synchronous (it drains the stream it was
// given first), which is a stricter ordering than RCCL's stream order, so whatever is correct here in terms of buffers,
// offsets, counts and who-talks-to-whom is what the real library is asked to do.
//
// Pairing rules are RCCL's: a collective is entered by every rank; a send is matched by exactly one receive of the same
// size on the peer; ranks with nothing to exchange do not enter the group at all (so nothing here may wait for "all").
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 1 << 20;              // per rank, collectives (chunk-sum rows: <= 64 KB; pose sums: 48 B)
constexpr size_t kBoxBytes = (size_t)1 << 30;             // per ordered pair, point-to-point (16 B per migrating offspring); the
                                                    // segment is sparse: only what a test really sends is ever touched
constexpr uint64_t kMagic = 0x6C6F6F706261636Bull;  // "loopback"
constexpr double kTimeoutS = 60.0;

struct Box {                    // one message in flight per ordered pair
    std::atomic<uint64_t> written, read;
    uint64_t bytes;
    char pad[40];
};
struct Shared {
    std::atomic<uint64_t> magic;
    std::atomic<int> arrived;
    std::atomic<int> generation;
    int nranks;
    char pad[44];
    Box box[kMaxRanks][kMaxRanks];
    // then: kMaxRanks slots of kSlotBytes, then kMaxRanks * kMaxRanks mailboxes of kBoxBytes
};
size_t total_bytes() { return sizeof(Shared) + kMaxRanks * kSlotBytes + (size_t)kMaxRanks * kMaxRanks * kBoxBytes; }

struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

[[noreturn]] void die(const char* what)
{
    fprintf(stderr, "rccl_loopback: %s\n", what);
    abort();
}

template <typename Pred> void wait_for(Pred p, const char* what)
{
    const double t0 = now();
    while (!p()) {
        sched_yield();
        if (now() - t0 > kTimeoutS) die(what);
    }
}

size_t size_of(ncclDataType_t t)
{
    switch (t) {
        case ncclFloat: return 4;
        case ncclDouble: return 8;
        default: die("data type not used by libparticlefilter");
    }
}

thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

}  // namespace

struct ncclComm {
    Shared* sh = nullptr;
    char name[64] = {};
    int rank = 0, nranks = 0;
    char* slot(int r) const { return reinterpret_cast<char*>(sh) + sizeof(Shared) + (size_t)r * kSlotBytes; }
    char* mail(int src, int dst) const
    {
        return reinterpret_cast<char*>(sh) + sizeof(Shared) + kMaxRanks * kSlotBytes + ((size_t)src * kMaxRanks + dst) * kBoxBytes;
    }
    void barrier() const
    {
        const int gen = sh->generation.load(std::memory_order_acquire);
        if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
            sh->arrived.store(0, std::memory_order_relaxed);
            sh->generation.store(gen + 1, std::memory_order_release);
        } else {
            wait_for([&] { return sh->generation.load(std::memory_order_acquire) != gen; }, "a rank did not reach a collective within 60 s");
        }
    }
};

namespace {

thread_local ncclComm* g_groupComm = nullptr;        // libparticlefilter uses one communicator per filter and group

ncclResult_t flush()
{
    ncclComm* c = g_groupComm;
    std::vector<Op> ops;
    ops.swap(g_ops);
    g_groupComm = nullptr;
    if (ops.empty()) return ncclSuccess;
    for (const Op& o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (const Op& o : ops) {
        if (!o.send) continue;
        if (o.bytes > kBoxBytes) die("message larger than the loopback mailbox");
        Box& b = c->sh->box[c->rank][o.peer];
        wait_for([&] { return b.read.load(std::memory_order_acquire) == b.written.load(std::memory_order_relaxed); },
                 "a peer did not take the previous message within 60 s");
        if (hipMemcpy(c->mail(c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        b.bytes = o.bytes;
        b.written.fetch_add(1, std::memory_order_release);
    }
    for (const Op& o : ops) {
        if (o.send) continue;
        Box& b = c->sh->box[o.peer][c->rank];
        wait_for([&] { return b.written.load(std::memory_order_acquire) > b.read.load(std::memory_order_relaxed); },
                 "a receive was not matched by a send within 60 s (the exchange plans of two ranks disagree)");
        if (b.bytes != o.bytes) {
            fprintf(stderr, "rccl_loopback: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, o.bytes, o.peer,
                    (unsigned long long)b.bytes);
            abort();
        }
        if (hipMemcpy(o.buf, c->mail(o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        b.read.fetch_add(1, std::memory_order_release);
    }
    return ncclSuccess;
}

ncclResult_t enqueue(ncclComm* c, const Op& o)
{
    if (o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank) die("send/recv peer out of range");
    if (g_groupComm && g_groupComm != c) die("one communicator per group");
    g_groupComm = c;
    g_ops.push_back(o);
    return g_group ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id->internal, 0, sizeof id->internal);
    const unsigned long long a = (unsigned long long)getpid(), b = (unsigned long long)(now() * 1e6);
    snprintf(id->internal, sizeof id->internal, "/pfx_rccl_loopback_%llx_%llx", a, b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->nranks = nranks;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total_bytes()) != 0) die("cannot create the shared segment");
    } else {
        wait_for([&] { fd = shm_open(c->name, O_RDWR, 0600); return fd >= 0; }, "rank 0 did not create the shared segment within 60 s");
        wait_for([&] { struct stat st; return fstat(fd, &st) == 0 && (size_t)st.st_size == total_bytes(); }, "shared segment not sized");
    }
    void* p = mmap(nullptr, total_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) die("mmap of the shared segment failed");
    c->sh = static_cast<Shared*>(p);
    if (rank == 0) {
        c->sh->nranks = nranks;
        c->sh->magic.store(kMagic, std::memory_order_release);       // (fresh pages are zero: counters, boxes)
    } else {
        wait_for([&] { return c->sh->magic.load(std::memory_order_acquire) == kMagic; }, "rank 0 did not initialise the shared segment");
    }
    c->barrier();
    if (rank == 0) shm_unlink(c->name);         // everybody has it mapped: the name can go, the memory goes with the last unmap
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    munmap(c->sh, total_bytes());
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_loopback: HIP copy failed or invalid argument"; }

ncclResult_t ncclGroupStart() { ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (g_group <= 0) return ncclInvalidUsage;
    return --g_group == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return enqueue(c, Op{ true, const_cast<void*>(buf), count * size_of(t), peer, s });
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return enqueue(c, Op{ false, buf, count * size_of(t), peer, s });
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    const size_t bytes = count * size_of(t);
    if (bytes > kSlotBytes) die("all-gather row larger than the loopback slot");
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->barrier();
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy(static_cast<char*>(recv) + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    c->barrier();
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (t != ncclDouble || op != ncclSum) die("all-reduce: only sums of doubles are used by libparticlefilter");
    const size_t bytes = count * sizeof(double);
    if (bytes > kSlotBytes) die("all-reduce vector larger than the loopback slot");
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->barrier();
    std::vector<double> acc(count, 0.0);
    for (int r = 0; r < c->nranks; ++r) {               // rank order on every rank: the same sum everywhere
        const double* p = reinterpret_cast<const double*>(c->slot(r));
        for (size_t i = 0; i < count; ++i) acc[i] += p[i];
    }
    c->barrier();
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
