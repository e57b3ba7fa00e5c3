// loopback_nccl.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// A loopback transport with the eleven RCCL entry points planetmapper_amd/csrc/pm_comm.hip binds at run
// time (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommAbort, ncclAllGather, ncclAllReduce,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), selected with
// PM_RCCL_LIBRARY=<this library>. RCCL refuses two ranks on one GPU ("Duplicate GPU detected : rank 0 and
// rank 1 both on CUDA device f1000", profiles/r04_multiproc_probe.jsonl), and the GPU boxes of this pool
// have one GPU: with this library pm_map_cube_sharded's Send / Recv groups, its closing all-reduce and its
// abort path run at world sizes 2 ... 8 on ONE card - ranks as threads of one process or as processes.
//
// Transport: one POSIX shared-memory segment per communicator (named by the unique id): a single-producer
// single-consumer byte ring per ordered pair of ranks. An operation stages device memory through the host
// (stream synchronise, D2H, ring, H2D) and is complete when the call - or the closing ncclGroupEnd - returns;
// that is a valid (if slow) implementation of the NCCL contract "complete in stream order". Every operation
// of a group makes progress in turn, so a pair of ranks sending to each other cannot deadlock on a full ring.
// A wait that sees no progress for PM_LOOPBACK_TIMEOUT_S seconds (default 60) fails with ncclSystemError;
// ncclCommAbort raises a flag in the segment that fails every rank's pending and future operation (so, unlike
// RCCL's, an abort here DOES release remote peers - unless PM_LOOPBACK_LOCAL_ABORT=1, the mode in which the
// failing-transport test shows what the peers' progress rests on: the transport's own timeout, nothing else;
// tests of "nobody is left waiting" must not rely on the flag
// and assert instead that the protocol itself keeps every rank moving).
//
// Fault injection (tests): PM_LOOPBACK_FAIL="<rank>:<nth>" makes the nth ncclSend of that rank (counted per
// communicator, from 1) return ncclInternalError before anything is sent.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

enum : int { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
             ncclRemoteError = 6 };

constexpr uint32_t kMagic = 0x706d6c62;  // "pmlb"
constexpr size_t kRingBytes = (size_t)256 << 10;

struct Ring {
    alignas(64) std::atomic<uint64_t> head;  // bytes written by the producer
    alignas(64) std::atomic<uint64_t> tail;  // bytes read by the consumer
    alignas(64) unsigned char data[kRingBytes];
};

struct Segment {
    std::atomic<uint32_t> magic;
    int world;
    std::atomic<int> arrived, aborted, detached;
    alignas(64) Ring rings[1];  // [src * world + dst]
};

size_t segment_bytes(int world) { return offsetof(Segment, rings) + (size_t)world * world * sizeof(Ring); }

struct Comm {
    Segment *seg = nullptr;
    size_t bytes = 0;
    int world = 1, rank = 0;
    std::string name;
    long sends = 0;
    int fail_rank = -1;
    long fail_nth = 0;
};

struct Op {
    enum Kind { kSend, kRecv } kind;
    Comm *comm;
    int peer;
    void *dev;              // device (or host) buffer of the caller
    size_t bytes, moved = 0;
    std::vector<unsigned char> host;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local int g_group_rc = ncclSuccess;

std::atomic<long> g_stat_groups{0}, g_stat_sends{0}, g_stat_recvs{0}, g_stat_bytes{0}, g_stat_allreduce{0}, g_stat_aborts{0};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double timeout_s()
{
    static const double t = [] {
        const char *e = std::getenv("PM_LOOPBACK_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        return v > 0.0 ? v : 60.0;
    }();
    return t;
}

// PM_LOOPBACK_HOST_ONLY=1: buffers are host memory and no HIP call is made (the CPU tests of the transport)
bool host_only()
{
    static const bool v = [] {
        const char *e = std::getenv("PM_LOOPBACK_HOST_ONLY");
        return e && e[0] == '1';
    }();
    return v;
}

bool dev_sync(hipStream_t s) { return host_only() || hipStreamSynchronize(s) == hipSuccess; }

bool dev_copy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return true;
    if (host_only()) {
        std::memcpy(dst, src, bytes);
        return true;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s) == hipSuccess;
}

size_t dtype_size(int dt)
{
    switch (dt) {
        case 0: case 1: return 1;          // int8 / uint8
        case 2: case 3: case 7: return 4;  // int32 / uint32 / float32
        case 4: case 5: case 8: return 8;  // int64 / uint64 / float64
        case 6: case 9: return 2;          // float16 / bfloat16
        default: return 0;
    }
}

// every operation's host buffer through its ring (send: op.host -> ring of (rank, peer); receive: ring of (peer, rank) ->
// op.host), all of them in turn so that two ranks sending to each other cannot block on a full ring
int move_through_rings(std::vector<Op> &ops)
{
    double last_progress = now_s();
    size_t open = ops.size();
    for (Op &op : ops)
        if (op.bytes == 0) open--;
    while (open > 0) {
        bool progressed = false;
        for (Op &op : ops) {
            if (op.moved == op.bytes) continue;
            Comm *c = op.comm;
            if (c->seg->aborted.load(std::memory_order_acquire)) return ncclRemoteError;
            if (op.kind == Op::kSend) {
                Ring &r = c->seg->rings[(size_t)c->rank * c->world + op.peer];
                const uint64_t head = r.head.load(std::memory_order_relaxed), tail = r.tail.load(std::memory_order_acquire);
                size_t room = kRingBytes - (size_t)(head - tail);
                size_t n = std::min(room, op.bytes - op.moved);
                if (n == 0) continue;
                const size_t at = (size_t)(head % kRingBytes), first = std::min(n, kRingBytes - at);
                std::memcpy(r.data + at, op.host.data() + op.moved, first);
                std::memcpy(r.data, op.host.data() + op.moved + first, n - first);
                r.head.store(head + n, std::memory_order_release);
                op.moved += n;
                progressed = true;
            } else {
                Ring &r = c->seg->rings[(size_t)op.peer * c->world + c->rank];
                const uint64_t tail = r.tail.load(std::memory_order_relaxed), head = r.head.load(std::memory_order_acquire);
                size_t n = std::min((size_t)(head - tail), op.bytes - op.moved);
                if (n == 0) continue;
                const size_t at = (size_t)(tail % kRingBytes), first = std::min(n, kRingBytes - at);
                std::memcpy(op.host.data() + op.moved, r.data + at, first);
                std::memcpy(op.host.data() + op.moved + first, r.data, n - first);
                r.tail.store(tail + n, std::memory_order_release);
                op.moved += n;
                progressed = true;
            }
            if (op.moved == op.bytes) open--;
        }
        if (progressed) {
            last_progress = now_s();
        } else {
            if (now_s() - last_progress > timeout_s()) {
                std::fprintf(stderr, "[pm loopback] no progress for %.0f s: a peer is not taking part in this operation\n", timeout_s());
                return ncclSystemError;
            }
            std::this_thread::yield();
        }
    }
    return ncclSuccess;
}

// every queued operation of the calling thread, to completion: copy out, rings, copy in
int run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    // what the caller's streams have produced so far is what gets sent
    std::vector<hipStream_t> streams;
    for (Op &op : ops) {
        bool seen = false;
        for (hipStream_t s : streams) seen = seen || s == op.stream;
        if (!seen) streams.push_back(op.stream);
    }
    for (hipStream_t s : streams)
        if (!dev_sync(s)) return ncclUnhandledCudaError;
    for (Op &op : ops) {
        op.host.resize(op.bytes);
        if (op.kind == Op::kSend && op.bytes) {
            if (!dev_copy(op.host.data(), op.dev, op.bytes, op.stream)) return ncclUnhandledCudaError;
        }
    }
    for (hipStream_t s : streams)
        if (!dev_sync(s)) return ncclUnhandledCudaError;
    const int mrc = move_through_rings(ops);
    if (mrc != ncclSuccess) return mrc;
    for (Op &op : ops)
        if (op.kind == Op::kRecv && op.bytes) {
            if (!dev_copy(op.dev, op.host.data(), op.bytes, op.stream)) return ncclUnhandledCudaError;
        }
    for (hipStream_t s : streams)
        if (!dev_sync(s)) return ncclUnhandledCudaError;
    return ncclSuccess;
}

int enqueue(Op &&op)
{
    if (!op.comm || !op.comm->seg) return ncclInvalidArgument;
    if (op.peer < 0 || op.peer >= op.comm->world || op.peer == op.comm->rank) return ncclInvalidArgument;
    if (g_depth > 0) {
        g_ops.push_back(std::move(op));
        return ncclSuccess;
    }
    std::vector<Op> one;
    one.push_back(std::move(op));
    return run_ops(one);
}

// every rank sends `bytes` to every peer and receives as much from each (the data plane of all-gather and
// all-reduce); recv[p] holds what peer p sent
int exchange_all(Comm *c, const void *send_dev, size_t bytes, std::vector<std::vector<unsigned char>> &recv, std::vector<unsigned char> &own,
                 hipStream_t stream)
{
    if (!dev_sync(stream)) return ncclUnhandledCudaError;
    own.resize(bytes);
    if (!dev_copy(own.data(), send_dev, bytes, stream)) return ncclUnhandledCudaError;
    if (!dev_sync(stream)) return ncclUnhandledCudaError;
    std::vector<Op> ops;
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) continue;
        Op s{Op::kSend, c, p, nullptr, bytes, 0, {}, stream};
        s.host = own;
        Op r{Op::kRecv, c, p, nullptr, bytes, 0, {}, stream};
        r.host.resize(bytes);
        ops.push_back(std::move(s));
        ops.push_back(std::move(r));
    }
    // (host-to-host: the ring loop of run_ops without its device copies)
    const int mrc = move_through_rings(ops);
    if (mrc != ncclSuccess) return mrc;
    recv.assign((size_t)c->world, {});
    for (Op &op : ops)
        if (op.kind == Op::kRecv) recv[(size_t)op.peer] = std::move(op.host);
    return ncclSuccess;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};
typedef Comm *ncclComm_t;

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{0};
    std::memset(id->internal, 0, sizeof(id->internal));
    const unsigned long long stamp = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof(id->internal), "/pmlb_%d_%u_%llx", (int)getpid(), counter.fetch_add(1), stamp);
    return ncclSuccess;
}

int ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (!out || world <= 0 || rank < 0 || rank >= world) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;
    Comm *c = new Comm();
    c->world = world;
    c->rank = rank;
    c->name = id.internal;
    c->bytes = segment_bytes(world);
    if (const char *f = std::getenv("PM_LOOPBACK_FAIL")) {
        int r = -1;
        long n = 0;
        if (std::sscanf(f, "%d:%ld", &r, &n) == 2) {
            c->fail_rank = r;
            c->fail_nth = n;
        }
    }
    // whoever comes first creates and sizes the segment; the others wait for its magic number
    bool creator = true;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
        creator = false;
        const double t0 = now_s();
        while ((fd = shm_open(c->name.c_str(), O_RDWR, 0600)) < 0) {
            if (now_s() - t0 > timeout_s()) {
                delete c;
                return ncclSystemError;
            }
            std::this_thread::yield();
        }
    } else if (ftruncate(fd, (off_t)c->bytes) != 0) {
        close(fd);
        shm_unlink(c->name.c_str());
        delete c;
        return ncclSystemError;
    }
    if (!creator) {
        // (the creator may not have sized it yet)
        const double t0 = now_s();
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->bytes) {
            if (now_s() - t0 > timeout_s()) {
                close(fd);
                delete c;
                return ncclSystemError;
            }
            std::this_thread::yield();
        }
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        if (creator) shm_unlink(c->name.c_str());
        delete c;
        return ncclSystemError;
    }
    c->seg = (Segment *)p;
    if (creator) {
        // (a fresh segment is zero-filled: rings, counters and flags start at 0)
        c->seg->world = world;
        c->seg->magic.store(kMagic, std::memory_order_release);
    }
    const double t0 = now_s();
    while (c->seg->magic.load(std::memory_order_acquire) != kMagic) {
        if (now_s() - t0 > timeout_s()) {
            munmap(p, c->bytes);
            delete c;
            return ncclSystemError;
        }
        std::this_thread::yield();
    }
    if (c->seg->world != world) {
        munmap(p, c->bytes);
        delete c;
        return ncclInvalidArgument;
    }
    // the call is collective: it returns once every rank is here (then the name is no longer needed)
    const int n = c->seg->arrived.fetch_add(1, std::memory_order_acq_rel) + 1;
    while (c->seg->arrived.load(std::memory_order_acquire) < world) {
        if (c->seg->aborted.load(std::memory_order_acquire) || now_s() - t0 > timeout_s()) {
            munmap(p, c->bytes);
            delete c;
            return ncclSystemError;
        }
        std::this_thread::yield();
    }
    if (n == world) shm_unlink(c->name.c_str());
    *out = c;
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->seg) {
        c->seg->detached.fetch_add(1, std::memory_order_acq_rel);
        munmap((void *)c->seg, c->bytes);
    }
    delete c;
    return ncclSuccess;
}

int ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    g_stat_aborts.fetch_add(1);
    // PM_LOOPBACK_LOCAL_ABORT=1: like RCCL's, the abort releases nobody but the caller - peers that wait for this rank go on
    // waiting until the transport's own timeout (PM_LOOPBACK_TIMEOUT_S) fails their operation
    const char *local = std::getenv("PM_LOOPBACK_LOCAL_ABORT");
    if (c->seg && !(local && local[0] == '1')) c->seg->aborted.store(1, std::memory_order_release);
    return ncclCommDestroy(c);
}

int ncclGroupStart()
{
    if (g_depth++ == 0) {
        g_ops.clear();
        g_group_rc = ncclSuccess;
    }
    return ncclSuccess;
}

int ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    g_stat_groups.fetch_add(1);
    int rc = g_group_rc;
    if (rc == ncclSuccess) rc = run_ops(g_ops);
    g_ops.clear();
    return rc;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t stream)
{
    const size_t esz = dtype_size(dtype);
    if (!c || esz == 0 || (!buf && count)) return ncclInvalidArgument;
    c->sends++;
    if (c->rank == c->fail_rank && c->sends == c->fail_nth) {
        if (g_depth > 0) g_group_rc = ncclInternalError;
        return ncclInternalError;
    }
    g_stat_sends.fetch_add(1);
    g_stat_bytes.fetch_add((long)(count * esz));
    return enqueue(Op{Op::kSend, c, peer, const_cast<void *>(buf), count * esz, 0, {}, stream});
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t stream)
{
    const size_t esz = dtype_size(dtype);
    if (!c || esz == 0 || (!buf && count)) return ncclInvalidArgument;
    g_stat_recvs.fetch_add(1);
    return enqueue(Op{Op::kRecv, c, peer, buf, count * esz, 0, {}, stream});
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t c, hipStream_t stream)
{
    const size_t esz = dtype_size(dtype);
    if (!c || !c->seg || esz == 0 || g_depth > 0) return ncclInvalidArgument;
    const size_t bytes = count * esz;
    std::vector<std::vector<unsigned char>> got;
    std::vector<unsigned char> own;
    const int rc = exchange_all(c, send, bytes, got, own, stream);
    if (rc != ncclSuccess) return rc;
    for (int p = 0; p < c->world; p++) {
        const unsigned char *src = p == c->rank ? own.data() : got[(size_t)p].data();
        if (!dev_copy((char *)recv + (size_t)p * bytes, src, bytes, stream)) return ncclUnhandledCudaError;
    }
    return dev_sync(stream) ? ncclSuccess : ncclUnhandledCudaError;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t stream)
{
    // sum of int32 / int64 / float32 / float64, added in rank order on every rank (the same bits everywhere)
    const size_t esz = dtype_size(dtype);
    if (!c || !c->seg || esz == 0 || op != 0 || g_depth > 0) return ncclInvalidArgument;
    if (dtype != 2 && dtype != 4 && dtype != 7 && dtype != 8) return ncclInvalidArgument;
    g_stat_allreduce.fetch_add(1);
    const size_t bytes = count * esz;
    std::vector<std::vector<unsigned char>> got;
    std::vector<unsigned char> own;
    const int rc = exchange_all(c, send, bytes, got, own, stream);
    if (rc != ncclSuccess) return rc;
    std::vector<unsigned char> sum(bytes, 0);
    for (int p = 0; p < c->world; p++) {
        const unsigned char *src = p == c->rank ? own.data() : got[(size_t)p].data();
        for (size_t i = 0; i < count; i++) {
            if (dtype == 2) ((int32_t *)sum.data())[i] += ((const int32_t *)src)[i];
            if (dtype == 4) ((int64_t *)sum.data())[i] += ((const int64_t *)src)[i];
            if (dtype == 7) ((float *)sum.data())[i] += ((const float *)src)[i];
            if (dtype == 8) ((double *)sum.data())[i] += ((const double *)src)[i];
        }
    }
    if (!dev_copy(recv, sum.data(), bytes, stream)) return ncclUnhandledCudaError;
    return dev_sync(stream) ? ncclSuccess : ncclUnhandledCudaError;
}

const char *ncclGetErrorString(int rc)
{
    switch (rc) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loopback: HIP call failed";
        case ncclSystemError: return "loopback: timed out waiting for a peer";
        case ncclInternalError: return "loopback: injected failure";
        case ncclInvalidArgument: return "loopback: invalid argument";
        case ncclInvalidUsage: return "loopback: invalid usage";
        case ncclRemoteError: return "loopback: the communicator was aborted";
        default: return "loopback: unknown error";
    }
}

// what this process has moved through the loopback (tests assert that the exchange really ran):
// [groups, sends, recvs, bytes sent, all-reduces, aborts]
void pm_loopback_stats(long out[6])
{
    out[0] = g_stat_groups.load();
    out[1] = g_stat_sends.load();
    out[2] = g_stat_recvs.load();
    out[3] = g_stat_bytes.load();
    out[4] = g_stat_allreduce.load();
    out[5] = g_stat_aborts.load();
}

}  // extern "C"
