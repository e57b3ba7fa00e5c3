// pm_comm.hip -- plane-sharded cube mapping behind the C ABI: pm_comm_* and pm_map_cube_sharded.
//
// The planes of a cube are independent in the reference (observation.py:892-904 maps them one
// after another with the same x/y map), so the path shards with no halo and no reduction: rank r
// maps the contiguous block pm_shard_bounds(P, world, r) on its own GPU and ONE all-gather of the
// mapped planes assembles (P, n0, n1) on every rank. The collective is RCCL's (xGMI between the
// GPUs of a node). RCCL is bound at run time (dlopen of librccl.so.1, the soname both ROCm and
// PyTorch-ROCm ship): a process that already holds it through torch.distributed shares that copy,
// and libplanetmapper_hip.so itself carries no link-time dependency on it - single-GPU users never
// load it.
#include <dlfcn.h>

#include <mutex>

#include <chrono>

#include "pm_host.hip.h"
#include "pm_rccl_abi.h"

namespace {

// the part of rccl.h this file needs: pm_rccl_abi.h (checked against the installed header by tests/rccl_abi_check.cpp)
using ncclComm_t = pm_rccl::Comm;
using ncclUniqueId = pm_rccl::UniqueId;
enum { ncclSuccess = pm_rccl::Success, ncclInt32 = pm_rccl::Int32, ncclFloat64 = pm_rccl::Float64, ncclSum = pm_rccl::Sum };

struct Rccl {
    void *handle = nullptr;
#define PM_RCCL_MEMBER(name, symbol) pm_rccl::name##_t name = nullptr;
    PM_RCCL_SYMBOLS(PM_RCCL_MEMBER)
#undef PM_RCCL_MEMBER
    std::string error;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // PM_RCCL_LIBRARY names the library to bind instead of the default sonames (a site with its own
        // RCCL build; the tests use it to rehearse the machine without RCCL)
        const char *override_name = std::getenv("PM_RCCL_LIBRARY");
        std::string tried;
        for (const char *name : {override_name, override_name ? nullptr : "librccl.so.1", override_name ? nullptr : "librccl.so"}) {
            if (!name) continue;
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            const char *e = dlerror();  // (one call: dlerror() clears the message it returns)
            if (tried.empty()) tried = e ? e : (std::string(name) + " not found");
        }
        if (!r.handle) {
            r.error = "cannot load RCCL: " + (tried.empty() ? std::string("librccl.so.1 not found") : tried);
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n;
            return p;
        };
#define PM_RCCL_BIND(name, symbol) r.name = (pm_rccl::name##_t)sym(symbol);
        PM_RCCL_SYMBOLS(PM_RCCL_BIND)
#undef PM_RCCL_BIND
    });
    return &r;
}

}  // namespace

struct pm_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    pm_ctx *ctx = nullptr;
    // the exchange runs on its own stream, next to the mapping of the following chunk
    hipStream_t stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    int *d_status = nullptr;  // [0..1] this rank's (failed, redone) flags, [2..3] their sums over the ranks
    int *h_status = nullptr;  // pinned mirror
    bool broken = false;      // a collective failed: the communicator has been aborted
};

namespace {

// planes per exchange of the pipelined all-gather: a function of the block size and the map size ONLY,
// so that every rank issues the same sequence of collectives whatever its own block, memory or route
int exchange_planes(int per_rank, size_t nmap)
{
    if (per_rank <= 0) return 1;
    const size_t by_count = ((size_t)per_rank + 7) / 8;                               // at most 8 exchanges
    const size_t by_bytes = (((size_t)4 << 20) + nmap * sizeof(double) - 1) / std::max<size_t>(nmap * sizeof(double), 1);  // of >= 4 MiB
    return (int)std::min<size_t>((size_t)per_rank, std::max<size_t>(std::max(by_count, by_bytes), 1));
}

int nccl_fail(pm_ctx *ctx, pm_comm *comm, const char *what, int nrc)
{
    Rccl *r = rccl();
    // ncclCommAbort frees THIS rank (kernels of the failed collective, the proxy threads); peers blocked in
    // the same collective see it only through RCCL's own error propagation / their watchdog - the protocol
    // of pm_map_cube_sharded therefore keeps every rank moving to the agreement for every failure that is
    // not RCCL's own, and aborts only when the transport itself has failed
    if (comm && comm->comm && r->CommAbort && !comm->broken) {
        (void)r->CommAbort(comm->comm);
        comm->comm = nullptr;
        comm->broken = true;
    }
    return pmh::fail(ctx, PM_ERR_HIP, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(nrc) : "?");
}

}  // namespace

extern "C" {

int pm_shard_bounds(int n_planes, int world, int rank, int *start, int *stop, int *per_rank)
{
    if (n_planes < 0 || world <= 0 || rank < 0 || rank >= world) return PM_ERR_INVALID_ARGUMENT;
    const int pr = n_planes ? (n_planes + world - 1) / world : 0;
    const long a = std::min<long>((long)rank * pr, n_planes);
    const long b = std::min<long>(a + pr, n_planes);
    if (start) *start = (int)a;
    if (stop) *stop = (int)b;
    if (per_rank) *per_rank = pr;
    return PM_OK;
}

int pm_exchange_planes(int per_rank, int n0, int n1)
{
    if (per_rank < 0 || n0 < 0 || n1 < 0) return PM_ERR_INVALID_ARGUMENT;
    return exchange_planes(per_rank, (size_t)n0 * n1);
}

int pm_comm_unique_id(void *id128)
{
    if (!id128) return PM_ERR_INVALID_ARGUMENT;
    Rccl *r = rccl();
    if (!r->error.empty()) return PM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return PM_ERR_HIP;
    std::memcpy(id128, &id, sizeof(id));
    return PM_OK;
}

int pm_comm_create(pm_ctx *ctx, int world, int rank, const void *id128, pm_comm **comm)
{
    if (!ctx || !comm || !id128 || world <= 0 || rank < 0 || rank >= world) return PM_ERR_INVALID_ARGUMENT;
    Rccl *r = rccl();
    if (!r->error.empty()) return pmh::fail(ctx, PM_ERR_UNSUPPORTED, "%s", r->error.c_str());
    PM_HIP(ctx, hipSetDevice(ctx->device));
    pm_comm *c = new (std::nothrow) pm_comm();
    if (!c) return pmh::fail(ctx, PM_ERR_ALLOC, "out of memory");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) {
        delete c;
        return pmh::fail(ctx, PM_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
    }
    c->world = world;
    c->rank = rank;
    c->ctx = ctx;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->d_status, 4 * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void **)&c->h_status, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        pm_comm_destroy(c);
        return pmh::fail(ctx, PM_ERR_HIP, "could not create the exchange stream of the communicator");
    }
    *comm = c;
    return PM_OK;
}

int pm_comm_destroy(pm_comm *comm)
{
    if (!comm) return PM_OK;
    Rccl *r = rccl();
    if (comm->ctx) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
    }
    if (comm->stream) (void)hipStreamSynchronize(comm->stream);
    if (comm->comm && r->CommDestroy) (void)r->CommDestroy(comm->comm);
    if (comm->ev_ready) (void)hipEventDestroy(comm->ev_ready);
    if (comm->ev_done) (void)hipEventDestroy(comm->ev_done);
    if (comm->d_status) (void)hipFree(comm->d_status);
    if (comm->h_status) (void)hipHostFree(comm->h_status);
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    delete comm;
    return PM_OK;
}

}  // extern "C"

namespace {

// the exchanges of one pm_map_cube_sharded call: issued in order, each as soon as the kernels of its
// planes are on the context stream (pm_set_chunk_callback), on the communicator's own stream
struct Exchanger {
    pm_ctx *ctx;
    pm_comm *comm;
    double *out_all, *mine;
    size_t nmap;
    int per_rank, mine_n, step;
    int next_e0 = 0;  // first plane of the next exchange to issue
    int nrc = ncclSuccess;
    hipError_t herr = hipSuccess;

    // one group of sends / receives: planes [e0, e1) of every rank's block, behind the context stream.
    // A HIP failure while ordering the exchange behind the mapping is remembered (this rank then reports
    // itself failed in the agreement) but the group is issued all the same: the peers wait for it.
    void issue(int e0, int e1)
    {
        if (nrc != ncclSuccess) return;  // (the communicator is gone: nccl_fail() aborts it)
        Rccl *r = rccl();
        hipError_t e = hipEventRecord(comm->ev_ready, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(comm->stream, comm->ev_ready, 0);
        if (e != hipSuccess && herr == hipSuccess) herr = e;
        const size_t count = (size_t)(e1 - e0) * nmap;
        nrc = r->GroupStart();
        for (int peer = 0; peer < comm->world && nrc == ncclSuccess; peer++) {
            if (peer == comm->rank) continue;
            nrc = r->Send(mine + (size_t)e0 * nmap, count, ncclFloat64, peer, comm->comm, comm->stream);
            if (nrc == ncclSuccess)
                nrc = r->Recv(out_all + ((size_t)peer * per_rank + e0) * nmap, count, ncclFloat64, peer, comm->comm, comm->stream);
        }
        const int erc = r->GroupEnd();
        if (nrc == ncclSuccess) nrc = erc;
    }
    // planes [0, done) of this rank's block have their kernels enqueued: issue what has become ready
    void progress(int done)
    {
        while (next_e0 < per_rank) {
            const int e1 = std::min(next_e0 + step, per_rank);
            if (std::min(e1, mine_n) > done) break;  // (planes beyond mine_n are padding, written up front)
            issue(next_e0, e1);
            next_e0 = e1;
        }
    }
    static void on_chunk(void *user, int first, int n) { ((Exchanger *)user)->progress(first + n); }
};

}  // namespace

extern "C" {

// The sharded form of Observation._get_mapped_data (observation.py:876-905). Protocol with gather != 0
// and more than one rank - the same sequence of collectives on every rank whatever happens on it:
//   1. the block is cut into exchanges of pm_exchange_planes(per_rank, n0, n1) planes (a function of
//      shapes only);
//   2. ONE pm_map_cube maps the block (its own pipeline - collecting / copying in chunk k + 1 while chunk
//      k is mapped - stays whole); each time the kernels of further planes are on the context stream
//      (pm_set_chunk_callback) the exchanges they complete are issued on the communicator's own stream:
//      one group of ncclSend / ncclRecv with every peer (an all-gather whose pieces land rank-major in
//      out_all), so exchange k crosses xGMI while the planes of exchange k + 1 are still on their way in;
//   3. when the mapping has finished (flag check, nanmedian replay) any exchange not yet issued - a rank
//      without planes, a rank whose mapping FAILED - is issued all the same: nobody is left waiting;
//   4. one 8-byte all-reduce closes the call: (ranks that failed, ranks that had to redo planes with their
//      nanmedian after those planes had been sent). Any failure: every rank returns an error (its own
//      code, PM_ERR_PEER for a failure elsewhere) - the gathered cube is valid on all ranks or on none.
//      Any redo (rare: +-inf pixels): every rank sends its whole block once more.
int pm_map_cube_sharded(pm_ctx *ctx, pm_comm *comm, const void *local_cube, int dtype, int n_planes_total,
                        const double *x_map, const double *y_map, int n0, int n1, int interpolation, int propagate_nan,
                        double *out_all, int mem, int gather)
{
    if (!ctx || !out_all) return PM_ERR_INVALID_ARGUMENT;
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (comm && comm->ctx != ctx) return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "communicator belongs to another context");
    if (comm && comm->broken) return pmh::fail(ctx, PM_ERR_STATE, "the communicator was aborted by an earlier failure");
    const bool exchange = gather && world > 1 && comm;
    // (argument errors every rank sees alike are returned at once; what can differ between ranks goes
    //  through the agreement below)
    if (mem != PM_MEM_DEVICE && mem != PM_MEM_HOST_CUBE && !(mem == PM_MEM_HOST && !gather))
        return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT,
                         "pm_map_cube_sharded takes PM_MEM_DEVICE or PM_MEM_HOST_CUBE (PM_MEM_HOST with gather = 0 only)");
    int a = 0, b = 0, per_rank = 0;
    if (pm_shard_bounds(n_planes_total, world, rank, &a, &b, &per_rank) != PM_OK || n0 < 0 || n1 < 0)
        return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "invalid shard request");
    const size_t nmap = (size_t)n0 * n1;
    if (n_planes_total == 0 || nmap == 0) return PM_OK;
    const int mine_n = b - a;
    if (mem == PM_MEM_HOST) {
        // every rank writes its planes into the caller's (P, n0, n1) host array - e.g. one array in shared
        // memory for all ranks: no collective at all (SURVEY 8e "each rank copies its slice")
        if (mine_n == 0) return PM_OK;
        if (!local_cube) return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "local_cube is NULL but this rank owns planes");
        return pm_map_cube(ctx, local_cube, dtype, mine_n, x_map, y_map, n0, n1, interpolation, propagate_nan,
                           out_all + (size_t)a * nmap, PM_MEM_HOST);
    }
    double *mine = out_all + (size_t)rank * per_rank * nmap;
    Rccl *r = rccl();
    const auto clock_ns = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_sharded = clock_ns();
    // planes of a short last block: NaN padding, so that the gathered buffer is defined everywhere
    int status = PM_OK;
    if (mine_n < per_rank) {
        const size_t pad = (size_t)(per_rank - mine_n) * nmap;
        std::vector<double> nanrow(std::min<size_t>(pad, nmap), std::nan(""));
        hipError_t he = hipSuccess;
        for (size_t off = 0; off < pad && he == hipSuccess; off += nanrow.size())
            he = hipMemcpyAsync(mine + (size_t)mine_n * nmap + off, nanrow.data(), std::min(nanrow.size(), pad - off) * sizeof(double),
                                hipMemcpyHostToDevice, ctx->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
        // (reported through the agreement, like every failure that can differ between ranks)
        if (he != hipSuccess) status = pmh::fail(ctx, PM_ERR_HIP, "writing the padding of a short block failed: %s", hipGetErrorString(he));
    }
    Exchanger ex{ctx, comm, out_all, mine, nmap, per_rank, mine_n, exchange ? exchange_planes(per_rank, nmap) : std::max(per_rank, 1)};
    if (mine_n > 0 && status == PM_OK) {
        if (!local_cube) {
            status = pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "local_cube is NULL but this rank owns planes");
        } else {
            void (*saved_cb)(void *, int, int) = ctx->chunk_cb;
            void *saved_user = ctx->chunk_user;
            if (exchange) {
                ctx->chunk_cb = &Exchanger::on_chunk;
                ctx->chunk_user = &ex;
            }
            status = pm_map_cube(ctx, local_cube, dtype, mine_n, x_map, y_map, n0, n1, interpolation, propagate_nan, mine, mem);
            ctx->chunk_cb = saved_cb;
            ctx->chunk_user = saved_user;
            // the mapped planes are final only after the flag check / nanmedian replay
            if (status == PM_OK) status = pm_synchronize(ctx);
        }
    }
    std::string first_error = status != PM_OK ? ctx->error : std::string();
    if (!exchange) return status;
    // From here to the agreement nothing returns early except on a failure of RCCL itself (which aborts the
    // communicator): a rank that left with a local error would strand its peers in their receives.
    ex.progress(per_rank);  // whatever has not been sent yet (no planes here, or a failed mapping)
    if (ex.nrc != ncclSuccess) return nccl_fail(ctx, comm, "exchange of mapped planes (ncclSend / ncclRecv)", ex.nrc);
    const double t_mapped = clock_ns();
    if (ctx->trace & 4) {
        // PM_OPT_LAST_STAGE_NS: what of the exchanges is still in flight once the mapping is done (waited for here, so that
        // the agreement behind it is timed on its own)
        (void)hipStreamSynchronize(comm->stream);
        ctx->last_stage_ns[pmh::kStageExchangeExposed] = clock_ns() - t_mapped;
    }
    const double t_exchanged = clock_ns();
    if (ex.herr != hipSuccess && status == PM_OK) {
        status = pmh::fail(ctx, PM_ERR_HIP, "ordering the exchange behind the mapping failed: %s", hipGetErrorString(ex.herr));
        first_error = ctx->error;
    }
    // the agreement: (failed ranks, ranks that redid planes after sending them)
    comm->h_status[0] = status != PM_OK ? 1 : 0;
    comm->h_status[1] = (status == PM_OK && mine_n > 0 && ctx->last_redo_planes > 0) ? 1 : 0;
    comm->h_status[2] = comm->h_status[3] = 0;
    hipError_t he = hipMemcpyAsync(comm->d_status, comm->h_status, 2 * sizeof(int), hipMemcpyHostToDevice, comm->stream);
    if (he != hipSuccess) {
        // this rank cannot even say that it failed: tear the communicator down rather than leave the peers in the all-reduce
        (void)nccl_fail(ctx, comm, "the status all-reduce (could not stage this rank's status)", ncclSuccess);
        return pmh::fail(ctx, PM_ERR_HIP, "staging the status of this rank failed: %s", hipGetErrorString(he));
    }
    int nrc = r->AllReduce(comm->d_status, comm->d_status + 2, 2, ncclInt32, ncclSum, comm->comm, comm->stream);
    if (nrc != ncclSuccess) return nccl_fail(ctx, comm, "ncclAllReduce of the ranks' status", nrc);
    he = hipMemcpyAsync(comm->h_status + 2, comm->d_status + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, comm->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(comm->stream);
    if (he != hipSuccess) {
        // A rank that cannot read the verdict cannot know whether a second exchange follows: peers that read "redo" would
        // wait in their receives for this rank's block. RCCL does not release remote peers on a local error - the
        // communicator is torn down instead, as for a failure of the transport itself.
        (void)nccl_fail(ctx, comm, "reading the ranks' verdict", ncclSuccess);
        return pmh::fail(ctx, PM_ERR_HIP, "reading the verdict of the ranks failed: %s", hipGetErrorString(he));
    }
    ctx->last_stage_ns[pmh::kStageAgreement] = clock_ns() - t_exchanged;
    int n_failed = comm->h_status[2];
    const int n_redo = comm->h_status[3];
    if (n_failed == 0 && n_redo > 0) {
        // somebody's planes changed after they had been sent: everybody sends the whole block again
        ex.issue(0, per_rank);
        if (ex.nrc != ncclSuccess) return nccl_fail(ctx, comm, "second exchange of mapped planes", ex.nrc);
    }
    // pm_synchronize() / later work on the context stream waits for the exchange; so does this call
    hipError_t hw = hipEventRecord(comm->ev_done, comm->stream);
    if (hw == hipSuccess) hw = hipStreamWaitEvent(ctx->stream, comm->ev_done, 0);
    if (hw == hipSuccess) hw = hipStreamSynchronize(comm->stream);
    if (!(ctx->trace & 4)) ctx->last_stage_ns[pmh::kStageExchangeExposed] = 0.0;  // (not separated from the agreement without the trace bit)
    ctx->last_stage_ns[pmh::kStageShardedTotal] = clock_ns() - t_sharded;
    if (status != PM_OK) {
        ctx->error = first_error;
        return status;
    }
    if (he != hipSuccess || hw != hipSuccess || ex.herr != hipSuccess)
        return pmh::fail(ctx, PM_ERR_HIP, "the exchange of mapped planes failed on this rank: %s",
                         hipGetErrorString(he != hipSuccess ? he : (hw != hipSuccess ? hw : ex.herr)));
    if (n_failed > 0)
        return pmh::fail(ctx, PM_ERR_PEER, "%d other rank(s) failed to map their planes: the gathered cube is not valid", n_failed);
    return PM_OK;
}

}  // extern "C"
