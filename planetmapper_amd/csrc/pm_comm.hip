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

#include "pm_host.hip.h"

namespace {

// the part of rccl.h this file needs (RCCL 2.x ABI)
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId {
    char internal[128];
};
enum { ncclSuccess = 0, ncclFloat64 = 8 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
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
        r.GetUniqueId = (int (*)(ncclUniqueId *))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
        r.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
        r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    });
    return &r;
}

}  // namespace

struct pm_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    pm_ctx *ctx = nullptr;
};

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
    if (comm->comm && r->CommDestroy) (void)r->CommDestroy(comm->comm);
    delete comm;
    return PM_OK;
}

int pm_map_cube_sharded(pm_ctx *ctx, pm_comm *comm, const void *local_cube, int dtype, int n_planes_total,
                        const double *x_map, const double *y_map, int n0, int n1, int interpolation, int propagate_nan,
                        double *out_all, int mem, int gather)
{
    if (!ctx || !out_all) return PM_ERR_INVALID_ARGUMENT;
    if (mem != PM_MEM_DEVICE && mem != PM_MEM_HOST_CUBE)
        return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "pm_map_cube_sharded takes PM_MEM_DEVICE or PM_MEM_HOST_CUBE");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (comm && comm->ctx != ctx) return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "communicator belongs to another context");
    int a = 0, b = 0, per_rank = 0;
    if (pm_shard_bounds(n_planes_total, world, rank, &a, &b, &per_rank) != PM_OK || n0 < 0 || n1 < 0)
        return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "invalid shard request");
    const size_t nmap = (size_t)n0 * n1;
    if (n_planes_total == 0 || nmap == 0) return PM_OK;
    double *mine = out_all + (size_t)rank * per_rank * nmap;
    if (b > a) {
        if (!local_cube) return pmh::fail(ctx, PM_ERR_INVALID_ARGUMENT, "local_cube is NULL but this rank owns planes");
        int rc = pm_map_cube(ctx, local_cube, dtype, b - a, x_map, y_map, n0, n1, interpolation, propagate_nan, mine, mem);
        if (rc != PM_OK) return rc;
    }
    // planes of a short last block: NaN padding, so that the gathered buffer is defined everywhere
    if (b - a < per_rank) {
        const size_t pad = (size_t)(per_rank - (b - a)) * nmap;
        std::vector<double> nanrow(std::min<size_t>(pad, nmap), std::nan(""));
        for (size_t off = 0; off < pad; off += nanrow.size())
            PM_HIP(ctx, hipMemcpyAsync(mine + (size_t)(b - a) * nmap + off, nanrow.data(),
                                       std::min(nanrow.size(), pad - off) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // the mapped planes are final only after the flag check / nanmedian replay: finish before peers see them
    int rc = pm_synchronize(ctx);
    if (rc != PM_OK) return rc;
    if (!gather || world == 1 || !comm) return PM_OK;
    Rccl *r = rccl();
    const int nrc = r->AllGather(mine, out_all, (size_t)per_rank * nmap, ncclFloat64, comm->comm, ctx->stream);
    if (nrc != ncclSuccess)
        return pmh::fail(ctx, PM_ERR_HIP, "ncclAllGather failed: %s", r->GetErrorString ? r->GetErrorString(nrc) : "?");
    return PM_OK;
}

}  // extern "C"
