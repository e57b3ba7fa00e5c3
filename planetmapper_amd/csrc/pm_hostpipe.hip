// pm_hostpipe.hip -- the host <-> HBM leg of PM_MEM_HOST calls.
//
// The reference hands numpy arrays around: Observation._get_mapped_data (observation.py:876-905)
// maps a host-resident cube plane by plane and every get_*_img returns a host array
// (body_xy.py:3166). A drop-in caller therefore sees the PCIe leg, not the kernels - one 4096^2
// frame is 0.2 ms of GPU work and 671 MB of results. This file makes that leg run at the rate of
// the link (measured on the MI355X boxes, tools/probes/hip/probe_host_path.hip: 57 GB/s per direction for
// pinned memory, 49 GB/s each way in duplex):
//
//   * results -> pageable caller memory: DMA into a ring of pinned staging buffers; a retire thread
//     waits for each DMA and posts the copy-out to a pool of CPU threads (a queue of jobs: it does
//     not wait for the copy, so the next DMA's completion is picked up at once) while the next DMAs
//     run (fresh numpy arrays take their page faults on several cores at once: 74 GB/s with 8
//     threads against 25 GB/s through the runtime's own pageable path); results -> pinned caller
//     memory (pm_host_alloc / pm_host_register): one DMA;
//   * large image planes of the disc (NaN outside the radius pre-mask): only bands of rows around that
//     circle are copied, as rectangles, and the pool writes the NaN of everything else (d2h_issue_disc:
//     74 % of the bytes of the headline frame; 12.0 -> 9.9 ms into pinned arrays);
//   * cube planes, whole (PM_OPT_ZERO_COPY 0, and what the library picks for a map fine enough to
//     read most of every plane): chunks of planes through a three-slot device ring - the H2D copy of
//     chunk k + 1, the kernel of chunk k and the D2H copy of finished output overlap on three
//     streams (4.3 GB cube: 89-94 ms, 46-48 GB/s sustained);
//   * cube planes, sparse (the default for a coarse map - config 5's 1 deg map reads 14 % of the
//     16-byte blocks of each plane): k_mark_blocks runs the sampling code once and flags the 16-byte blocks of a plane it
//     loads from - the same in every plane; the copy threads collect those blocks of each chunk of
//     planes into pinned staging (prefetching by hand: scattered reads defeat the hardware
//     prefetchers) while the DMA of the previous chunk's table runs, and k_reproject_blocks samples
//     the table: 0.59 GB cross the link instead of 4.3 GB (16-35 ms depending on the host, pageable
//     or pinned cube alike);
//   * a PINNED cube without CPU threads: the GPU fetches the 128-byte blocks the map samples (the granularity
//     of PCIe reads; PM_OPT_FETCH_BLOCK_BYTES), each once, into the table (PM_OPT_HOST_CUBE_ROUTE 2: 2.2 GB over
//     the link, 44 ms), or the reprojection kernel gathers from host memory in place (1: uncached 128-byte
//     line requests that neighbouring waves repeat, 68-71 ms; the result can be stored straight into a pinned
//     output);
//   * a pinned cube and FEW CPU threads (a rank of a sharded cube whose node shares one CPU quota): a hybrid
//     of the last two (4) - short chunks of planes dealt out as the call runs, to the GPU's fetch while fewer
//     than two of its chunks are queued, else to the copy threads, who collect while those fetches cross the
//     link; the split follows the speed of the two legs by itself (64 planes, 2 threads: 3.7-4.6 ms against
//     5.5-6.0 fetched and 6.3-9.1 collected, box to box; tools/probes/route_ab.py).
//
// No compute happens on the CPU here: the threads move bytes (and write the constant NaN where the
// kernels' own pre-mask says nothing else can be).
#include <chrono>

#include "pm_host.hip.h"
#include "pm_hostpool.h"

namespace pmh {

namespace {

int usable_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    // container CPU quota, if any
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long period = 0;
        if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long quota = std::atol(q);
            if (quota > 0) n = std::min<long>(n, std::max<long>(1, quota / period));
        }
        std::fclose(f);
    }
    return n;
}

}  // namespace

// the device side of the pool's staging ring: DMA engines and events of the HIP runtime
struct HipBackend : CopyBackend {
    int device = 0;
    hipEvent_t ev_stage[HostPool::kSlots] = {};
    void thread_init() override { (void)hipSetDevice(device); }
    int copy_d2h(void *dst, const void *src, size_t bytes, void *stream) override
    {
        return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    }
    int copy_d2h_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, void *stream) override
    {
        return (int)hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)stream);
    }
    int record(int slot, void *stream) override { return (int)hipEventRecord(ev_stage[slot], (hipStream_t)stream); }
    int wait(int slot) override { return (int)hipEventSynchronize(ev_stage[slot]); }
    void *alloc_pinned(size_t bytes) override
    {
        void *p = nullptr;
        return hipHostMalloc(&p, bytes, hipHostMallocNonCoherent) == hipSuccess ? p : nullptr;
    }
    void free_pinned(void *p) override { (void)hipHostFree(p); }
};

// the pool + ring of pm_hostpool.h (HIP-free, sanitizer-tested on the CPU) and the HIP state of the cube pipeline
struct HostPipe : HostPool {
    HipBackend hip;
    // ---- streams / events of the cube pipeline
    hipStream_t s_in = nullptr, s_out = nullptr;
    static constexpr int kRing = 3;
    hipEvent_t ev_in[kRing] = {}, ev_k[kRing] = {};
    hipEvent_t ev_tmp = nullptr;
    static constexpr int kFq = 4;  // GPU-fetched chunks of a hybrid segment in flight at most (+1: ring of events)
    hipEvent_t ev_f0[kFq] = {}, ev_f1[kFq] = {};  // start / end of each (timing enabled: the fetch rate is measured as it runs)
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // (with timing: stage times of a probe chunk)
    // PM_OPT_TRACE bit 4: device-side durations of every chunk's H2D copy and kernels (start / end per ring slot, timing enabled)
    hipEvent_t ev_d0[kRing] = {}, ev_d1[kRing] = {}, ev_q0[kRing] = {}, ev_q1[kRing] = {};
    // ---- pinned staging of the H2D leg of block tables collected by the pool (one per ring slot)
    char *in_stage[kRing] = {};
    size_t in_stage_bytes = 0;
    // ---- how a host cube crosses the link (PM_OPT_HOST_CUBE_ROUTE)
    enum Route : int { kWhole = 0, kInPlace = 1, kFetch = 2, kCollect = 3, kHybrid = 4, kRoutes = 5 };
    // ---- a block table's bookkeeping (grow-only buffers): flags / tile sums / block -> row / row -> block
    // on the device, the pinned host mirror of the count + list - and what they currently describe: the
    // table is kept between calls and reused while the x/y map (by fingerprint), the plane geometry and
    // the sampling mode are the same
    struct Table {
        unsigned char *d_flags = nullptr;
        int *d_tiles = nullptr, *d_blkmap = nullptr, *d_blklist = nullptr;
        int *h_list = nullptr;  // [0] = count, [16 ...] = row -> block
        size_t d_flags_cap = 0, d_tiles_cap = 0, d_blkmap_cap = 0, d_blklist_cap = 0, h_list_cap = 0;
        bool valid = false, have_list = false;
        unsigned long long hash[2] = {0, 0};
        size_t n_map = 0, plane_bytes = 0, esz = 0, n_list = 0;
        int shift = 0, ny = 0, nx = 0, interpolation = 0, propagate_nan = 0;
    };
    Table tab[2];  // [0]: 16-byte blocks (collected by the pool), [1]: the GPU's fetch blocks (128 bytes; `t256`, `have256` below keep the names of its first size)
    unsigned long long *d_hash = nullptr, *h_hash = nullptr;
    long table_hits = 0, table_builds = 0;
    // ---- what each route has cost on the problem at hand (ns per plane, whole pipeline), measured
    struct RouteStats {
        size_t plane_bytes = 0, n_list16 = 0, esz = 0;
        bool pinned = false, device_out = false;
        int threads = 0;
        double ns_per_plane[kRoutes] = {0.0, 0.0, 0.0, 0.0, 0.0};  // 0: not measured yet
        // the two legs of the collected route, apart (probe chunk): the copy threads' collection and the DMA of
        // the table - what a hybrid's split is computed from
        double collect_cpu_ns = 0.0, collect_dma_ns = 0.0;
        double hybrid_fetch_share = 0.0;  // of the planes of a hybrid segment: fetched by the GPU
        double hybrid_fetch_ns = 0.0;     // the GPU's fetch per plane as measured INSIDE hybrid segments (the link is shared there)
        int committed = -1;
        // a hybrid is a candidate on the strength of single-chunk probes, which are noisy (cold threads, a lone chunk):
        // it is committed only after WHOLE calls by the best single route and by the hybrid have been timed in turn
        int single = -1;            // the best single route by the probes
        bool trial = false;         // whole calls still alternate between `single` and the hybrid
        double trial_ns[2] = {0.0, 0.0};  // [0] single, [1] hybrid: the MEDIAN of its three whole calls, ns per plane
        double trial_samples[2][3] = {};
        int trial_n[2] = {0, 0};
        double committed_trial_ns = 0.0;  // what the committed route showed in its trial: the trial re-opens if it drifts away
    };
    RouteStats rstats;
    double *d_maps = nullptr;  // the x/y maps of a PM_MEM_HOST call
    size_t d_maps_cap = 0;
};

static int pipe_get(pm_ctx *ctx, HostPipe **out)
{
    if (!ctx->pipe) {
        HostPipe *hp = new (std::nothrow) HostPipe();
        if (!hp) return fail(ctx, PM_ERR_ALLOC, "out of memory");
        ctx->pipe = hp;
        PM_HIP(ctx, hipStreamCreateWithFlags(&hp->s_in, hipStreamNonBlocking));
        PM_HIP(ctx, hipStreamCreateWithFlags(&hp->s_out, hipStreamNonBlocking));
        for (int i = 0; i < HostPipe::kRing; i++) {
            PM_HIP(ctx, hipEventCreateWithFlags(&hp->ev_in[i], hipEventDisableTiming));
            PM_HIP(ctx, hipEventCreateWithFlags(&hp->ev_k[i], hipEventDisableTiming));
        }
        PM_HIP(ctx, hipEventCreateWithFlags(&hp->ev_tmp, hipEventDisableTiming));
        for (int i = 0; i < HostPipe::kFq; i++) {
            PM_HIP(ctx, hipEventCreate(&hp->ev_f0[i]));
            PM_HIP(ctx, hipEventCreate(&hp->ev_f1[i]));
        }
        PM_HIP(ctx, hipEventCreate(&hp->ev_t0));
        PM_HIP(ctx, hipEventCreate(&hp->ev_t1));
        for (int i = 0; i < HostPipe::kSlots; i++)
            PM_HIP(ctx, hipEventCreateWithFlags(&hp->hip.ev_stage[i], hipEventDisableTiming));
        hp->hip.device = ctx->device;
        hp->be = &hp->hip;
    }
    HostPipe *hp = ctx->pipe;
    int threads = ctx->host_copy_threads;
    if (threads <= 0) {
        // the cores this process may count on: its share of the node when a launcher started one
        // process per GPU (torchrun exports LOCAL_WORLD_SIZE)
        int cores = usable_cores();
        if (const char *lws = std::getenv("LOCAL_WORLD_SIZE")) {
            const int n = std::atoi(lws);
            if (n > 1) cores = std::max(2, cores / n);
        }
        threads = std::min(16, cores);
    }
    if (const char *e = pm_debug_env("PM_POOL_SPIN_US")) hp->spin_us = std::max(0, std::atoi(e));  // (A/B of tools/)
    hp->start_workers(threads);
    hp->start_retirer();
    *out = hp;
    return PM_OK;
}

void pipe_destroy(pm_ctx *ctx)
{
    HostPipe *hp = ctx->pipe;
    if (!hp) return;
    hp->stop_retirer();
    hp->stop_workers();
    hp->free_stage();
    hp->be = nullptr;  // (the pool's own destructor runs after the backend member is gone)
    for (int i = 0; i < HostPipe::kSlots; i++)
        if (hp->hip.ev_stage[i]) (void)hipEventDestroy(hp->hip.ev_stage[i]);
    for (auto &t : hp->tab) {
        if (t.d_flags) (void)hipFree(t.d_flags);
        if (t.d_tiles) (void)hipFree(t.d_tiles);
        if (t.d_blkmap) (void)hipFree(t.d_blkmap);
        if (t.d_blklist) (void)hipFree(t.d_blklist);
        if (t.h_list) (void)hipHostFree(t.h_list);
    }
    if (hp->d_hash) (void)hipFree(hp->d_hash);
    if (hp->h_hash) (void)hipHostFree(hp->h_hash);
    if (hp->d_maps) (void)hipFree(hp->d_maps);
    for (int i = 0; i < HostPipe::kRing; i++) {
        if (hp->in_stage[i]) (void)hipHostFree(hp->in_stage[i]);
        if (hp->ev_in[i]) (void)hipEventDestroy(hp->ev_in[i]);
        if (hp->ev_k[i]) (void)hipEventDestroy(hp->ev_k[i]);
    }
    if (hp->ev_tmp) (void)hipEventDestroy(hp->ev_tmp);
    for (int i = 0; i < HostPipe::kFq; i++) {
        if (hp->ev_f0[i]) (void)hipEventDestroy(hp->ev_f0[i]);
        if (hp->ev_f1[i]) (void)hipEventDestroy(hp->ev_f1[i]);
    }
    if (hp->ev_t0) (void)hipEventDestroy(hp->ev_t0);
    if (hp->ev_t1) (void)hipEventDestroy(hp->ev_t1);
    if (hp->s_in) (void)hipStreamDestroy(hp->s_in);
    if (hp->s_out) (void)hipStreamDestroy(hp->s_out);
    delete hp;
    ctx->pipe = nullptr;
}

long pipe_table_hits(const pm_ctx *ctx) { return ctx->pipe ? ctx->pipe->table_hits : 0; }
long pipe_route_ns_per_plane(const pm_ctx *ctx, int route)
{
    if (!ctx->pipe || route < 0 || route >= HostPipe::kRoutes) return 0;
    return (long)ctx->pipe->rstats.ns_per_plane[route];
}
long pipe_hybrid_fetch_permille(const pm_ctx *ctx) { return ctx->pipe ? (long)std::llround(ctx->pipe->rstats.hybrid_fetch_share * 1000.0) : 0; }
int pipe_copy_threads(const pm_ctx *ctx) { return ctx->pipe ? (int)ctx->pipe->workers.size() + 1 : 0; }
void pipe_reset_route_stats(pm_ctx *ctx)
{
    if (ctx->pipe) ctx->pipe->rstats = HostPipe::RouteStats{};
}

// Is [p, p + bytes) page-locked host memory the GPU can address (hipHostMalloc / hipHostRegister)?
bool host_is_pinned(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    auto pinned_at = [](const void *q) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();  // plain malloc memory: "invalid value", not an error here
            return false;
        }
        return at.type == hipMemoryTypeHost;
    };
    return pinned_at(p) && pinned_at((const char *)p + bytes - 1);
}

static int ensure_stage(pm_ctx *ctx, HostPipe *hp)
{
    const size_t want = std::min<size_t>(std::max<size_t>(ctx->host_chunk_bytes / 2, (size_t)4 << 20), (size_t)64 << 20);
    if (!hp->ensure_stage(want)) return fail(ctx, PM_ERR_ALLOC, "hipHostMalloc of a %zu-byte staging buffer failed", want);
    return PM_OK;
}

// grow-only buffer of the pipe (device memory, or pinned host memory); contents are not kept
template <typename T>
static int grow(pm_ctx *ctx, T **p, size_t *cap, size_t bytes, bool host)
{
    if (*p && *cap >= bytes) return PM_OK;
    if (*p) PM_HIP(ctx, host ? hipHostFree(*p) : hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    const hipError_t e = host ? hipHostMalloc((void **)p, want, hipHostMallocDefault) : hipMalloc((void **)p, want);
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "allocation of %zu bytes for the block table failed", want);
    *cap = want;
    return PM_OK;
}

// pinned buffers the pool collects block tables into, one per slot of the device ring
static int ensure_in_stage(pm_ctx *ctx, HostPipe *hp, size_t bytes)
{
    if (hp->in_stage[0] && hp->in_stage_bytes >= bytes) return PM_OK;
    PM_HIP(ctx, hipStreamSynchronize(hp->s_in));
    for (int i = 0; i < HostPipe::kRing; i++) {
        if (hp->in_stage[i]) PM_HIP(ctx, hipHostFree(hp->in_stage[i]));
        hp->in_stage[i] = nullptr;
    }
    hp->in_stage_bytes = 0;
    for (int i = 0; i < HostPipe::kRing; i++) {
        hipError_t e = hipHostMalloc((void **)&hp->in_stage[i], bytes, hipHostMallocNonCoherent);
        if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipHostMalloc of a %zu-byte staging buffer failed", bytes);
    }
    hp->in_stage_bytes = bytes;
    return PM_OK;
}

// Enqueue dst_host <- src_dev on `stream`. Pinned destinations: one DMA. Pageable destinations:
// pieces through the pool's staging ring (pm_hostpool.h). d2h_finish() completes everything.
int d2h_issue(pm_ctx *ctx, hipStream_t stream, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return PM_OK;
    HostPipe *hp;
    int rc = pipe_get(ctx, &hp);
    if (rc != PM_OK) return rc;
    if (bytes < ((size_t)256 << 10) || host_is_pinned(dst_host, bytes)) {
        // (small pageable copies: the runtime's own path is as good as anything)
        PM_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        return PM_OK;
    }
    rc = ensure_stage(ctx, hp);
    if (rc != PM_OK) return rc;
    const int e = hp->issue(dst_host, src_dev, bytes, (void *)stream);
    if (e != 0) return fail(ctx, PM_ERR_HIP, "staged D2H copy failed: %s", hipGetErrorString((hipError_t)e));
    return PM_OK;
}

// A frame plane that is NaN outside the circle (x - x0)^2 + (y - y0)^2 <= r2 (the image kernels' radius
// pre-mask): only bands of rows around that circle cross the link (HostPool::issue_disc).
int d2h_issue_disc(pm_ctx *ctx, hipStream_t stream, double *dst_host, const double *src_dev, size_t nx, size_t n_rows, double y_first,
                   double x0, double y0, double r2)
{
    HostPipe *hp;
    int rc = pipe_get(ctx, &hp);
    if (rc != PM_OK) return rc;
    const bool pinned = host_is_pinned(dst_host, nx * n_rows * sizeof(double));
    if (!pinned) {
        rc = ensure_stage(ctx, hp);
        if (rc != PM_OK) return rc;
    }
    const int e = hp->issue_disc(dst_host, src_dev, nx, n_rows, y_first, x0, y0, r2, (void *)stream, pinned);
    if (e != 0) return fail(ctx, PM_ERR_HIP, "staged D2H copy failed: %s", hipGetErrorString((hipError_t)e));
    return PM_OK;
}

// An error return in the middle of a pipelined call: nothing of it may still be running when the
// caller gets its arrays back. Waits for every issued piece (DMA, retire thread, pool jobs) and for the
// three streams; errors met on the way are dropped - the call already reports one.
void pipe_abort(pm_ctx *ctx)
{
    HostPipe *hp = ctx->pipe;
    if (hp) {
        hp->drain();
        if (hp->s_in) (void)hipStreamSynchronize(hp->s_in);
        if (hp->s_out) (void)hipStreamSynchronize(hp->s_out);
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
}

int d2h_finish(pm_ctx *ctx, hipStream_t stream)
{
    HostPipe *hp = ctx->pipe;
    if (hp) {
        const int e = hp->finish();
        if (e != 0) return fail(ctx, PM_ERR_HIP, "D2H staging failed: %s", hipGetErrorString((hipError_t)e));
    }
    PM_HIP(ctx, hipStreamSynchronize(stream));
    return PM_OK;
}

// ------------------------------------------------------------------ cube pipeline (nearest / linear, propagate_nan)
namespace {

struct CubeJob {
    const char *cube;  // host
    int dtype;
    size_t esz, plane_bytes, nmap;
    int n_planes;
    const double *x_map, *y_map;  // host
    double *out;                  // host (device with device_out)
    bool device_out;
    pm::ReprojectArgs a;          // ny, nx, n_map, interpolation, propagate_nan filled in
};

// planes whose sampled pixels turned out to need the plane nanmedian: one at a time, synchronously
// (rare: +-inf pixels or neighbourhoods without a finite pixel)
int redo_with_median(pm_ctx *ctx, const CubeJob &j, const std::vector<int> &planes, const double *dxm, const double *dym,
                     char *dplane, double *dout)
{
    int rc = ensure_stats(ctx, 1);
    if (rc != PM_OK) return rc;
    const size_t plane_elems = (size_t)j.a.ny * j.a.nx;
    for (int p : planes) {
        PM_HIP(ctx, hipMemcpyAsync(dplane, j.cube + (size_t)p * j.plane_bytes, j.plane_bytes, hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, sizeof(pm::PlaneStats), ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, 512 * sizeof(unsigned int), ctx->stream));
        pm_launch_plane_medians(dplane, j.dtype, 1, plane_elems, ctx->stats, ctx->hist, ctx->stream);
        pm::ReprojectArgs b = j.a;
        b.cube = dplane;
        b.x_map = dxm;
        b.y_map = dym;
        b.out = dout;
        b.n_planes = 1;
        b.plane_flags = ctx->flags + p;
        b.plane_stats = ctx->stats;
        pm_launch_reproject(b, j.dtype, ctx->stream);
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipMemcpyAsync(j.out + (size_t)p * j.nmap, dout, j.nmap * sizeof(double),
                                   j.device_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PM_OK;
}

double now_ns()
{
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// 128-bit fingerprint of the device-resident x/y maps (k_hash_maps): 16 bytes and one host round trip
int maps_fingerprint(pm_ctx *ctx, HostPipe *hp, const double *dxm, const double *dym, size_t nmap, unsigned long long h[2])
{
    if (!hp->d_hash) PM_HIP(ctx, hipMalloc((void **)&hp->d_hash, 2 * sizeof(unsigned long long)));
    if (!hp->h_hash) PM_HIP(ctx, hipHostMalloc((void **)&hp->h_hash, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    const hipStream_t sk = ctx->stream;
    PM_HIP(ctx, hipMemsetAsync(hp->d_hash, 0, 2 * sizeof(unsigned long long), sk));
    pm_launch_hash_maps(dxm, dym, (int)nmap, hp->d_hash, sk);
    PM_HIP(ctx, hipGetLastError());
    PM_HIP(ctx, hipMemcpyAsync(hp->h_hash, hp->d_hash, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, sk));
    PM_HIP(ctx, hipStreamSynchronize(sk));
    h[0] = hp->h_hash[0];
    h[1] = hp->h_hash[1];
    return PM_OK;
}

// The block table of shift `shift` for this map and plane geometry: the one the pipe holds (the reference
// keeps its x/y map between calls, body_xy.py:3478 - a cube mapped plane block by plane block, or frame
// after frame with the same navigation, meets the same table every time) or a fresh one: k_mark_blocks runs
// the sampling code itself over the map once and flags the blocks it loads from, k_blocks_* number them;
// the host reads the count and, when it is the one that collects (want_list), the list.
int table_get(pm_ctx *ctx, HostPipe *hp, HostPipe::Table &t, int shift, const pm::ReprojectArgs &am, int dtype, size_t plane_bytes,
              size_t esz, bool have_hash, const unsigned long long hash[2], bool want_list)
{
    const hipStream_t sk = ctx->stream;
    const size_t n_blk = (plane_bytes + ((size_t)1 << shift) - 1) >> shift;
    const bool hit = have_hash && t.valid && t.shift == shift && t.hash[0] == hash[0] && t.hash[1] == hash[1] &&
                     t.n_map == (size_t)am.n_map && t.plane_bytes == plane_bytes && t.ny == am.ny && t.nx == am.nx && t.esz == esz &&
                     t.interpolation == am.interpolation && t.propagate_nan == am.propagate_nan;
    int rc;
    if (!hit) {
        t.valid = false;
        t.have_list = false;
        const size_t n_tiles = (n_blk + 4095) / 4096, n_pad = n_tiles * 4096;
        if ((rc = grow(ctx, &t.d_flags, &t.d_flags_cap, n_pad, false)) != PM_OK) return rc;
        if ((rc = grow(ctx, &t.d_tiles, &t.d_tiles_cap, (n_tiles + 1) * sizeof(int), false)) != PM_OK) return rc;
        if ((rc = grow(ctx, &t.d_blkmap, &t.d_blkmap_cap, n_blk * sizeof(int), false)) != PM_OK) return rc;
        if ((rc = grow(ctx, &t.d_blklist, &t.d_blklist_cap, n_blk * sizeof(int), false)) != PM_OK) return rc;
        if ((rc = grow(ctx, &t.h_list, &t.h_list_cap, 64, true)) != PM_OK) return rc;
        int *d_total = t.d_tiles + n_tiles;
        PM_HIP(ctx, hipMemsetAsync(t.d_flags, 0, n_pad, sk));
        pm_launch_mark_blocks(am, t.d_flags, shift, dtype, sk);
        pm_launch_number_blocks(t.d_flags, n_blk, t.d_tiles, t.d_blkmap, t.d_blklist, d_total, sk);
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipMemcpyAsync(t.h_list, d_total, sizeof(int), hipMemcpyDeviceToHost, sk));
        PM_HIP(ctx, hipStreamSynchronize(sk));
        t.n_list = (size_t)t.h_list[0];
        t.shift = shift;
        t.n_map = (size_t)am.n_map;
        t.plane_bytes = plane_bytes;
        t.ny = am.ny;
        t.nx = am.nx;
        t.esz = esz;
        t.interpolation = am.interpolation;
        t.propagate_nan = am.propagate_nan;
        if (have_hash) {
            t.hash[0] = hash[0];
            t.hash[1] = hash[1];
            t.valid = true;
        }
        hp->table_builds++;
    } else {
        hp->table_hits++;
    }
    if (want_list && !t.have_list && t.n_list > 0) {
        // (the first 64 bytes of h_list hold the count; a regrown buffer gets it back)
        if ((rc = grow(ctx, &t.h_list, &t.h_list_cap, 64 + t.n_list * sizeof(int), true)) != PM_OK) return rc;
        t.h_list[0] = (int)t.n_list;
        PM_HIP(ctx, hipMemcpyAsync(t.h_list + 16, t.d_blklist, t.n_list * sizeof(int), hipMemcpyDeviceToHost, sk));
        PM_HIP(ctx, hipStreamSynchronize(sk));
        t.have_list = true;
    }
    return PM_OK;
}

// One stretch of planes of a call, fed by one route (HostPipe::Route). A call is one segment - or, the
// first time the library is left to choose for a problem, a few short ones, one per candidate route,
// timed, and the rest by the fastest.
struct Segment {
    int route;
    size_t p0, n;       // planes [p0, p0 + n) of the call
    size_t chunk;       // planes per chunk of the pipeline (0: the route's default)
    bool probe;         // timed for the route statistics
};

struct SegLayout {
    size_t chunk, batch, slot_bytes, table_row_bytes, blk_off, need;
    size_t chunk_f = 0, table_row_bytes_f = 0;  // hybrid: planes per GPU-fetched chunk, one plane's rows of the fetched table
    bool zero_copy, direct_out;
};

// planes per chunk of a hybrid segment: collected chunks, and the (shorter) GPU-fetched ones, of which up to
// kHybridInFlight are queued ahead of the threads
// (the collected chunk grows with the copy threads, one plane each: their jobs must stay worth a wake-up - with 16
//  threads chunks of 4 planes made the hybrid 30 % SLOWER than the collected route, chunks of 16 made it faster)
constexpr size_t kHybridCollectChunk = 4, kHybridFetchChunk = 4, kHybridCollectChunkMax = 16;
inline size_t hybrid_collect_chunk(const pm_ctx *ctx)
{
    size_t c = kHybridCollectChunk;
    if (ctx->pipe) c = std::max<size_t>(c, ctx->pipe->workers.size() + 1);
    if (const char *e = pm_debug_env("PM_HYBRID_CCHUNK")) c = (size_t)std::max(1, std::atoi(e));  // (A/B of tools/probes/route_ab.py)
    return std::min(c, kHybridCollectChunkMax);
}
constexpr int kHybridInFlight = 3;

// (hybrid: n_list / shift describe the 16-byte table, n_list_f the fetched one)
SegLayout seg_layout(const pm_ctx *ctx, const CubeJob &j, const Segment &sg, size_t n_list, int shift, bool dst_pinned, size_t n_list_f = 0)
{
    SegLayout L{};
    const size_t nmap = j.nmap;
    const bool hybrid = sg.route == HostPipe::kHybrid;
    const bool gather = sg.route == HostPipe::kInPlace, blocks = sg.route == HostPipe::kFetch, host_blocks = sg.route == HostPipe::kCollect || hybrid;
    L.zero_copy = gather || blocks;  // no staging ring: the kernels read the caller's cube
    L.table_row_bytes = (blocks || host_blocks) ? n_list << shift : 0;  // one plane's rows of the block table
    size_t chunk = std::max<size_t>(1, ctx->host_chunk_bytes / j.plane_bytes);
    chunk = std::min<size_t>(std::min<size_t>(chunk, sg.n), 32768);
    if (L.zero_copy) chunk = std::min<size_t>(std::max<size_t>(chunk, (sg.n + 7) / 8), 32768);
    if (blocks) chunk = std::max<size_t>(1, std::min<size_t>(chunk, ((size_t)256 << 20) / L.table_row_bytes));
    if (host_blocks)  // chunks of the table, not of the cube (smaller chunks for short cubes: no gain, measured)
        chunk = std::min<size_t>(std::max<size_t>(1, ctx->host_chunk_bytes / L.table_row_bytes), std::min<size_t>(sg.n, 32768));
    if (sg.chunk) chunk = std::min<size_t>(sg.chunk, sg.n);
    if (hybrid) {
        // short chunks, dealt out as the segment runs (run_segment): to the GPU while fewer than kHybridInFlight
        // fetched chunks are queued, else to the copy threads
        chunk = std::min<size_t>(hybrid_collect_chunk(ctx), sg.n);
        L.table_row_bytes_f = n_list_f << ctx->fetch_shift;
        // (the closing chunk of a segment may hand the GPU whatever is left after the threads' last share)
        L.chunk_f = std::min<size_t>(sg.n, std::max<size_t>(kHybridFetchChunk, 3 * kHybridCollectChunk) + chunk);
    }
    L.chunk = chunk;
    L.batch = std::min<size_t>(sg.n, std::max<size_t>(chunk, ((size_t)1 << 30) / (nmap * sizeof(double))));
    L.slot_bytes = ((host_blocks ? chunk * L.table_row_bytes : chunk * j.plane_bytes) + 255) & ~(size_t)255;
    // the kernel stores straight into the caller's array: device memory, or pinned memory next to a zero-copy cube
    L.direct_out = j.device_out || (L.zero_copy && dst_pinned);
    size_t need = 2 * nmap * sizeof(double) + 256;
    if (!L.zero_copy) need += HostPipe::kRing * L.slot_bytes;
    if (!L.direct_out) need += L.batch * nmap * sizeof(double);
    L.blk_off = (need + 255) & ~(size_t)255;  // the table of a GPU-fetched chunk
    if (blocks) need = L.blk_off + chunk * L.table_row_bytes;
    if (hybrid) need = L.blk_off + L.chunk_f * L.table_row_bytes_f;
    L.need = need;
    return L;
}

// the pipeline over one segment: H2D of chunk k + 1 (whole planes, or the block table the pool has
// collected), kernel of chunk k and the staged D2H of finished output overlap on three streams
// `stage_ns` (probe segments: ONE chunk): the time of the route's slowest pipeline stage for that chunk -
// what a long run of such chunks costs per chunk once the stages overlap: the copy threads' collection
// against the DMA of the table (3), the DMA of the planes (0), the GPU's own fetch over the link (2).
// (The wall time of a lone chunk would add the stages up and charge the pipeline's fill to the route.)
// A hybrid segment (route 4) deals its chunks out between the GPU's fetch (`tab_f`, the table of fetch blocks) and the
// copy threads (`tab`, the 16-byte table) as it runs: see the loop. `stage2` (probes): [0] the CPU leg, [1] the device leg of the stage.
int run_segment(pm_ctx *ctx, HostPipe *hp, const CubeJob &j, const Segment &sg, const SegLayout &L, const HostPipe::Table *tab,
                const double *dxm, const double *dym, const char *cube_dev, double *out_dev, double *stage_ns, double *stage2 = nullptr,
                const HostPipe::Table *tab_f = nullptr, double *hybrid_obs = nullptr)
{
    // (hybrid_obs, hybrid segments. In: [0] ns per plane the threads took to collect in earlier calls, [2] the GPU to
    //  fetch. Out: [0] ns the threads spent collecting, [1] planes they collected, [2] ns the GPU spent fetching,
    //  [3] planes it fetched)
    double hy_cpu_ns = 0.0, hy_c_planes = 0.0, hy_f_ns = 0.0, hy_f_planes = 0.0;
    const double t_seg0 = now_ns();
    // GPU-fetched chunks of a hybrid segment in flight: [f_tail, f_head) in the ring of event pairs
    size_t f_head = 0, f_tail = 0, f_planes_out = 0;
    size_t f_np[HostPipe::kFq] = {};
    double t_f_est = hybrid_obs ? hybrid_obs[2] : 0.0, t_c_est = hybrid_obs ? hybrid_obs[0] : 0.0;  // ns per plane, from earlier calls
    bool endgame = false;
    size_t end_f = 0, end_c = 0;  // the closing split: planes still to hand to the GPU / to the threads
    auto retire_fetched = [&](bool wait) -> int {
        while (f_tail < f_head) {
            const int k = (int)(f_tail % HostPipe::kFq);
            if (wait) {
                PM_HIP(ctx, hipEventSynchronize(hp->ev_f1[k]));
            } else {
                const hipError_t q = hipEventQuery(hp->ev_f1[k]);
                if (q == hipErrorNotReady) {
                    (void)hipGetLastError();
                    break;
                }
                PM_HIP(ctx, q);
            }
            float ms = 0.0f;
            PM_HIP(ctx, hipEventElapsedTime(&ms, hp->ev_f0[k], hp->ev_f1[k]));
            hy_f_ns += (double)ms * 1e6;
            hy_f_planes += (double)f_np[k];
            f_planes_out -= f_np[k];
            f_tail++;
        }
        return PM_OK;
    };
    double cpu_stage_ns = 0.0;
    bool timed_events = false;
    const hipStream_t sk = ctx->stream;
    const size_t nmap = j.nmap;
    const bool hybrid = sg.route == HostPipe::kHybrid;
    // stage record of the call (PM_OPT_LAST_STAGE_NS): host-side times always, device-side sums with PM_OPT_TRACE bit 4
    double *const st = ctx->last_stage_ns;
    const bool dev_times = (ctx->trace & 4) != 0;
    if (dev_times && !hp->ev_d0[0])
        for (int i = 0; i < HostPipe::kRing; i++) {
            PM_HIP(ctx, hipEventCreate(&hp->ev_d0[i]));
            PM_HIP(ctx, hipEventCreate(&hp->ev_d1[i]));
            PM_HIP(ctx, hipEventCreate(&hp->ev_q0[i]));
            PM_HIP(ctx, hipEventCreate(&hp->ev_q1[i]));
        }
    bool slot_timed[HostPipe::kRing] = {};
    auto bank_slot = [&](int slot) -> int {  // the device durations of the chunk that used `slot` last (it has finished)
        if (!dev_times || !slot_timed[slot]) return PM_OK;
        float ms = 0.0f;
        PM_HIP(ctx, hipEventSynchronize(hp->ev_q1[slot]));
        PM_HIP(ctx, hipEventElapsedTime(&ms, hp->ev_d0[slot], hp->ev_d1[slot]));
        st[kStageDmaDevice] += (double)ms * 1e6;
        PM_HIP(ctx, hipEventElapsedTime(&ms, hp->ev_q0[slot], hp->ev_q1[slot]));
        st[kStageKernelDevice] += (double)ms * 1e6;
        slot_timed[slot] = false;
        return PM_OK;
    };
    bool first_h2d = true;
    const bool seg_blocks = sg.route == HostPipe::kFetch, seg_host_blocks = sg.route == HostPipe::kCollect;
    char *base = (char *)ctx->scratch;
    char *ring = base + ((2 * nmap * sizeof(double) + 255) & ~(size_t)255);
    double *dout_all = (double *)(ring + (L.zero_copy ? 0 : HostPipe::kRing * L.slot_bytes));
    auto describe = [&](const HostPipe::Table *t) {
        pm::BlockTable bt{};
        bt.blkmap = t->d_blkmap;
        bt.blklist = t->d_blklist;
        bt.n_list = (unsigned)t->n_list;
        bt.shift = t->shift;
        bt.plane_bytes = j.plane_bytes;
        return bt;
    };
    pm::BlockTable table{}, table_f{};
    if (seg_blocks || seg_host_blocks || hybrid) table = describe(tab);
    if (seg_blocks) table.table = base + L.blk_off;
    if (hybrid) {
        table_f = describe(tab_f);
        table_f.table = base + L.blk_off;
    }
    const int *hlist = ((seg_host_blocks || hybrid) && tab->h_list) ? tab->h_list + 16 : nullptr;
    int rc;
    // s_out drains finished output while later chunks are still being copied in / mapped
    size_t c = 0;   // running chunk number of the segment
    size_t cs = 0;  // running number of the chunks that use a ring slot (slot = cs % kRing; hybrid: the collected ones)
    for (size_t b0 = 0; b0 < sg.n; b0 += L.batch) {
        const size_t nb = std::min(L.batch, sg.n - b0);
        size_t drained = 0;  // planes of this batch already handed to the D2H leg
        size_t launched = 0;
        endgame = false;  // (the closing split belongs to a batch)
        // (Tried: ramping the chunks of a collected table up and down - c/8, c/4 ... c ... halves of the rest -
        //  to shorten the pipeline's fill and drain for short blocks. Same-box A/B, 64 and 512 planes, three
        //  process pairs: 2.26-2.86 vs 2.17-2.52 ms and 11.9-16.3 vs 13.0-14.6 ms - inside the run-to-run
        //  spread of the collecting threads' placement; not kept. Round 5, with the stage record (PM_OPT_LAST_STAGE_NS) on a
        //  64-plane block, 16 threads: a first chunk of 2 planes doubling up to 27 takes "first_fill" from 0.55 to 0.09 ms and
        //  leaves the call at 2.2 ms, a closing chunk of 3 planes leaves "drain" at 0.55 ms: the DMA of the table (21 us per
        //  plane) and its collection (22 us) run at the same rate, so the link is always one chunk behind the threads - the
        //  call is collection + the DMA of its last FULL chunk however the chunks are cut, and every extra chunk costs the
        //  collection 50-80 us of waking and joining sixteen threads.)
        size_t np = 0;
        for (size_t q0 = 0; q0 < nb; q0 += np, c++) {
            // Hybrid: chunks are dealt out as the segment runs - to the GPU while fewer than kHybridInFlight of its
            // fetches are queued, else to the copy threads (who collect while those fetches cross the link): the
            // split follows the speed of the two legs by itself. Only the END needs rates: when little is left, the
            // threads take the share x of it that lets both legs finish together - x t_c = (queued + rest - x) t_f,
            // with t_c, t_f as measured in this segment so far (earlier calls before that) - and the GPU the rest.
            bool fetch_chunk = false;
            size_t want = L.chunk;
            if (hybrid) {
                if ((rc = retire_fetched(false)) != PM_OK) return rc;
                const size_t rest = nb - q0;
                const double tc = hy_c_planes > 0.0 ? hy_cpu_ns / hy_c_planes : t_c_est;
                const double tf = hy_f_planes > 0.0 ? hy_f_ns / hy_f_planes : t_f_est;
                if (!endgame && rest <= 2 * L.chunk + kHybridFetchChunk && tc > 0.0 && tf > 0.0) {
                    // the closing split: x planes for the threads, the others for the GPU - whose share is issued FIRST
                    // (it runs behind what is queued while the threads collect theirs)
                    const double x = (double)(f_planes_out + rest) * tf / (tc + tf);
                    end_c = std::min<size_t>(rest, (size_t)std::llround(x));
                    end_f = rest - end_c;
                    endgame = true;
                }
                if (endgame) {
                    fetch_chunk = end_f > 0;
                    want = fetch_chunk ? end_f : std::min(end_c, L.chunk);  // (a collected chunk never exceeds its ring slot)
                    want = std::max<size_t>(want, 1);
                    if (fetch_chunk) end_f -= std::min(std::min(want, L.chunk_f), rest);
                    else end_c -= std::min(want, rest);
                } else {
                    fetch_chunk = (int)(f_head - f_tail) < kHybridInFlight;
                    want = fetch_chunk ? kHybridFetchChunk : L.chunk;
                }
                if (fetch_chunk) want = std::min(want, L.chunk_f);
                if (fetch_chunk && (int)(f_head - f_tail) >= HostPipe::kFq && (rc = retire_fetched(true)) != PM_OK) return rc;
            }
            const bool blocks = seg_blocks || fetch_chunk, host_blocks = seg_host_blocks || (hybrid && !fetch_chunk);
            const bool chunk_zero_copy = L.zero_copy || fetch_chunk;
            np = std::min(want, nb - q0);
            if (hybrid && (ctx->trace & 1))
                std::fprintf(stderr, "[pm hostpipe]   hybrid chunk %zu at %.3f ms: %s %zu planes (fetches in flight %zu, %zu planes)\n", c,
                             (now_ns() - t_seg0) * 1e-6, fetch_chunk ? "F" : "C", np, f_head - f_tail, f_planes_out);
            const size_t pl = sg.p0 + b0 + q0;  // first plane of the chunk within the call
            const int slot = (int)(cs % HostPipe::kRing);
            const size_t cslot = cs;  // (this chunk's number among the slot users)
            if (!chunk_zero_copy) cs++;
            pm::ReprojectArgs b = j.a;
            b.x_map = dxm;
            b.y_map = dym;
            b.n_planes = (int)np;
            b.plane_flags = ctx->flags + pl;
            b.out = L.direct_out ? out_dev + pl * nmap : dout_all + q0 * nmap;
            pm::BlockTable tb = fetch_chunk ? table_f : table;
            if (chunk_zero_copy) {
                b.cube = cube_dev + pl * j.plane_bytes;
            } else if (host_blocks) {
                // the pool fills this slot's pinned buffer (free once the DMA of three chunks ago is
                // done) while the DMA of the previous chunk runs
                char *dslot = ring + (size_t)slot * L.slot_bytes;
                if (cslot >= (size_t)HostPipe::kRing) PM_HIP(ctx, hipEventSynchronize(hp->ev_in[slot]));
                if ((rc = bank_slot(slot)) != PM_OK) return rc;
                const double tg = now_ns();
                hp->gather(hp->in_stage[slot], j.cube + pl * j.plane_bytes, j.plane_bytes, np, hlist, tab->n_list, tab->shift);
                const double t_gathered = now_ns();
                st[kStageCollect] += t_gathered - tg;
                if (stage_ns) cpu_stage_ns += t_gathered - tg;
                if (hybrid) {
                    hy_cpu_ns += t_gathered - tg;
                    hy_c_planes += (double)np;
                }
                if (first_h2d) {
                    st[kStageFirstFill] += t_gathered - t_seg0;  // (the link has carried nothing of this segment until now)
                    first_h2d = false;
                }
                if (cslot >= (size_t)HostPipe::kRing) PM_HIP(ctx, hipStreamWaitEvent(hp->s_in, hp->ev_k[slot], 0));
                if (stage_ns && cslot == 0) PM_HIP(ctx, hipEventRecord(hp->ev_t0, hp->s_in));
                if (dev_times) PM_HIP(ctx, hipEventRecord(hp->ev_d0[slot], hp->s_in));
                PM_HIP(ctx, hipMemcpyAsync(dslot, hp->in_stage[slot], np * L.table_row_bytes, hipMemcpyHostToDevice, hp->s_in));
                if (dev_times) PM_HIP(ctx, hipEventRecord(hp->ev_d1[slot], hp->s_in));
                if (stage_ns && cslot == 0) {
                    PM_HIP(ctx, hipEventRecord(hp->ev_t1, hp->s_in));
                    timed_events = true;
                }
                PM_HIP(ctx, hipEventRecord(hp->ev_in[slot], hp->s_in));
                PM_HIP(ctx, hipStreamWaitEvent(sk, hp->ev_in[slot], 0));
                b.cube = nullptr;  // no plane to fall back on: see cleaned_value / BlockLoader
                tb.table = dslot;
            } else {
                char *dslot = ring + (size_t)slot * L.slot_bytes;
                if (cslot >= (size_t)HostPipe::kRing) PM_HIP(ctx, hipStreamWaitEvent(hp->s_in, hp->ev_k[slot], 0));
                if ((rc = bank_slot(slot)) != PM_OK) return rc;
                if (stage_ns && c == 0) PM_HIP(ctx, hipEventRecord(hp->ev_t0, hp->s_in));
                if (first_h2d) {
                    st[kStageFirstFill] += now_ns() - t_seg0;
                    first_h2d = false;
                }
                if (dev_times) PM_HIP(ctx, hipEventRecord(hp->ev_d0[slot], hp->s_in));
                const double tc = stage_ns ? now_ns() : 0.0;
                PM_HIP(ctx, hipMemcpyAsync(dslot, j.cube + pl * j.plane_bytes, np * j.plane_bytes, hipMemcpyHostToDevice, hp->s_in));
                if (stage_ns) cpu_stage_ns += now_ns() - tc;  // (pageable planes: the runtime stages them inside the call)
                if (dev_times) PM_HIP(ctx, hipEventRecord(hp->ev_d1[slot], hp->s_in));
                if (stage_ns && c == 0) {
                    PM_HIP(ctx, hipEventRecord(hp->ev_t1, hp->s_in));
                    timed_events = true;
                }
                PM_HIP(ctx, hipEventRecord(hp->ev_in[slot], hp->s_in));
                PM_HIP(ctx, hipStreamWaitEvent(sk, hp->ev_in[slot], 0));
                b.cube = dslot;
            }
            // (the GPU's reads over the link ARE the stage; a hybrid segment times its SECOND fetched chunk - the one
            //  that shares the link with the threads' table, as every later one does)
            const bool time_kernel = stage_ns && c == 0 && chunk_zero_copy;
            const int fk = (int)(f_head % HostPipe::kFq);
            if (fetch_chunk) PM_HIP(ctx, hipEventRecord(hp->ev_f0[fk], sk));
            if (time_kernel) PM_HIP(ctx, hipEventRecord(hp->ev_t0, sk));
            const bool slot_chunk = dev_times && !chunk_zero_copy;  // (a chunk that holds a ring slot: its copy and kernels are timed per slot)
            if (slot_chunk) PM_HIP(ctx, hipEventRecord(hp->ev_q0[slot], sk));
            if (blocks || host_blocks)
                pm_launch_reproject_blocks(b, tb, j.dtype, sk, /*fetch=*/blocks);
            else
                pm_launch_reproject(b, j.dtype, sk);
            if (slot_chunk) {
                PM_HIP(ctx, hipEventRecord(hp->ev_q1[slot], sk));
                slot_timed[slot] = true;
            }
            if (time_kernel) {
                PM_HIP(ctx, hipEventRecord(hp->ev_t1, sk));
                timed_events = true;
            }
            PM_HIP(ctx, hipGetLastError());
            // (a fetched chunk of a hybrid segment holds no ring slot: its kernel gets the segment's own event)
            hipEvent_t ev_done = fetch_chunk ? hp->ev_f1[fk] : hp->ev_k[slot];
            PM_HIP(ctx, hipEventRecord(ev_done, sk));
            if (fetch_chunk) {
                f_np[fk] = np;
                f_planes_out += np;
                f_head++;
            }
            if (ctx->chunk_cb) ctx->chunk_cb(ctx->chunk_user, (int)pl, (int)np);
            launched = q0 + np;
            // hand finished output to the D2H leg in pieces worth a DMA
            if (!L.direct_out && (launched - drained) * nmap * sizeof(double) >= ((size_t)8 << 20)) {
                PM_HIP(ctx, hipStreamWaitEvent(hp->s_out, ev_done, 0));
                rc = d2h_issue(ctx, hp->s_out, j.out + (sg.p0 + b0 + drained) * nmap, dout_all + drained * nmap,
                               (launched - drained) * nmap * sizeof(double));
                if (rc != PM_OK) return rc;
                drained = launched;
            }
        }
        if (!L.direct_out) {
            if (launched > drained) {
                PM_HIP(ctx, hipEventRecord(hp->ev_tmp, sk));
                PM_HIP(ctx, hipStreamWaitEvent(hp->s_out, hp->ev_tmp, 0));
                rc = d2h_issue(ctx, hp->s_out, j.out + (sg.p0 + b0 + drained) * nmap, dout_all + drained * nmap,
                               (launched - drained) * nmap * sizeof(double));
                if (rc != PM_OK) return rc;
            }
            // the next batch reuses dout_all
            rc = d2h_finish(ctx, hp->s_out);
            if (rc != PM_OK) return rc;
        }
    }
    // the ring and the events are free for the next segment / the flag check
    if (hybrid && (ctx->trace & 1)) std::fprintf(stderr, "[pm hostpipe]   hybrid: all chunks issued at %.3f ms\n", (now_ns() - t_seg0) * 1e-6);
    const double t_issued = now_ns();
    st[kStageIssue] += t_issued - t_seg0;
    PM_HIP(ctx, hipStreamSynchronize(sk));
    PM_HIP(ctx, hipStreamSynchronize(hp->s_in));
    st[kStageDrain] += now_ns() - t_issued;
    for (int slot = 0; slot < HostPipe::kRing; slot++)
        if ((rc = bank_slot(slot)) != PM_OK) return rc;
    if (stage_ns) {
        float ms = 0.0f;
        if (timed_events) PM_HIP(ctx, hipEventElapsedTime(&ms, hp->ev_t0, hp->ev_t1));
        *stage_ns = std::max(cpu_stage_ns, (double)ms * 1e6);
        if (stage2) {
            stage2[0] = cpu_stage_ns;
            stage2[1] = (double)ms * 1e6;
        }
    }
    if (hybrid) {
        if ((rc = retire_fetched(true)) != PM_OK) return rc;
        st[kStageKernelDevice] += hy_f_ns;  // (the fetched chunks' kernels: their reads over the link are in there)
        if (hybrid_obs) {
            hybrid_obs[0] = hy_cpu_ns;
            hybrid_obs[1] = hy_c_planes;
            hybrid_obs[2] = hy_f_ns;
            hybrid_obs[3] = hy_f_planes;
        }
    }
    return PM_OK;
}

}  // namespace

static int map_cube_host_pipelined_impl(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *x_map,
                                        const double *y_map, size_t nmap, pm::ReprojectArgs a, double *out, bool device_out);

// pm_map_cube for host buffers, interpolation nearest / linear with NaN propagation (the default of
// Observation.get_mapped_data). Caller has validated the arguments and sized ctx->flags.
// `device_out`: x_map / y_map / out are DEVICE pointers (PM_MEM_HOST_CUBE): only the cube travels.
int map_cube_host_pipelined(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *x_map,
                            const double *y_map, size_t nmap, pm::ReprojectArgs a, double *out, bool device_out)
{
    const int rc = map_cube_host_pipelined_impl(ctx, cube, dtype, n_planes, x_map, y_map, nmap, a, out, device_out);
    if (rc != PM_OK) pipe_abort(ctx);  // (copies / jobs of the failed call must not outlive it)
    return rc;
}

static int map_cube_host_pipelined_impl(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *x_map,
                                        const double *y_map, size_t nmap, pm::ReprojectArgs a, double *out, bool device_out)
{
    const bool trace = (ctx->trace & 1) != 0;  // PM_OPT_TRACE: stage times of every call on stderr
    const double t_call = now_ns();
    for (int k = 0; k < kStageCount; k++) ctx->last_stage_ns[k] = 0.0;
    HostPipe *hp;
    int rc = pipe_get(ctx, &hp);
    if (rc != PM_OK) return rc;
    if (ctx->pending) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }
    CubeJob j;
    j.cube = (const char *)cube;
    j.dtype = dtype;
    j.esz = dtype_size(dtype);
    j.plane_bytes = (size_t)a.ny * a.nx * j.esz;
    j.nmap = nmap;
    j.n_planes = n_planes;
    j.x_map = x_map;
    j.y_map = y_map;
    j.out = out;
    j.device_out = device_out;
    a.plane_stats = nullptr;
    a.seq = ++ctx->map_seq;
    j.a = a;

    const size_t cube_bytes = (size_t)n_planes * j.plane_bytes;
    const size_t out_bytes = (size_t)n_planes * nmap * sizeof(double);
    const bool src_pinned = host_is_pinned(cube, cube_bytes);
    const bool dst_pinned = !device_out && host_is_pinned(out, out_bytes);
    const hipStream_t sk = ctx->stream;
    // the x/y maps on the device: the caller's (PM_MEM_HOST_CUBE) or a copy
    const double *dxm = x_map, *dym = y_map;
    if (!device_out) {
        rc = grow(ctx, &hp->d_maps, &hp->d_maps_cap, 2 * nmap * sizeof(double), false);
        if (rc != PM_OK) return rc;
        PM_HIP(ctx, hipMemcpyAsync(hp->d_maps, x_map, nmap * sizeof(double), hipMemcpyHostToDevice, sk));
        PM_HIP(ctx, hipMemcpyAsync(hp->d_maps + nmap, y_map, nmap * sizeof(double), hipMemcpyHostToDevice, sk));
        dxm = hp->d_maps;
        dym = hp->d_maps + nmap;
    }
    // How the cube crosses the link (PM_OPT_HOST_CUBE_ROUTE): whole planes by DMA (0); a pinned cube
    // gathered in place by the kernel (1); the blocks the map samples brought into a table in HBM,
    // either fetched by the GPU from a pinned cube in 128-byte blocks (2) or collected by the copy
    // threads in 16-byte blocks into pinned staging, chunk by chunk, and sent by DMA (3: any host
    // memory). Left to the library (-1) the route is the one that MEASURED fastest for this problem on
    // this context (below); until enough planes have come by to measure, (3) when the table is well
    // under half the size of the planes, else (0).
    const int mode = ctx->zero_copy;
    pm::ReprojectArgs am = a;
    am.x_map = dxm;
    am.y_map = dym;
    auto splits = [&](int shift) { return j.plane_bytes % ((size_t)1 << shift) == 0 && (j.plane_bytes >> shift) < ((size_t)1 << 31); };
    const bool can_collect = splits(pm::kBlkShiftHost), can_fetch = src_pinned && splits(ctx->fetch_shift);
    unsigned long long hash[2] = {0, 0};
    bool have_hash = false;
    const bool want16 = (mode < 0 || mode == HostPipe::kCollect || mode == HostPipe::kHybrid) && can_collect;
    const bool want256 = (mode == HostPipe::kFetch || (mode == HostPipe::kHybrid && can_collect)) && can_fetch;
    bool have256 = false;  // the table of fetch blocks describes THIS call's map
    if ((want16 || want256) && ctx->table_cache) {
        if ((rc = maps_fingerprint(ctx, hp, dxm, dym, nmap, hash)) != PM_OK) return rc;
        have_hash = true;
    }
    HostPipe::Table &t16 = hp->tab[0], &t256 = hp->tab[1];
    if (want16 && (rc = table_get(ctx, hp, t16, pm::kBlkShiftHost, am, dtype, j.plane_bytes, j.esz, have_hash, hash, /*want_list=*/true)) != PM_OK)
        return rc;
    if (want256) {
        if ((rc = table_get(ctx, hp, t256, ctx->fetch_shift, am, dtype, j.plane_bytes, j.esz, have_hash, hash, false)) != PM_OK) return rc;
        have256 = t256.n_list > 0;
    }

    const double t_tables = now_ns();
    // ---- the plan: segments of planes, each fed by one route
    std::vector<Segment> plan;
    const size_t P = (size_t)n_planes;
    HostPipe::RouteStats &rs = hp->rstats;
    bool exploring = false;
    if (mode == HostPipe::kInPlace && src_pinned) {
        plan.push_back({HostPipe::kInPlace, 0, P, 0, false});
    } else if (mode == HostPipe::kHybrid && have256 && want16 && t16.n_list > 0) {
        // (asked for: the split from what has been measured on this problem, else half and half)
        if (rs.hybrid_fetch_share <= 0.0) rs.hybrid_fetch_share = 0.5;
        plan.push_back({HostPipe::kHybrid, 0, P, 0, false});
    } else if ((mode == HostPipe::kFetch || mode == HostPipe::kHybrid) && have256) {
        plan.push_back({HostPipe::kFetch, 0, P, 0, false});
    } else if ((mode == HostPipe::kCollect || mode == HostPipe::kHybrid) && want16 && t16.n_list > 0) {
        // (a hybrid asked for on pageable memory, which the GPU cannot read: the collected half of it)
        plan.push_back({HostPipe::kCollect, 0, P, 0, false});
    } else if (mode < 0 && want16 && t16.n_list > 0) {
        // The library chooses, per context (= per rank), from what it has measured on this very problem:
        // the routes differ in which resource they lean on - the copy threads and the host's memory
        // system (3), the PCIe link alone (0), the GPU's own reads over the link (2) - and how those
        // compare depends on the box, on how many ranks share the host and on the threads this rank
        // was given, none of which the size of the table knows.
        const int threads = (int)hp->workers.size() + 1;
        if (rs.plane_bytes != j.plane_bytes || rs.n_list16 != t16.n_list || rs.pinned != src_pinned || rs.threads != threads ||
            rs.device_out != device_out || rs.esz != j.esz) {
            rs = HostPipe::RouteStats{};
            rs.plane_bytes = j.plane_bytes;
            rs.n_list16 = t16.n_list;
            rs.pinned = src_pinned;
            rs.threads = threads;
            rs.device_out = device_out;
            rs.esz = j.esz;
        }
        const size_t row16 = t16.n_list << pm::kBlkShiftHost;
        const int by_size = row16 * 5 < j.plane_bytes * 2 ? HostPipe::kCollect : HostPipe::kWhole;
        // probe segments: ONE chunk of the route's usual size each, judged by its slowest stage (run_segment)
        const size_t c3 = std::max<size_t>(1, std::min<size_t>(ctx->host_chunk_bytes / std::max<size_t>(row16, 1), 32768));
        const size_t c0 = std::max<size_t>(1, ctx->host_chunk_bytes / j.plane_bytes);
        const size_t c2 = std::max<size_t>(1, std::min<size_t>(c3, ((size_t)64 << 20) / j.plane_bytes));
        const size_t need_planes = c3 + c0 + (can_fetch ? c2 : 0);
        if ((rs.committed == HostPipe::kFetch || rs.committed == HostPipe::kHybrid || rs.trial) && can_fetch) {
            if ((rc = table_get(ctx, hp, t256, ctx->fetch_shift, am, dtype, j.plane_bytes, j.esz, have_hash, hash, false)) != PM_OK) return rc;
            have256 = t256.n_list > 0;
        }
        // calls long enough to be pipelines take part in the trial (below); shorter ones go by the single route
        const bool long_call = P >= 6 * kHybridCollectChunk;
        if (rs.trial && rs.single >= 0 && (rs.single != HostPipe::kFetch || have256)) {
            if (have256 && long_call && rs.trial_n[1] < rs.trial_n[0])
                plan.push_back({HostPipe::kHybrid, 0, P, 0, false});
            else
                plan.push_back({rs.single, 0, P, 0, false});
        } else if (rs.committed >= 0 && ((rs.committed != HostPipe::kFetch && rs.committed != HostPipe::kHybrid) || have256)) {
            plan.push_back({rs.committed, 0, P, 0, false});
        } else if (ctx->route_explore && P > need_planes) {
            exploring = true;
            size_t at = 0;
            plan.push_back({HostPipe::kCollect, at, c3, c3, true});
            at += c3;
            plan.push_back({HostPipe::kWhole, at, c0, c0, true});
            at += c0;
            if (can_fetch) {
                if ((rc = table_get(ctx, hp, t256, ctx->fetch_shift, am, dtype, j.plane_bytes, j.esz, have_hash, hash, false)) != PM_OK) return rc;
                have256 = t256.n_list > 0;
                if (have256) {
                    plan.push_back({HostPipe::kFetch, at, c2, c2, true});
                    at += c2;
                }
            }
            plan.push_back({-1, at, P - at, 0, false});  // route filled in once the probes are in
        } else {
            plan.push_back({by_size, 0, P, 0, false});
        }
    } else {
        // whole planes (asked for, or a route that does not apply: pageable memory for 1 / 2, planes that do
        // not split into whole blocks, a map that samples nothing)
        plan.push_back({HostPipe::kWhole, 0, P, 0, false});
    }

    // ---- device scratch for the largest segment, staging for the collected tables
    std::vector<SegLayout> lay(plan.size());
    size_t need = 2 * nmap * sizeof(double) + 512 + j.plane_bytes + nmap * sizeof(double);  // median redo
    size_t in_stage_need = 0;
    for (size_t i = 0; i < plan.size(); i++) {
        // (the remainder of an exploring call may take any route: size it for the most demanding)
        const int routes[4] = {HostPipe::kCollect, HostPipe::kWhole, HostPipe::kFetch, HostPipe::kHybrid};
        for (int r = 0; r < (plan[i].route < 0 ? 4 : 1); r++) {
            Segment sg = plan[i];
            if (sg.route < 0) sg.route = routes[r];
            if ((sg.route == HostPipe::kFetch || sg.route == HostPipe::kHybrid) && !have256) continue;
            const HostPipe::Table &tt = sg.route == HostPipe::kFetch ? t256 : t16;
            // (the split of a hybrid remainder is not known yet: size it for the widest fetched chunk)
            const SegLayout L = seg_layout(ctx, j, sg, tt.n_list, tt.shift, dst_pinned, t256.n_list);
            need = std::max(need, L.need);
            if (sg.route == HostPipe::kCollect || sg.route == HostPipe::kHybrid) in_stage_need = std::max(in_stage_need, L.slot_bytes);
            if (plan[i].route >= 0) lay[i] = L;
        }
    }
    rc = ensure_scratch(ctx, need);
    if (rc != PM_OK) return rc;
    if (in_stage_need) {
        rc = ensure_in_stage(ctx, hp, in_stage_need);
        if (rc != PM_OK) return rc;
    }
    const char *cube_dev = nullptr;  // zero copy: the device's view of the caller's pinned cube
    double *out_dev = nullptr;
    bool dev_view = false;  // some segment has the GPU read the caller's cube in place
    for (const Segment &sg : plan)
        dev_view = dev_view || sg.route == HostPipe::kInPlace || sg.route == HostPipe::kFetch || sg.route == HostPipe::kHybrid || (sg.route < 0 && have256);
    if (dev_view) PM_HIP(ctx, hipHostGetDevicePointer((void **)&cube_dev, (void *)cube, 0));
    if (device_out)
        out_dev = out;
    else if (dst_pinned)
        PM_HIP(ctx, hipHostGetDevicePointer((void **)&out_dev, (void *)out, 0));

    const double t_plan = now_ns();
    for (size_t i = 0; i < plan.size(); i++) {
        Segment &sg = plan[i];
        if (sg.n == 0) continue;
        if (sg.route < 0) {
            // the probes are in: the rest of this call, and every later call on this problem, by the fastest
            int best = -1;
            for (int r = 0; r < 4; r++)
                if (rs.ns_per_plane[r] > 0.0 && (best < 0 || rs.ns_per_plane[r] < rs.ns_per_plane[best])) best = r;
            rs.committed = best < 0 ? HostPipe::kCollect : best;
            rs.single = rs.committed;
            // ... or, later, by a hybrid of the collected and the fetched route, when the copy threads are the slower
            // leg of the collected one: with a share f of the planes fetched by the GPU (t_f per plane, all of it
            // link time) the threads collect the rest (t_c per plane) while the link also carries their table (t_d
            // per plane): both finish together at f = (t_c - t_d) / (t_c - t_d + t_f), in (1 - f) t_c per plane.
            // Where the probes predict a gain, the next whole calls alternate between the two and the faster stays.
            const double t_c = rs.collect_cpu_ns, t_d = rs.collect_dma_ns, t_f = rs.ns_per_plane[HostPipe::kFetch];
            if (have256 && t_c > 0.0 && t_f > 0.0 && t_c > 1.2 * t_d) {
                const double f = (t_c - t_d) / (t_c - t_d + t_f);
                const double t_h = (1.0 - f) * t_c;
                if (f >= 0.1 && best >= 0 && t_h < 0.95 * rs.ns_per_plane[best]) {
                    rs.hybrid_fetch_share = f;
                    rs.hybrid_fetch_ns = t_f;
                    rs.trial = true;
                }
            }
            sg.route = rs.committed;
            const HostPipe::Table &tt = sg.route == HostPipe::kFetch ? t256 : t16;
            lay[i] = seg_layout(ctx, j, sg, tt.n_list, tt.shift, dst_pinned, t256.n_list);
        }
        const HostPipe::Table *tt = sg.route == HostPipe::kFetch ? &t256 : &t16;
        const double t_begin = now_ns();
        double stage = 0.0, stage2[2] = {0.0, 0.0};
        double hy[4] = {rs.collect_cpu_ns, 0.0, rs.hybrid_fetch_ns > 0.0 ? rs.hybrid_fetch_ns : rs.ns_per_plane[HostPipe::kFetch], 0.0};
        rc = run_segment(ctx, hp, j, sg, lay[i], tt, dxm, dym, cube_dev, out_dev, sg.probe ? &stage : nullptr, stage2, &t256,
                         sg.route == HostPipe::kHybrid ? hy : nullptr);
        if (rc != PM_OK) return rc;
        const double wall = now_ns() - t_begin;
        if (sg.route == HostPipe::kHybrid && hy[1] > 0.0 && hy[3] > 0.0) {
            // the rates the two legs showed while they ran side by side (the next call's closing split starts from them)
            rs.collect_cpu_ns = hy[0] / hy[1];
            rs.hybrid_fetch_ns = hy[2] / hy[3];
            rs.hybrid_fetch_share = hy[3] / (hy[1] + hy[3]);
            if (trace)
                std::fprintf(stderr, "[pm hostpipe] hybrid: threads collected %.0f planes in %.3f ms (%.1f us each), GPU fetched %.0f in %.3f ms (%.1f us each)\n",
                             hy[1], hy[0] * 1e-6, hy[0] / hy[1] * 1e-3, hy[3], hy[2] * 1e-6, hy[2] / hy[3] * 1e-3);
        }
        if (mode < 0 && rs.trial && !exploring && plan.size() == 1 && P >= 6 * kHybridCollectChunk &&
            (sg.route == HostPipe::kHybrid || sg.route == rs.single)) {
            // three whole calls per side, judged by their medians (the calls run while other ranks share the link and the
            // CPUs: one lucky call must not fix the route for the life of the context)
            const int k = sg.route == HostPipe::kHybrid ? 1 : 0;
            if (rs.trial_n[k] < 3) rs.trial_samples[k][rs.trial_n[k]++] = wall / (double)sg.n;
            if (rs.trial_n[0] >= 3 && rs.trial_n[1] >= 3) {
                for (int q = 0; q < 2; q++) {
                    double *v = rs.trial_samples[q];
                    rs.trial_ns[q] = std::max(std::min(v[0], v[1]), std::min(std::max(v[0], v[1]), v[2]));
                }
                rs.trial = false;
                // (the hybrid must earn its keep: it also loads the link and the GPU's fetch path)
                rs.committed = rs.trial_ns[1] < 0.93 * rs.trial_ns[0] ? (int)HostPipe::kHybrid : rs.single;
                rs.committed_trial_ns = rs.trial_ns[rs.committed == HostPipe::kHybrid ? 1 : 0];
            }
        }
        ctx->last_cube_route = sg.route;
        if (sg.probe && sg.route == HostPipe::kCollect) {
            rs.collect_cpu_ns = stage2[0] / (double)sg.n;
            rs.collect_dma_ns = stage2[1] / (double)sg.n;
        }
        if (trace) std::fprintf(stderr, "[pm hostpipe] segment route %d planes %zu+%zu: %.3f ms%s (stage %.3f ms)\n", sg.route, sg.p0, sg.n,
                                wall * 1e-6, sg.probe ? " probe" : "", stage * 1e-6);
        if (mode < 0 && sg.route >= 0 && sg.route < HostPipe::kRoutes) {
            double &v = rs.ns_per_plane[sg.route];
            if (sg.probe)
                v = stage / (double)sg.n;
            else if (!exploring && sg.n >= 3 * (lay[i].chunk + lay[i].chunk_f))  // (a running mean once committed, from calls long enough to be pipelines)
                v = v > 0.0 ? 0.75 * v + 0.25 * wall / (double)sg.n : wall / (double)sg.n;
            // the route was committed on what the box looked like during its trial: when its running cost has moved 15 % away
            // from that (other ranks came or went), the next whole calls hold the trial again
            if (!sg.probe && !exploring && !rs.trial && rs.committed_trial_ns > 0.0 && sg.route == rs.committed && v > 1.15 * rs.committed_trial_ns &&
                rs.single >= 0 && P >= 6 * kHybridCollectChunk) {
                rs.trial = true;
                rs.trial_n[0] = rs.trial_n[1] = 0;
                rs.committed_trial_ns = 0.0;
            }
        }
    }
    const double t_segs = now_ns();
    // per-plane flags of the whole call: one read-back
    std::vector<int> hflags((size_t)n_planes);
    PM_HIP(ctx, hipMemcpyAsync(hflags.data(), ctx->flags, (size_t)n_planes * sizeof(int), hipMemcpyDeviceToHost, sk));
    PM_HIP(ctx, hipStreamSynchronize(sk));
    PM_HIP(ctx, hipStreamSynchronize(hp->s_in));
    ctx->checked_seq = a.seq;
    std::vector<int> redo;
    for (int p = 0; p < n_planes; p++)
        if (hflags[(size_t)p] == a.seq) redo.push_back(p);
    ctx->last_redo_planes = (int)redo.size();
    if (trace)
        std::fprintf(stderr, "[pm hostpipe] call %d planes: setup+tables %.3f ms, plan+scratch %.3f ms, segments %.3f ms, flags %.3f ms\n", n_planes,
                     (t_tables - t_call) * 1e-6, (t_plan - t_tables) * 1e-6, (t_segs - t_plan) * 1e-6, (now_ns() - t_segs) * 1e-6);
    if (!redo.empty()) {
        char *base = (char *)ctx->scratch;
        char *dplane = base + ((2 * nmap * sizeof(double) + 255) & ~(size_t)255);  // scratch was sized for one plane + one mapped plane behind the maps
        double *dout1 = (double *)(dplane + ((j.plane_bytes + 255) & ~(size_t)255));
        rc = redo_with_median(ctx, j, redo, dxm, dym, dplane, dout1);
        if (rc != PM_OK) return rc;
    }
    {
        const double t_end = now_ns();
        double *st = ctx->last_stage_ns;
        st[kStageTables] = t_tables - t_call;
        st[kStagePlan] = t_plan - t_tables;
        st[kStageFinish] = t_end - t_segs;
        st[kStageTotal] = t_end - t_call;
    }
    return PM_OK;
}

}  // namespace pmh
