// pm_host.hip.h -- declarations shared by the host-side sources of libplanetmapper_hip.so
// (pm_capi.hip: context + C ABI; pm_reproject.hip: the host logic of pm_map_cube).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pm_device.hip.h"



void pm_launch_disc(const pm::Params &p, int flags, hipStream_t s);
void pm_launch_disc_spheroid(const pm::Params &p, int flags, hipStream_t s);
void pm_launch_sky(const pm::Params &p, bool limb, hipStream_t s);
void pm_launch_map(const pm::Params &p, const double *lon, const double *lat, bool general, hipStream_t s);
void pm_launch_map_xy(const pm::Params &p, const double *lon, const double *lat, hipStream_t s);
void pm_launch_transform(const pm::Params &p, const pm::TransformArgs &t, hipStream_t s);
void pm_launch_radec_query(const pm::Params &p, const double *ra, const double *dec, unsigned long long n,
                           int ring_only_visible, double *out, bool b0, hipStream_t s);
void pm_launch_spline(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, int dtype, pm::PlaneStats *stats, unsigned int *hist,
                      hipStream_t s, int stage);
void pm_launch_reproject(const pm::ReprojectArgs &a, int dtype, hipStream_t s);
void pm_launch_mark_blocks(const pm::ReprojectArgs &a, unsigned char *flags, int shift, int dtype, hipStream_t s);
void pm_launch_number_blocks(const unsigned char *flags, size_t n_blk, int *tile_sums, int *blkmap, int *blklist, int *total,
                             hipStream_t s);
void pm_launch_hash_maps(const double *x_map, const double *y_map, int n, unsigned long long *out, hipStream_t s);
void pm_launch_reproject_blocks(const pm::ReprojectArgs &a, const pm::BlockTable &t, int dtype, hipStream_t s, bool fetch);
void pm_launch_mapped_data(const pm::Params &p, const pm::ReprojectArgs &a, const double *lon, const double *lat, double *xo,
                           double *yo, int dtype, hipStream_t s);
dim3 pm_smooth_grid(int n_map, int n_planes);
void pm_launch_reproject_smooth(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, int dtype, unsigned *redo, hipStream_t s);
void pm_launch_map_limits(const double *x_map, const double *y_map, int n, double *limits, hipStream_t s);
void pm_launch_clean(const pm::ReprojectArgs &a, double *work, int dtype, hipStream_t s);
void pm_launch_clean_lazy(const pm::ReprojectArgs &a, double *work, int dtype, pm::PlaneStats *stats, unsigned int *hist, hipStream_t s);
void pm_launch_plane_medians(const void *cube, int dtype, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                             unsigned int *hist, hipStream_t s);

// Debug / A-B knobs of the environment are honoured ONLY with PM_DEBUG_ENV=1 beside them: a stray PM_LT_MODE in a
// user's shell must not select another algorithm. Every one of them has a pm_set_option() form; the tools under
// tools/ set PM_DEBUG_ENV themselves. (PM_RCCL_LIBRARY and the launcher's LOCAL_WORLD_SIZE are deployment
// settings, not debug knobs, and are read as they are.)
inline const char *pm_debug_env(const char *name)
{
    const char *gate = std::getenv("PM_DEBUG_ENV");
    return (gate && gate[0] == '1') ? std::getenv(name) : nullptr;
}

namespace pmh { struct HostPipe; }

struct pm_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool have_geometry = false;
    bool have_disc = false;
    pm_geometry geometry{};
    pm_disc disc{};
    std::string error;
    // grow-only device scratch for host-buffer callers
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int *flags = nullptr;  // per-plane flags of pm_map_cube
    size_t flags_count = 0;
    pm::PlaneStats *stats = nullptr;  // per-plane nanmedian state (NaN pre-clean)
    unsigned int *hist = nullptr;
    size_t stats_count = 0;
    // device-mode pm_map_cube is asynchronous: planes that turn out to need the nanmedian
    // are finished by pm_synchronize(), which replays the call with the statistics
    bool pending = false;
    pm::ReprojectArgs pending_args{};
    int pending_dtype = 0;
    // spline reprojection: coefficient workspace + per-axis knots / LU (cached per (n, k))
    double *work = nullptr;
    size_t work_bytes = 0;
    struct AxisCache {
        int n = 0, k = 0;
        double *t = nullptr, *lu = nullptr;
    } axis[2];
    // 'smooth' interpolation options (map_img smooth_oversample_by / smooth_max_oversampled_img_size)
    int smooth_oversample_by = 5;
    int smooth_max_size = 10000;
    double *limits = nullptr;  // nanmin / nanmax of the x and y maps (4 doubles), then the partial values of pm_launch_map_limits
    double spline_smoothing = 0.0;  // map_img spline_smoothing (FITPACK s), 0 = interpolating splines
    // smoothing-spline fits (pm_smoothing.hip): descriptors + per-plane workspace of the planes fitted together
    void *sm_arena = nullptr;
    size_t sm_arena_bytes = 0;
    int *sm_status_host = nullptr;  // pinned: the per-round read-back (smoothing splines); [0] also the interpolating splines' "a plane asked for its median"
    hipEvent_t spline_ev = nullptr; // after the axis-0 solve of reproject_spline_resident
    int last_sm_knife_edges = 0;    // PM_OPT_LAST_SM_KNIFE_EDGES
    int last_sm_ill_conditioned = 0;  // PM_OPT_LAST_SM_ILL_CONDITIONED
    size_t sm_lds_limit = 0;        // dynamic LDS the smoothing kernels may ask for on THIS context's device (0: not asked yet)
    int spline_segment = 0;         // PM_OPT_SPLINE_SEGMENT: 0 = the library's choice, -1 = never, n = samples per segment
    int last_spline_segment = 0;    // PM_OPT_LAST_SPLINE_SEGMENT
    int sm_batch_planes = 0;        // PM_OPT_SM_BATCH_PLANES: 0 = as many as the workspace budget holds
    int map_seq = 0;        // sequence number of the latest pm_map_cube call
    int checked_seq = 0;    // calls up to this number have had their flags examined
    bool force_general = false;   // PM_OPT_GENERAL_KERNEL: never take the spheroid fast path
    int last_disc_kernel = 0;     // PM_OPT_LAST_DISC_KERNEL
    // pipelined host path (pm_hostpipe.hip): options + lazily created state
    size_t host_chunk_bytes = (size_t)32 << 20;
    int host_copy_threads = 0;   // 0: the library's choice (pm_hostpipe.hip)
    int zero_copy = -1;          // PM_OPT_ZERO_COPY
    int sparse_frame = -1;       // PM_OPT_SPARSE_FRAME
    int table_cache = 1;         // PM_OPT_BLOCK_TABLE_CACHE
    int trace = 0;               // PM_OPT_TRACE: 1 stage times of the host path, 2 the smoothing-spline search, on stderr
    int fuse_planes = 0;         // PM_OPT_FUSE_PLANES
    int lt_mode = 0;             // PM_OPT_LT_MODE (A/B runs of tools/, the light-time test): 0 closed-form light time of the
                                 // spheroid kernel, 1 the reference's sequence of epochs, 2 Newton step on its seed
    int route_explore = 1;       // PM_OPT_ROUTE_EXPLORE
    int fetch_shift = 7;         // PM_OPT_FETCH_BLOCK_BYTES: log2 of the blocks the GPU fetches from a pinned cube (routes 2, 4)
    int last_cube_route = -1;    // PM_OPT_LAST_CUBE_ROUTE
    int last_lt_path = 0;        // PM_OPT_LAST_LT_PATH
    // PM_OPT_LAST_STAGE_NS + k: where the latest host-fed pm_map_cube / pm_map_cube_sharded of this context spent its time, ns
    // (include/planetmapper_hip.h lists the stages; kStage* below)
    double last_stage_ns[16] = {};
    int last_redo_planes = 0;    // PM_OPT_LAST_REDO_PLANES: planes of the latest finished pm_map_cube redone with their nanmedian
    // pm_set_chunk_callback: told, on the calling thread, each time the kernels of further planes of a
    // nearest / linear pm_map_cube have been ENQUEUED on the context stream (planes arrive in order)
    void (*chunk_cb)(void *, int, int) = nullptr;
    void *chunk_user = nullptr;
    pmh::HostPipe *pipe = nullptr;
};

namespace pmh {

enum Stage : int {
    kStageTotal = 0, kStageTables, kStagePlan, kStageFirstFill, kStageCollect, kStageIssue, kStageDrain, kStageFinish,
    kStageDmaDevice, kStageKernelDevice, kStageExchangeExposed, kStageAgreement, kStageShardedTotal, kStageCount
};

// records the message in the context and returns `code`
int fail(pm_ctx *ctx, int code, const char *fmt, ...);

#define PM_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return pmh::fail(ctx, PM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                             \
    } while (0)

size_t dtype_size(int dtype);
int ensure_scratch(pm_ctx *ctx, size_t bytes);  // grow-only device buffers of the context
int ensure_flags(pm_ctx *ctx, size_t count);
int ensure_stats(pm_ctx *ctx, size_t count);
void fill_params(const pm_ctx *ctx, double alt, pm::Params &p);

// pm_reproject.hip: reprojection of planes resident on the device
int reproject_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, bool sync_now);
int finish_reproject(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype);
int reproject_spline_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols);
int reproject_smooth_resident(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype, const double *limits);
int reproject_smoothing_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols, double s);  // pm_smoothing.hip
int ensure_work(pm_ctx *ctx, size_t bytes);


// pm_hostpipe.hip: the host <-> HBM leg of PM_MEM_HOST calls
void pipe_destroy(pm_ctx *ctx);
// counters / measurements of the host pipe for pm_get_option (0 when the pipe does not exist yet)
long pipe_table_hits(const pm_ctx *ctx);
long pipe_route_ns_per_plane(const pm_ctx *ctx, int route);
int pipe_copy_threads(const pm_ctx *ctx);
long pipe_hybrid_fetch_permille(const pm_ctx *ctx);
void pipe_reset_route_stats(pm_ctx *ctx);
bool host_is_pinned(const void *p, size_t bytes);
// dst_host <- src_dev on `stream` (staged through pinned buffers + copy threads for pageable
// destinations); d2h_finish completes every issued copy and synchronises the stream
int d2h_issue(pm_ctx *ctx, hipStream_t stream, void *dst_host, const void *src_dev, size_t bytes);
// the same for a frame plane that is NaN outside a circle: only bands around the circle are copied
int d2h_issue_disc(pm_ctx *ctx, hipStream_t stream, double *dst_host, const double *src_dev, size_t nx, size_t n_rows, double y_first,
                   double x0, double y0, double r2);
int d2h_finish(pm_ctx *ctx, hipStream_t stream);
// after an error in the middle of a host-buffer call: waits until nothing of it is running any more
void pipe_abort(pm_ctx *ctx);
int map_cube_host_pipelined(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *x_map,
                            const double *y_map, size_t nmap, pm::ReprojectArgs a, double *out, bool device_out);

}  // namespace pmh
