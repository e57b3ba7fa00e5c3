// pm_reproject.hip -- host side of pm_map_cube: the reprojection of planes resident on the
// device for every interpolation mode of BodyXY.map_img (body_xy.py:1414-1904). Kernels:
// pm_kernels_reproject.hip. Host work here is set-up only: knots + banded factors of the spline
// axes, the footprint / oversampling grid of 'smooth', (the smoothing splines: pm_smoothing.hip);
// all per-pixel arithmetic runs on the GPU.
#include "pm_host.hip.h"

namespace pmh {

// Knots and banded LU of the B-spline collocation matrix of one axis (n unit-spaced samples,
// degree k, FITPACK's s = 0 knot placement), uploaded once per (n, k).
int ensure_axis(pm_ctx *ctx, int which, int n, int k, pm::SplineAxis &out)
{
    pm_ctx::AxisCache &ac = ctx->axis[which];
    if (ac.n != n || ac.k != k) {
        const int w = 2 * k + 1;
        std::vector<double> t((size_t)n + k + 1, 0.0), a((size_t)n * w, 0.0);
        for (int i = 0; i <= k; i++) t[n + i] = (double)(n - 1);
        const int k3 = k / 2;
        for (int l = 0; l < n - k - 1; l++) {
            const int j = k3 + 1 + l;
            t[k + 1 + l] = (k3 * 2 == k) ? 0.5 * ((double)j + (double)(j - 1)) : (double)j;
        }
        for (int i = 0; i < n; i++) {
            int l = k;
            while (l < n - 1 && (double)i >= t[l + 1]) l++;
            double h[6], hh[6];
            h[0] = 1.0;
            for (int j = 1; j <= k; j++) {  // fpbspl
                for (int q = 0; q < j; q++) hh[q] = h[q];
                h[0] = 0.0;
                for (int q = 1; q <= j; q++) {
                    const int li = l + q, lj = li - j;
                    const double f = hh[q - 1] / (t[li] - t[lj]);
                    h[q - 1] += f * (t[li] - (double)i);
                    h[q] = f * ((double)i - t[lj]);
                }
            }
            for (int q = 0; q <= k; q++) {
                const int j = l - k + q;
                a[(size_t)i * w + (j - i + k)] = h[q];
            }
        }
        for (int p = 0; p < n; p++) {  // banded LU, no pivoting (totally positive matrix)
            const double piv = a[(size_t)p * w + k];
            for (int i = p + 1; i <= p + k && i < n; i++) {
                const double f = a[(size_t)i * w + (p - i + k)] / piv;
                if (f == 0.0) continue;
                a[(size_t)i * w + (p - i + k)] = f;
                for (int j = p + 1; j <= p + k && j < n; j++) a[(size_t)i * w + (j - i + k)] -= f * a[(size_t)p * w + (j - p + k)];
            }
        }
        // the back substitution multiplies by the reciprocal of the diagonal (a division would sit in the chain of every step)
        for (int p = 0; p < n; p++) a[(size_t)p * w + k] = 1.0 / a[(size_t)p * w + k];
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ac.t) PM_HIP(ctx, hipFree(ac.t));
        if (ac.lu) PM_HIP(ctx, hipFree(ac.lu));
        ac = pm_ctx::AxisCache();
        if (hipMalloc((void **)&ac.t, t.size() * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&ac.lu, a.size() * sizeof(double)) != hipSuccess)
            return fail(ctx, PM_ERR_ALLOC, "hipMalloc of spline factors failed");
        PM_HIP(ctx, hipMemcpy(ac.t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
        PM_HIP(ctx, hipMemcpy(ac.lu, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
        ac.n = n;
        ac.k = k;
    }
    out.t = ac.t;
    out.lu = ac.lu;
    out.n = n;
    out.k = k;
    return PM_OK;
}

int ensure_work(pm_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->work_bytes) return PM_OK;
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->work) PM_HIP(ctx, hipFree(ctx->work));
    ctx->work = nullptr;
    ctx->work_bytes = 0;
    hipError_t e = hipMalloc((void **)&ctx->work, bytes);
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipMalloc(%zu) of the spline workspace failed", bytes);
    ctx->work_bytes = bytes;
    return PM_OK;
}

// get_xy_pchip body_xy.py:1724-1741: original pixel coordinates within `pad` of the map's
// footprint [lo, hi], refined by the largest factor <= oversample_by whose grid fits
// max_size. Returns false if no pixel is in range or the grid has a single point.
bool smooth_axis(int n, double lo, double hi, int oversample_by, int max_size, pm::SmoothAxis &ax)
{
    const double pad = 5.0;  // limit_padding
    int first = -1, last = -1;
    for (int j = 0; j < n; j++)
        if ((double)j >= lo - pad && (double)j <= hi + pad) {
            if (first < 0) first = j;
            last = j;
        }
    if (first < 0) return false;
    const long old_size = last - first + 1;
    ax.first = first;
    ax.last = last;
    ax.num = (int)old_size;
    ax.oversampled = 0;
    ax.step = 1.0;
    for (long o = oversample_by; o > 1; o--) {
        const long num = old_size * o - (o - 1);
        if (num <= max_size) {
            if (num > 1) {
                ax.num = (int)num;
                ax.oversampled = 1;
                ax.step = ((double)last - (double)first) / (double)(num - 1);  // numpy.linspace
            }
            break;
        }
    }
    return ax.num >= 2;
}

// 'smooth' reprojection of planes resident on the device. `limits`: nanmin / nanmax of the
// x and y maps (host values).
int reproject_smooth_resident(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype, const double *limits)
{
    if (!(limits[0] <= limits[1]) || !(limits[2] <= limits[3])) {
        // no visible map cell: every output is NaN (the reference only gets this far for
        // all-NaN planes; for others its own axis trimming fails with an IndexError)
        std::vector<double> nanrow((size_t)a.n_map, std::nan(""));
        for (int p = 0; p < a.n_planes; p++)
            PM_HIP(ctx, hipMemcpyAsync(a.out + (size_t)p * a.n_map, nanrow.data(), (size_t)a.n_map * sizeof(double),
                                       hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PM_OK;
    }
    pm::SmoothArgs sm;
    if (!smooth_axis(a.nx, limits[0], limits[1], ctx->smooth_oversample_by, ctx->smooth_max_size, sm.x) ||
        !smooth_axis(a.ny, limits[2], limits[3], ctx->smooth_oversample_by, ctx->smooth_max_size, sm.y))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT,
                    "smooth interpolation needs at least two image pixels per axis near the mapped region");
    sm.general = ctx->force_general ? 1 : 0;
    sm.planes_per_lane = 1;
    const size_t plane_bytes = (size_t)a.ny * a.nx * dtype_size(dtype);
    for (int p0 = 0; p0 < a.n_planes; p0 += 32768) {  // (blockIdx.y: at most 65535 groups of planes per launch)
        pm::ReprojectArgs b = a;
        b.n_planes = std::min(32768, a.n_planes - p0);
        b.cube = (const char *)a.cube + (size_t)p0 * plane_bytes;
        b.out = a.out + (size_t)p0 * a.n_map;
        const dim3 grid = pm_smooth_grid(b.n_map, b.n_planes);
        // the list of the workgroups with cells left to the gap-aware kernel
        const int rc = ensure_work(ctx, (1 + (size_t)grid.x * grid.y) * sizeof(unsigned));
        if (rc != PM_OK) return rc;
        pm_launch_reproject_smooth(b, sm, dtype, (unsigned *)ctx->work, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}


// Spline reprojection of planes resident on the device (plane chunks bound the workspace).
int reproject_spline_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols)
{
    const size_t plane_elems = (size_t)a.ny * a.nx;
    pm::SplineArgs sa;
    int rc = ensure_axis(ctx, 0, a.ny, k_rows, sa.rows);
    if (rc != PM_OK) return rc;
    rc = ensure_axis(ctx, 1, a.nx, k_cols, sa.cols);
    if (rc != PM_OK) return rc;
    size_t chunk = (size_t)(2ull << 30) / (plane_elems * sizeof(double));
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)a.n_planes) chunk = (size_t)a.n_planes;
    if (chunk > 32768) chunk = 32768;
    // Few, large planes: with one lane per line the chunk is `chunk * n / 64` waves, each a chain of 2 n dependent steps. Under
    // one wave per SIMD the lines are cut into segments (k_spline_seg_*: the same steps on the same operands, a second
    // buffer) - as many as bring the launch to ~2 waves per SIMD, of 64 .. 256 samples (measured, profiles/EXPERIMENTS_r06.md:
    // shorter segments re-read more warm-up samples, longer ones leave the chain long; from one wave per SIMD up the
    // line solves are as fast or faster).
    auto segment = [&](int lines, int n) {
        if (ctx->spline_segment < 0) return 0;
        if (ctx->spline_segment > 0) return std::min((ctx->spline_segment + 15) / 16 * 16, (n + 15) / 16 * 16);
        const long waves = (long)chunk * ((lines + pm::kSolveLines - 1) / pm::kSolveLines);
        if (waves >= pm::kSolveFillWaves || n < 256) return 0;
        const long pieces = (2 * pm::kSolveFillWaves + waves - 1) / waves;
        const int seg = (int)(((n + pieces - 1) / pieces + 15) / 16 * 16);
        return std::min(std::max(seg, 64), 256);
    };
    sa.seg_rows = segment(a.nx, a.ny);
    sa.seg_cols = segment(a.ny, a.nx);
    // (one form for both axes: the segmented passes go from buffer to buffer)
    if (!sa.seg_rows != !sa.seg_cols) {
        if (!sa.seg_rows) sa.seg_rows = (a.ny + 15) / 16 * 16;
        if (!sa.seg_cols) sa.seg_cols = (a.nx + 15) / 16 * 16;
    }
    ctx->last_spline_segment = sa.seg_rows;
    const size_t chunk_bytes = chunk * plane_elems * sizeof(double);
    rc = ensure_work(ctx, chunk_bytes * (sa.seg_rows ? 2 : 1));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, chunk + 1);
    if (rc != PM_OK) return rc;
    if (!ctx->sm_status_host) PM_HIP(ctx, hipHostMalloc((void **)&ctx->sm_status_host, 4 * sizeof(int)));
    if (!ctx->spline_ev) PM_HIP(ctx, hipEventCreateWithFlags(&ctx->spline_ev, hipEventDisableTiming));
    sa.work = ctx->work;
    if (sa.seg_rows) sa.work2 = ctx->work + chunk * plane_elems;
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += chunk) {
        const int np = (int)std::min(chunk, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        // The second round (nanmedian of the planes with a pixel that has no finite neighbour, their solves again) is
        // eighteen launches that leave at once for most data - 80 us of the 0.5 ms a plane alone takes. Whether any plane
        // asked is known after the axis-0 solve: its flag comes back while the axis-1 solve runs.
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)(np + 1) * sizeof(pm::PlaneStats), ctx->stream));
        pm_launch_spline(b, sa, dtype, ctx->stats, ctx->hist, ctx->stream, 0);
        PM_HIP(ctx, hipMemcpyAsync(ctx->sm_status_host, &ctx->stats[np].needs_median, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipEventRecord(ctx->spline_ev, ctx->stream));
        pm_launch_spline(b, sa, dtype, ctx->stats, ctx->hist, ctx->stream, 1);
        PM_HIP(ctx, hipEventSynchronize(ctx->spline_ev));
        if (ctx->sm_status_host[0]) {
            PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), ctx->stream));
            pm_launch_spline(b, sa, dtype, ctx->stats, ctx->hist, ctx->stream, 2);
        }
        pm_launch_spline(b, sa, dtype, ctx->stats, ctx->hist, ctx->stream, 3);
    }
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}

// Reproject `a.n_planes` planes that are resident on the device. First pass without plane
// statistics; if a plane reports that it needs its nanmedian (flags bit 1), the medians are
// computed and the planes are mapped again. `sync_now`: examine the flags immediately
// (host-buffer callers) or leave that to pm_synchronize().
int reproject_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, bool sync_now);

int finish_reproject(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype)
{
    std::vector<int> hflags(a.n_planes);
    PM_HIP(ctx, hipMemcpyAsync(hflags.data(), a.plane_flags, (size_t)a.n_planes * sizeof(int), hipMemcpyDeviceToHost,
                               ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool any = false, stale = false;
    int n_redo = 0;
    for (int f : hflags) {
        any = any || (f == a.seq);
        n_redo += (f == a.seq);
        stale = stale || (f > ctx->checked_seq && f < a.seq);
    }
    ctx->checked_seq = a.seq;
    ctx->last_redo_planes = n_redo;
    if (stale)
        return fail(ctx, PM_ERR_STATE,
                    "an earlier asynchronous pm_map_cube call sampled pixels that need the plane nanmedian (+-inf "
                    "or all-NaN neighbourhoods) and was superseded before pm_synchronize(); synchronize after each "
                    "call for such data");
    if (!any) return PM_OK;
    int rc = ensure_stats(ctx, (size_t)a.n_planes);
    if (rc != PM_OK) return rc;
    size_t plane_elems = (size_t)a.ny * a.nx;
    PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)a.n_planes * sizeof(pm::PlaneStats), ctx->stream));
    PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)a.n_planes * 512 * sizeof(unsigned int), ctx->stream));
    for (int p0 = 0; p0 < a.n_planes; p0 += 32768) {
        int np = std::min(32768, a.n_planes - p0);
        pm_launch_plane_medians((const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype), dtype, np, plane_elems,
                                ctx->stats + p0, ctx->hist + (size_t)p0 * 512, ctx->stream);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + (size_t)p0 * a.n_map;
        b.plane_flags = a.plane_flags + p0;
        b.plane_stats = ctx->stats + p0;
        pm_launch_reproject(b, dtype, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int reproject_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, bool sync_now)
{
    size_t plane_elems = (size_t)a.ny * a.nx;
    a.plane_stats = nullptr;
    a.seq = ++ctx->map_seq;
    // blockIdx.y is limited to 65535 planes per launch
    for (int p0 = 0; p0 < a.n_planes; p0 += 32768) {
        pm::ReprojectArgs b = a;
        b.n_planes = std::min(32768, a.n_planes - p0);
        b.cube = (const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + (size_t)p0 * a.n_map;
        b.plane_flags = a.plane_flags + p0;
        pm_launch_reproject(b, dtype, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    if (ctx->chunk_cb) ctx->chunk_cb(ctx->chunk_user, 0, a.n_planes);
    if (sync_now) return finish_reproject(ctx, a, dtype);
    ctx->pending = true;
    ctx->pending_args = a;
    ctx->pending_dtype = dtype;
    return PM_OK;
}

}  // namespace pmh
