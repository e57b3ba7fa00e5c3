// pm_capi.hip -- C ABI of libplanetmapper_hip.so (see include/planetmapper_hip.h).
//
// Host side only: context management, kernel-argument block construction, launches,
// host<->device staging for callers that hand over host buffers. There is NO CPU
// compute path here: without a gfx950 device pm_create() fails.
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
void pm_launch_map(const pm::Params &p, const double *lon, const double *lat, hipStream_t s);
void pm_launch_transform(const pm::Params &p, const pm::TransformArgs &t, hipStream_t s);
void pm_launch_spline(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, int dtype, hipStream_t s);
void pm_launch_reproject(const pm::ReprojectArgs &a, int dtype, hipStream_t s);
void pm_launch_reproject_smooth(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, int dtype, hipStream_t s);
void pm_launch_map_limits(const double *x_map, const double *y_map, int n, double *limits, hipStream_t s);
void pm_launch_clean(const pm::ReprojectArgs &a, double *work, int dtype, hipStream_t s);
void pm_launch_sm_solve(const pm::SmoothFitAxis &ax, const double *in, size_t si, size_t sq, int nrhs, double *g,
                        double *c, hipStream_t s);
void pm_launch_transpose(const double *in, double *out, int rows, int cols, hipStream_t s);
void pm_launch_sm_resid(const pm::SmoothFitAxis &ay, const pm::SmoothFitAxis &ax, const double *z, const double *ct,
                        double *rowsum, double *colsum, hipStream_t s);
void pm_launch_sm_eval(const pm::ReprojectArgs &a, const pm::SmoothEvalArgs &e, int dtype, hipStream_t s);
void pm_launch_plane_medians(const void *cube, int dtype, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                             unsigned int *hist, hipStream_t s);

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
    double *limits = nullptr;  // 4 doubles: nanmin / nanmax of the x and y maps
    double spline_smoothing = 0.0;  // map_img spline_smoothing (FITPACK s), 0 = interpolating splines
    void *sm_arena = nullptr;       // device workspace of the smoothing-spline fit
    size_t sm_arena_bytes = 0;
    int map_seq = 0;        // sequence number of the latest pm_map_cube call
    int checked_seq = 0;    // calls up to this number have had their flags examined
    bool force_general = false;   // PM_FORCE_GENERAL=1: never take the spheroid fast path (testing)
};

namespace {

int fail(pm_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->error = buf;
    }
    return code;
}

#define PM_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, PM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// golden-ratio row stride, made coprime with the row count so the row map is a bijection
int32_t golden_stride(int rows)
{
    auto gcd = [](long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; };
    const long n = rows > 0 ? rows : 1;
    long s = (long)(0.6180339887498949 * n);
    if (s < 1) s = 1;
    while (gcd(s, n) != 1) s++;
    return (int32_t)s;
}

int ensure_scratch(pm_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return PM_OK;
    if (ctx->scratch) {
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PM_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->scratch, bytes);
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->scratch_bytes = bytes;
    return PM_OK;
}

int ensure_flags(pm_ctx *ctx, size_t count)
{
    if (count <= ctx->flags_count) return PM_OK;
    if (ctx->flags) {
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PM_HIP(ctx, hipFree(ctx->flags));
        ctx->flags = nullptr;
        ctx->flags_count = 0;
    }
    hipError_t e = hipMalloc((void **)&ctx->flags, count * sizeof(int));
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipMalloc flags failed: %s", hipGetErrorString(e));
    PM_HIP(ctx, hipMemsetAsync(ctx->flags, 0, count * sizeof(int), ctx->stream));
    ctx->flags_count = count;
    return PM_OK;
}

int ensure_stats(pm_ctx *ctx, size_t count)
{
    if (count <= ctx->stats_count) return PM_OK;
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stats) PM_HIP(ctx, hipFree(ctx->stats));
    if (ctx->hist) PM_HIP(ctx, hipFree(ctx->hist));
    ctx->stats = nullptr;
    ctx->hist = nullptr;
    ctx->stats_count = 0;
    if (hipMalloc((void **)&ctx->stats, count * sizeof(pm::PlaneStats)) != hipSuccess ||
        hipMalloc((void **)&ctx->hist, count * 512 * sizeof(unsigned int)) != hipSuccess)
        return fail(ctx, PM_ERR_ALLOC, "hipMalloc of plane statistics failed");
    ctx->stats_count = count;
    return PM_OK;
}

size_t dtype_size(int dtype)
{
    switch (dtype) {
    case PM_F64: return 8;
    case PM_F32: return 4;
    case PM_I16: return 2;
    case PM_I32: return 4;
    case PM_U8: return 1;
    case PM_U16: return 2;
    }
    return 0;
}

// Kernel-argument block for the current geometry + disc + altitude.
// A / Ai: BodyXY._get_xy2angular_matrix body_xy.py:354-373; r2: body_xy.py:3189-3203;
// K: Body._get_angular2km_matrix body.py:1625-1641; radii: body.py:172-229.
void fill_params(const pm_ctx *ctx, double alt, pm::Params &p)
{
    const pm_geometry &g = ctx->geometry;
    const pm_disc &d = ctx->disc;
    p.g = g;
    for (int i = 0; i < 3; i++) p.radii[i] = g.radii[i] + alt;
    double s = g.diameter_arcsec / (2.0 * d.r0);
    double th = -d.rotation_rad;
    double c = std::cos(th), sn = std::sin(th);
    double m00 = s * c, m01 = s * sn, m10 = s * -sn, m11 = s * c;
    p.A[0] = m00; p.A[1] = m01; p.A[2] = -(m00 * d.x0 + m01 * d.y0);
    p.A[3] = m10; p.A[4] = m11; p.A[5] = -(m10 * d.x0 + m11 * d.y0);
    double det = m00 * m11 - m01 * m10;
    double i00 = m11 / det, i01 = -m01 / det, i10 = -m10 / det, i11 = m00 / det;
    p.Ai[0] = i00; p.Ai[1] = i01; p.Ai[2] = -(i00 * p.A[2] + i01 * p.A[5]);
    p.Ai[3] = i10; p.Ai[4] = i11; p.Ai[5] = -(i10 * p.A[2] + i11 * p.A[5]);
    double rmax = std::fmax(p.radii[0], std::fmax(p.radii[1], p.radii[2]));
    double r = d.r0 * rmax / p.radii[0];
    double rc = r * 1.05 + 1.0;
    p.r2 = rc * rc;
    p.x0 = d.x0;
    p.y0 = d.y0;
    double ks = 1.0 / g.km_per_arcsec;
    double kc = std::cos(g.np_angle_rad), ksn = std::sin(g.np_angle_rad);
    double k00 = ks * kc, k01 = ks * ksn, k10 = -ks * ksn, k11 = ks * kc;
    double kdet = k00 * k11 - k01 * k10;
    p.K[0] = k11 / kdet; p.K[1] = -k01 / kdet; p.K[2] = -k10 / kdet; p.K[3] = k00 / kdet;
    p.t0 = g.et - g.lt_c;
    // spheroid fast-path constants (B0 frame)
    auto rot = [&](const double *v, double *o, double sgn) {
        for (int i = 0; i < 3; i++) o[i] = sgn * (g.R0[3 * i] * v[0] + g.R0[3 * i + 1] * v[1] + g.R0[3 * i + 2] * v[2]);
    };
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0.0;
            for (int k = 0; k < 3; k++) acc += g.R0[3 * i + k] * g.M[3 * j + k];  // R0 * M^T
            p.C[3 * i + j] = acc;
        }
    rot(g.T0, p.O0, -1.0);
    rot(g.VT, p.VB, 1.0);
    rot(g.AT, p.AB, 1.0);
    rot(g.S0, p.SB0, 1.0);
    rot(g.VS, p.VSB, 1.0);
    rot(g.AS, p.ASB, 1.0);
    rot(g.VO, p.VOB, 1.0);
    p.ira = 1.0 / p.radii[0];
    p.irc = 1.0 / p.radii[2];
    p.inv_c = 1.0 / g.clight;
    p.lat_k = (p.radii[0] / p.radii[2]) * (p.radii[0] / p.radii[2]);
    for (int i = 0; i < 3; i++) p.ir[i] = 1.0 / p.radii[i];
    {
        double m = std::fmin(p.radii[0], p.radii[2]);
        double a1 = m / p.radii[0], c1 = m / p.radii[2];
        p.limb_n[0] = a1 * a1;
        p.limb_n[1] = c1 * c1;
    }
    p.nx = d.nx;
    p.ny = d.ny;
    p.y_off = 0;
    p.rows = d.ny;
    p.row_stride = golden_stride(d.ny);
    p.pad_ = 0;
    p.optimize_speed = d.optimize_speed;
    p.n0 = p.n1 = 0;
    p.mask = 0;
    for (int i = 0; i < PM_NUM_PLANES; i++) p.out[i] = nullptr;
}

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
    pm_launch_reproject_smooth(a, sm, dtype, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}


// ------------------------------------------------------------------ smoothing splines (spline_smoothing > 0)
// BodyXY._do_spline_interpolation with s > 0 (body_xy.py:1673-1680) is FITPACK's `regrid`
// (Dierckx): grow the knot sets from the least-squares polynomial until the least-squares
// spline has a residual sum fp <= s, then find the smoothing parameter p with fp(p) = s by
// rational interpolation. The control flow (a few dozen scalar decisions per plane) and the
// QR factors of the two small banded design matrices stay on the host; every fit - two
// directional least-squares solves over all image columns / coefficient rows and the residual
// sums over all pixels - runs on the GPU.
struct SmAxis {
    int m = 0, k = 0, n = 0, nplus = 0;
    std::vector<double> t, fpint;
    std::vector<int> nrdata;
    std::vector<double> hb, R, Bp;  // tables for the current knots / p
    std::vector<int> lb, span;
    bool knots_changed = true;
    int nc() const { return n - k - 1; }
    int nrint() const { return n - 2 * k - 1; }

    void init(int m_, int k_)
    {
        m = m_; k = k_; n = 2 * (k + 1); nplus = 0;
        t.assign((size_t)m + k + 2, 0.0);
        for (int i = 0; i <= k; i++) t[k + 1 + i] = (double)(m - 1);
        fpint.assign((size_t)m + 1, 0.0);
        nrdata.assign((size_t)m + 1, 0);
        nrdata[0] = m - 2;
        knots_changed = true;
    }
    // fpknot: new knot at the middle data point of the interval with the largest residual sum
    void add_knot()
    {
        const int nri = nrint();
        double fpmax = 0.0;
        int number = -1, maxpt = 0, maxbeg = 0, jbegin = 1;
        for (int j = 0; j < nri; j++) {
            const int jp = nrdata[j];
            if (!(fpmax >= fpint[j] || jp == 0)) { fpmax = fpint[j]; number = j; maxpt = jp; maxbeg = jbegin; }
            jbegin += jp + 1;
        }
        if (number < 0) return;
        const int ihalf = maxpt / 2 + 1, nrx = maxbeg + ihalf;  // 1-based data index: abscissa nrx - 1
        for (int j = nri - 1; j > number; j--) { fpint[j + 1] = fpint[j]; nrdata[j + 1] = nrdata[j]; }
        for (int j = n - 1; j >= number + k + 1; j--) t[j + 1] = t[j];
        nrdata[number] = ihalf - 1;
        nrdata[number + 1] = maxpt - ihalf;
        fpint[number] = fpmax * (double)nrdata[number] / (double)maxpt;
        fpint[number + 1] = fpmax * (double)nrdata[number + 1] / (double)maxpt;
        t[number + k + 1] = (double)(nrx - 1);
        n += 1;
        knots_changed = true;
    }
    static void bspl(const double *t, int k, double x, int l, double *h)
    {
        double hh[6];
        h[0] = 1.0;
        for (int j = 1; j <= k; j++) {  // fpbspl
            for (int q = 0; q < j; q++) hh[q] = h[q];
            h[0] = 0.0;
            for (int q = 1; q <= j; q++) {
                const int li = l + q, lj = li - j;
                const double f = hh[q - 1] / (t[li] - t[lj]);
                h[q - 1] += f * (t[li] - x);
                h[q] = f * (x - t[lj]);
            }
        }
    }
    // B-spline values of every sample + knot interval of every integer abscissa
    void build_tables()
    {
        hb.assign((size_t)m * 6, 0.0);
        lb.assign((size_t)m, 0);
        span.assign((size_t)m, 0);
        int l = k;
        for (int i = 0; i < m; i++) {
            while (l < n - k - 2 && (double)i >= t[l + 1]) l++;
            bspl(t.data(), k, (double)i, l, &hb[(size_t)i * 6]);
            lb[i] = l - k;
            span[i] = l;
        }
    }
    // triangular band of the QR factor of [A; B / p] (Givens rotations, fpgivs / fprota), and
    // the scaled jump rows B / p themselves (fpdisc) for the refinement step on the device
    void factor(double p)
    {
        const int ncf = nc(), band = k + 2, nri = nrint();
        const int nb = (p > 0.0 && nri > 1) ? nri - 1 : 0;
        R.assign((size_t)ncf * pm::kSmBand, 0.0);
        Bp.assign((size_t)(nb > 0 ? nb : 1) * pm::kSmBand, 0.0);
        if (nb) {
            const double fac = (double)nri / (t[n - k - 1] - t[k]);
            for (int r = 0; r < nb; r++) {
                const int l = r + k + 1;
                for (int j = 0; j < band; j++) {
                    const int i = r + j;
                    double prod = 1.0;
                    bool first = true;
                    for (int q = 0; q < band; q++) {
                        if (i + q == l) continue;
                        const double h = t[l] - t[i + q];
                        prod = first ? h : prod * h * fac;
                        first = false;
                    }
                    Bp[(size_t)r * pm::kSmBand + j] = (t[i + k + 1] - t[i]) / prod / p;
                }
            }
        }
        for (int i = 0; i < m + nb; i++) {
            double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int j0;
            if (i < m) {
                for (int e = 0; e <= k; e++) h[e] = hb[(size_t)i * 6 + e];
                j0 = lb[i];
            } else {
                const int r = i - m;
                for (int e = 0; e < band; e++) h[e] = Bp[(size_t)r * pm::kSmBand + e];
                j0 = r;
            }
            for (int j = j0; j < ncf; j++) {
                const double piv = h[0];
                if (piv != 0.0) {
                    double *Rj = &R[(size_t)j * pm::kSmBand];
                    const double ww = Rj[0], store = std::fabs(piv);
                    const double dd = (store >= ww) ? store * std::sqrt(1.0 + (ww / piv) * (ww / piv))
                                                    : ww * std::sqrt(1.0 + (piv / ww) * (piv / ww));
                    const double cs = ww / dd, sn = piv / dd;
                    Rj[0] = dd;
                    for (int b = 1; b < band; b++) {
                        const double s1 = h[b], s2 = Rj[b];
                        Rj[b] = cs * s2 + sn * s1;
                        h[b] = cs * s1 - sn * s2;
                    }
                }
                bool any = false;
                for (int b = 0; b < band - 1; b++) { h[b] = h[b + 1]; any |= (h[b] != 0.0); }
                h[band - 1] = 0.0;
                if (!any) break;
            }
        }
    }
    // per-interval residual sums from per-sample sums: a sample on a knot gives half to each side
    void account(const double *sums)
    {
        const int nri = nrint();
        for (int j = 0; j < nri; j++) fpint[j] = 0.0;
        int old = 0;
        for (int i = 0; i < m; i++) {
            const int num = lb[i];
            fpint[num] += sums[i];
            if (num != old) { fpint[num] -= 0.5 * sums[i]; fpint[num - 1] += 0.5 * sums[i]; }
            old = num;
        }
    }
};

struct SmDevice {  // carve-up of ctx->sm_arena for one (ny, nx) plane
    double *U, *UT, *G, *CT, *hb_y, *hb_x, *R_y, *R_x, *Bp_y, *Bp_x, *rowsum, *colsum, *t_y, *t_x;
    int *lb_y, *lb_x, *span_y, *span_x;
};

int ensure_sm_arena(pm_ctx *ctx, int ny, int nx, SmDevice &d)
{
    const size_t npx = (size_t)ny * nx, mx = (size_t)std::max(ny, nx) + 8;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t oU = take(npx * 8), oUT = take(npx * 8), oG = take(npx * 8), oCT = take(npx * 8);
    const size_t ohy = take(mx * 6 * 8), ohx = take(mx * 6 * 8), oRy = take(mx * pm::kSmBand * 8), oRx = take(mx * pm::kSmBand * 8);
    const size_t oBy = take(mx * pm::kSmBand * 8), oBx = take(mx * pm::kSmBand * 8), ors = take(mx * 8), ocs = take(mx * 8);
    const size_t oty = take(mx * 8), otx = take(mx * 8), oly = take(mx * 4), olx = take(mx * 4), osy = take(mx * 4), osx = take(mx * 4);
    if (off > ctx->sm_arena_bytes) {
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->sm_arena) PM_HIP(ctx, hipFree(ctx->sm_arena));
        ctx->sm_arena = nullptr;
        ctx->sm_arena_bytes = 0;
        if (hipMalloc(&ctx->sm_arena, off) != hipSuccess)
            return fail(ctx, PM_ERR_ALLOC, "hipMalloc(%zu) of the smoothing-spline workspace failed", off);
        ctx->sm_arena_bytes = off;
    }
    char *b = (char *)ctx->sm_arena;
    d.U = (double *)(b + oU); d.UT = (double *)(b + oUT); d.G = (double *)(b + oG); d.CT = (double *)(b + oCT);
    d.hb_y = (double *)(b + ohy); d.hb_x = (double *)(b + ohx); d.R_y = (double *)(b + oRy); d.R_x = (double *)(b + oRx);
    d.Bp_y = (double *)(b + oBy); d.Bp_x = (double *)(b + oBx); d.rowsum = (double *)(b + ors); d.colsum = (double *)(b + ocs);
    d.t_y = (double *)(b + oty); d.t_x = (double *)(b + otx);
    d.lb_y = (int *)(b + oly); d.lb_x = (int *)(b + olx); d.span_y = (int *)(b + osy); d.span_x = (int *)(b + osx);
    return PM_OK;
}

// one fit for the current knots and p (p <= 0: least-squares spline); returns fp and updates
// the per-interval residual sums of both axes. z: cleaned plane on the device.
int sm_fit(pm_ctx *ctx, const double *z, SmAxis &ay, SmAxis &ax, double p, const SmDevice &d, double &fp,
           std::vector<double> &sums)
{
    hipStream_t s = ctx->stream;
    // the host vectors below are reused by the next fit: every upload is followed by a stream
    // synchronisation before they change (the D2H of the residual sums at the end of this fit)
    if (ay.knots_changed) {
        ay.build_tables();
        PM_HIP(ctx, hipMemcpyAsync(d.hb_y, ay.hb.data(), ay.hb.size() * 8, hipMemcpyHostToDevice, s));
        PM_HIP(ctx, hipMemcpyAsync(d.lb_y, ay.lb.data(), ay.lb.size() * 4, hipMemcpyHostToDevice, s));
        ay.knots_changed = false;
    }
    if (ax.knots_changed) {
        ax.build_tables();
        PM_HIP(ctx, hipMemcpyAsync(d.hb_x, ax.hb.data(), ax.hb.size() * 8, hipMemcpyHostToDevice, s));
        PM_HIP(ctx, hipMemcpyAsync(d.lb_x, ax.lb.data(), ax.lb.size() * 4, hipMemcpyHostToDevice, s));
        ax.knots_changed = false;
    }
    ay.factor(p);
    ax.factor(p);
    PM_HIP(ctx, hipMemcpyAsync(d.R_y, ay.R.data(), ay.R.size() * 8, hipMemcpyHostToDevice, s));
    PM_HIP(ctx, hipMemcpyAsync(d.R_x, ax.R.data(), ax.R.size() * 8, hipMemcpyHostToDevice, s));
    PM_HIP(ctx, hipMemcpyAsync(d.Bp_y, ay.Bp.data(), ay.Bp.size() * 8, hipMemcpyHostToDevice, s));
    PM_HIP(ctx, hipMemcpyAsync(d.Bp_x, ax.Bp.data(), ax.Bp.size() * 8, hipMemcpyHostToDevice, s));
    const int nby = (p > 0.0 && ay.nrint() > 1) ? ay.nrint() - 1 : 0, nbx = (p > 0.0 && ax.nrint() > 1) ? ax.nrint() - 1 : 0;
    pm::SmoothFitAxis fy = {d.hb_y, d.lb_y, d.R_y, d.Bp_y, ay.m, ay.k, ay.nc(), nby};
    pm::SmoothFitAxis fx = {d.hb_x, d.lb_x, d.R_x, d.Bp_x, ax.m, ax.k, ax.nc(), nbx};
    const int ny = ay.m, nx = ax.m, nr = ay.nc(), ncx = ax.nc();
    // along image rows for every image column: U (nr x nx)
    pm_launch_sm_solve(fy, z, (size_t)nx, 1, nx, d.G, d.U, s);
    // U' (nx x nr), then along image columns for every row coefficient: CT (ncx x nr)
    pm_launch_transpose(d.U, d.UT, nr, nx, s);
    pm_launch_sm_solve(fx, d.UT, (size_t)nr, 1, nr, d.G, d.CT, s);
    PM_HIP(ctx, hipMemsetAsync(d.rowsum, 0, (size_t)ny * 8, s));
    PM_HIP(ctx, hipMemsetAsync(d.colsum, 0, (size_t)nx * 8, s));
    pm_launch_sm_resid(fy, fx, z, d.CT, d.rowsum, d.colsum, s);
    PM_HIP(ctx, hipGetLastError());
    sums.resize((size_t)ny + nx);
    PM_HIP(ctx, hipMemcpyAsync(sums.data(), d.rowsum, (size_t)ny * 8, hipMemcpyDeviceToHost, s));
    PM_HIP(ctx, hipMemcpyAsync(sums.data() + ny, d.colsum, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    PM_HIP(ctx, hipStreamSynchronize(s));
    (void)ncx;
    fp = 0.0;
    for (int i = 0; i < ny; i++) fp += sums[i];
    if (std::getenv("PM_SM_DEBUG"))  // trace of the knot / smoothing-parameter search
        std::fprintf(stderr, "sm_fit ny=%d nx=%d knots=(%d,%d) p=%g fp=%.17g\n", ny, nx, ay.n, ax.n, p, fp);
    ay.account(sums.data());
    ax.account(sums.data() + ny);
    return PM_OK;
}

// FITPACK fpregr for one cleaned plane: on return ay / ax hold the knots and d.CT the coefficients
int sm_regrid(pm_ctx *ctx, const double *z, int ny, int nx, int k_rows, int k_cols, double s, SmAxis &ay, SmAxis &ax,
              const SmDevice &d)
{
    const double tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04;
    const int maxit = 20;
    const double acc = tol * s;
    ay.init(ny, k_rows);
    ax.init(nx, k_cols);
    const int nminy = 2 * (k_rows + 1), nminx = 2 * (k_cols + 1), nmaxy = ny + k_rows + 1, nmaxx = nx + k_cols + 1;
    std::vector<double> sums;
    int lastdi = 0, rc;
    bool poly = false, done = false;
    double fp = 0.0, fp0 = 0.0, fpold = 0.0, reducy = 0.0, reducx = 0.0, fpms = 0.0;
    // (FITPACK's "x" is the first array axis = image rows, "y" the image columns)
    for (int iter = 0; iter < ny + nx; iter++) {
        poly = (ay.n == nminy && ax.n == nminx);
        rc = sm_fit(ctx, z, ay, ax, -1.0, d, fp, sums);
        if (rc != PM_OK) return rc;
        if (poly) fp0 = fp;
        fpms = fp - s;
        if (std::fabs(fpms) < acc) { done = true; break; }
        if (fpms < 0.0) break;
        if (ay.n == nmaxy && ax.n == nmaxx) { done = true; break; }  // interpolating spline
        if (lastdi < 0) reducy = fpold - fp;
        else if (lastdi > 0) reducx = fpold - fp;
        fpold = fp;
        auto nplus = [&](const SmAxis &a, int nmin, double reduc) {
            if (a.n == nmin) return 1;
            int npl1 = a.nplus * 2;
            if (reduc > acc) npl1 = (int)((double)a.nplus * fpms / reduc);
            return std::min(a.nplus * 2, std::max(std::max(npl1, a.nplus / 2), 1));
        };
        const int nply = nplus(ay, nminy, reducy), nplx = nplus(ax, nminx, reducx);
        bool first_axis = (nply < nplx) || (nply == nplx && lastdi >= 0);
        if (first_axis && ay.n == nmaxy) first_axis = false;
        if (!first_axis && ax.n == nmaxx) first_axis = true;
        SmAxis &a = first_axis ? ay : ax;
        lastdi = first_axis ? -1 : 1;
        a.nplus = first_axis ? nply : nplx;
        const int nmax = first_axis ? nmaxy : nmaxx;
        for (int l = 0; l < a.nplus; l++) {
            a.add_knot();
            if (a.n == nmax) break;
        }
    }
    if (!done && !poly) {
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 1.0;
        bool ich1 = false, ich3 = false;
        for (int iter = 0; iter < maxit; iter++) {
            rc = sm_fit(ctx, z, ay, ax, p, d, fp, sums);
            if (rc != PM_OK) return rc;
            fpms = fp - s;
            if (std::fabs(fpms) < acc || iter == maxit - 1) break;
            const double p2 = p, f2 = fpms;
            if (!ich3) {
                if ((f2 - f3) <= acc) {  // initial p too large
                    p3 = p2; f3 = f2;
                    p *= con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    continue;
                }
                if (f2 < 0.0) ich3 = true;
            }
            if (!ich1) {
                if ((f1 - f2) <= acc) {  // initial p too small
                    p1 = p2; f1 = f2;
                    p /= con4;
                    if (p3 >= 0.0 && p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                }
                if (f2 > 0.0) ich1 = true;
            }
            if (f2 >= f1 || f2 <= f3) break;
            if (p3 > 0.0) {  // fprati
                const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
                p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
            } else {
                p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
            }
            if (f2 < 0.0) { p3 = p2; f3 = f2; } else { p1 = p2; f1 = f2; }
        }
    }
    return PM_OK;
}

// smoothing-spline reprojection of planes resident on the device
int reproject_smoothing_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols, double s)
{
    const size_t plane_elems = (size_t)a.ny * a.nx;
    SmDevice d;
    int rc = ensure_sm_arena(ctx, a.ny, a.nx, d);
    if (rc != PM_OK) return rc;
    size_t chunk = (size_t)(1ull << 30) / (plane_elems * sizeof(double));
    chunk = std::max<size_t>(1, std::min<size_t>(chunk, (size_t)a.n_planes));
    if (chunk > 32768) chunk = 32768;
    rc = ensure_work(ctx, chunk * plane_elems * sizeof(double));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, chunk);
    if (rc != PM_OK) return rc;
    std::vector<pm::PlaneStats> stats(chunk);
    std::vector<double> nan_row((size_t)a.n_map, std::nan(""));
    SmAxis ay, ax;
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += chunk) {
        const int np = (int)std::min(chunk, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)np * sizeof(pm::PlaneStats), ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), ctx->stream));
        pm_launch_plane_medians(b.cube, dtype, np, plane_elems, ctx->stats, ctx->hist, ctx->stream);
        pm_launch_clean(b, ctx->work, dtype, ctx->stream);
        PM_HIP(ctx, hipMemcpyAsync(stats.data(), ctx->stats, (size_t)np * sizeof(pm::PlaneStats), hipMemcpyDeviceToHost,
                                   ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int pl = 0; pl < np; pl++) {
            if (stats[pl].all_nan) {  // body_xy.py:1668-1670: the map of an all-NaN image is all NaN
                PM_HIP(ctx, hipMemcpyAsync(b.out + (size_t)pl * a.n_map, nan_row.data(), (size_t)a.n_map * 8,
                                           hipMemcpyHostToDevice, ctx->stream));
                continue;
            }
            const double *z = ctx->work + (size_t)pl * plane_elems;
            rc = sm_regrid(ctx, z, a.ny, a.nx, k_rows, k_cols, s, ay, ax, d);
            if (rc != PM_OK) return rc;
            PM_HIP(ctx, hipMemcpyAsync(d.t_y, ay.t.data(), (size_t)ay.n * 8, hipMemcpyHostToDevice, ctx->stream));
            PM_HIP(ctx, hipMemcpyAsync(d.t_x, ax.t.data(), (size_t)ax.n * 8, hipMemcpyHostToDevice, ctx->stream));
            PM_HIP(ctx, hipMemcpyAsync(d.span_y, ay.span.data(), (size_t)a.ny * 4, hipMemcpyHostToDevice, ctx->stream));
            PM_HIP(ctx, hipMemcpyAsync(d.span_x, ax.span.data(), (size_t)a.nx * 4, hipMemcpyHostToDevice, ctx->stream));
            pm::SmoothEvalArgs e = {d.CT, d.t_y, d.t_x, d.span_y, d.span_x, ay.nc(), ax.nc(), k_rows, k_cols, pl};
            pm_launch_sm_eval(b, e, dtype, ctx->stream);
            PM_HIP(ctx, hipGetLastError());
            PM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // knots / spans are reused by the next plane
        }
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
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
    rc = ensure_work(ctx, chunk * plane_elems * sizeof(double));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, chunk);
    if (rc != PM_OK) return rc;
    sa.work = ctx->work;
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += chunk) {
        const int np = (int)std::min(chunk, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)np * sizeof(pm::PlaneStats), ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), ctx->stream));
        pm_launch_plane_medians(b.cube, dtype, np, plane_elems, ctx->stats, ctx->hist, ctx->stream);
        pm_launch_spline(b, sa, dtype, ctx->stream);
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
    for (int f : hflags) {
        any = any || (f == a.seq);
        stale = stale || (f > ctx->checked_seq && f < a.seq);
    }
    ctx->checked_seq = a.seq;
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
    if (sync_now) return finish_reproject(ctx, a, dtype);
    ctx->pending = true;
    ctx->pending_args = a;
    ctx->pending_dtype = dtype;
    return PM_OK;
}

constexpr uint64_t bit(int p) { return ((uint64_t)1) << p; }
constexpr uint64_t kIllumBits = bit(PM_PHASE) | bit(PM_INCIDENCE) | bit(PM_EMISSION) | bit(PM_AZIMUTH);
constexpr uint64_t kStateBits = bit(PM_DISTANCE) | bit(PM_RADIAL_VELOCITY) | bit(PM_DOPPLER);
constexpr uint64_t kRingBits = bit(PM_RING_RADIUS) | bit(PM_RING_LON_GRAPHIC) | bit(PM_RING_DISTANCE);
constexpr uint64_t kLimbBits = bit(PM_LIMB_DISTANCE) | bit(PM_LIMB_LON_GRAPHIC) | bit(PM_LIMB_LAT_GRAPHIC);
constexpr uint64_t kSkyBits = bit(PM_RA) | bit(PM_DEC) | bit(PM_PIXEL_X) | bit(PM_PIXEL_Y) | bit(PM_KM_X) |
                              bit(PM_KM_Y) | bit(PM_ANGULAR_X) | bit(PM_ANGULAR_Y) | kLimbBits;
constexpr uint64_t kDiscBits = bit(PM_LON_GRAPHIC) | bit(PM_LAT_GRAPHIC) | bit(PM_LON_CENTRIC) |
                               bit(PM_LAT_CENTRIC) | bit(PM_LOCAL_SOLAR_TIME) | kIllumBits | kStateBits | kRingBits;
constexpr uint64_t kAllBits = (bit(PM_NUM_PLANES) - 1);

int check_ready(pm_ctx *ctx, bool need_disc)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    if (!ctx->have_geometry) return fail(ctx, PM_ERR_STATE, "pm_set_geometry has not been called");
    if (need_disc && !ctx->have_disc) return fail(ctx, PM_ERR_STATE, "pm_set_disc has not been called");
    PM_HIP(ctx, hipSetDevice(ctx->device));
    return PM_OK;
}

}  // namespace

extern "C" {

int pm_abi_version(void) { return PM_ABI_VERSION; }

int pm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

pm_ctx *pm_create(int device, int *status)
{
    auto set = [&](int s) {
        if (status) *status = s;
    };
    int n = pm_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        set(PM_ERR_NO_DEVICE);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // kernels are built for gfx950 only
        set(PM_ERR_NO_DEVICE);
        return nullptr;
    }
    pm_ctx *ctx = new (std::nothrow) pm_ctx();
    if (!ctx) {
        set(PM_ERR_ALLOC);
        return nullptr;
    }
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        set(PM_ERR_HIP);
        return nullptr;
    }
    ctx->own_stream = true;
    const char *fg = std::getenv("PM_FORCE_GENERAL");
    ctx->force_general = fg && fg[0] == '1';
    set(PM_OK);
    return ctx;
}

void pm_destroy(pm_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->flags) (void)hipFree(ctx->flags);
    if (ctx->work) (void)hipFree(ctx->work);
    if (ctx->limits) (void)hipFree(ctx->limits);
    if (ctx->sm_arena) (void)hipFree(ctx->sm_arena);
    for (auto &ac : ctx->axis) {
        if (ac.t) (void)hipFree(ac.t);
        if (ac.lu) (void)hipFree(ac.lu);
    }
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->hist) (void)hipFree(ctx->hist);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pm_last_error(const pm_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int pm_synchronize(pm_ctx *ctx)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->pending) {
        ctx->pending = false;
        return finish_reproject(ctx, ctx->pending_args, ctx->pending_dtype);
    }
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

void *pm_stream(pm_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pm_set_stream(pm_ctx *ctx, void *stream)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) PM_HIP(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
    return PM_OK;
}

int pm_device_malloc(pm_ctx *ctx, uint64_t bytes, void **dptr)
{
    if (!ctx || !dptr) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return PM_OK;
}

int pm_device_free(pm_ctx *ctx, void *dptr)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PM_HIP(ctx, hipFree(dptr));
    return PM_OK;
}

int pm_memcpy_h2d(pm_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_memcpy_d2h(pm_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_set_geometry(pm_ctx *ctx, const pm_geometry *geometry)
{
    if (!ctx || !geometry) return PM_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 3; i++)
        if (!(geometry->radii[i] > 0.0)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "radii must be positive");
    if (!(geometry->clight > 0.0) || !(geometry->diameter_arcsec > 0.0) || !(geometry->km_per_arcsec > 0.0))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "clight, diameter_arcsec and km_per_arcsec must be positive");
    ctx->geometry = *geometry;
    ctx->have_geometry = true;
    return PM_OK;
}

int pm_set_disc(pm_ctx *ctx, const pm_disc *disc)
{
    if (!ctx || !disc) return PM_ERR_INVALID_ARGUMENT;
    // BodyXY.set_x0/y0/r0/rotation body_xy.py:805-890, set_img_size :941-961
    if (!std::isfinite(disc->x0) || !std::isfinite(disc->y0) || !std::isfinite(disc->r0) ||
        !std::isfinite(disc->rotation_rad))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "disc parameters must be finite");
    if (!(disc->r0 > 0.0)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "r0 must be greater than zero");
    if (disc->nx < 0 || disc->ny < 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "nx and ny must be non-negative");
    // the image kernels address a row with a 32-bit byte offset and put rows on gridDim.y
    if (disc->nx > (1 << 28) || disc->ny > 65535)
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "image size is limited to nx <= 2^28, ny <= 65535");
    ctx->disc = *disc;
    ctx->have_disc = true;
    return PM_OK;
}

int pm_backplanes_img(pm_ctx *ctx, uint64_t plane_mask, double alt, double *const *out, int mem)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    return pm_backplanes_img_rows(ctx, plane_mask, alt, 0, ctx->disc.ny, out, mem);
}

int pm_backplanes_img_rows(pm_ctx *ctx, uint64_t plane_mask, double alt, int row_begin, int n_rows,
                           double *const *out, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if (!out) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "out is NULL");
    if (plane_mask & ~kAllBits) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown plane bit in mask");
    if (!std::isfinite(alt)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "alt must be finite");
    const pm_disc &d = ctx->disc;
    // BodyXY._make_empty_img body_xy.py:3166-3168
    if (d.nx <= 0 || d.ny <= 0)
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "nx and ny must be positive to create a backplane image");
    if (row_begin < 0 || n_rows < 0 || row_begin > d.ny - n_rows)
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "rows [%d, %d) are outside the image (ny = %d)", row_begin,
                    row_begin + n_rows, d.ny);
    if (plane_mask == 0 || n_rows == 0) return PM_OK;
    for (int i = 0; i < 3; i++)
        if (!(ctx->geometry.radii[i] + alt > 0.0))
            return fail(ctx, PM_ERR_INVALID_ARGUMENT, "radii + alt must be positive");
    size_t npx = (size_t)d.nx * n_rows;

    pm::Params p;
    fill_params(ctx, alt, p);
    p.mask = plane_mask;
    p.y_off = row_begin;
    p.rows = n_rows;
    p.row_stride = golden_stride(n_rows);
    int nreq = 0;
    for (int i = 0; i < PM_NUM_PLANES; i++)
        if ((plane_mask >> i) & 1) {
            if (!out[i]) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "out[%d] is NULL but plane %d is requested", i, i);
            nreq++;
        }
    if (mem == PM_MEM_DEVICE) {
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) p.out[i] = out[i];
    } else {
        rc = ensure_scratch(ctx, (size_t)nreq * npx * sizeof(double));
        if (rc != PM_OK) return rc;
        int k = 0;
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) p.out[i] = (double *)ctx->scratch + (size_t)(k++) * npx;
    }

    if (plane_mask & kDiscBits) {
        int flags = 0;
        if (plane_mask & kIllumBits) flags |= 1;
        if (plane_mask & kStateBits) flags |= 2;
        if (plane_mask & kRingBits) flags |= 4;
        pm::Params pd = p;
        pd.mask = plane_mask & kDiscBits;
        // spheroids (every planet in pck00010) take the rotation-free fast path, triaxial bodies
        // (most moons) its variant with one small rotation per light-time evaluation
        // the fast path assumes an observer well outside the body (|O0| > 2 radii in scaled
        // coordinates); anything else, e.g. a lander, goes through the general kernel
        double y2 = 0.0;
        for (int i = 0; i < 3; i++) y2 += (pd.O0[i] / pd.radii[i]) * (pd.O0[i] / pd.radii[i]);
        // ... and it does not carry the target's / Sun's acceleration over the light-time span of a
        // disc intercept (|d| <= R / c): fine while A (R/c)^2 / 2 is below 1e-12 of the smallest
        // radius (Jupiter: 6e-9 km of 66 854 km; the ray itself is rounded at 1e-7 km)
        const double rmax = std::fmax(pd.radii[0], std::fmax(pd.radii[1], pd.radii[2]));
        const double rmin = std::fmin(pd.radii[0], std::fmin(pd.radii[1], pd.radii[2]));
        const double span = rmax / ctx->geometry.clight;
        double acc2 = 0.0, accs2 = 0.0;
        for (int i = 0; i < 3; i++) {
            acc2 += ctx->geometry.AT[i] * ctx->geometry.AT[i];
            accs2 += ctx->geometry.AS[i] * ctx->geometry.AS[i];
        }
        const bool slow = 0.5 * std::sqrt(std::fmax(acc2, accs2)) * span * span < 1e-12 * rmin;
        // a triaxial body is turned by its spin angle per light-time evaluation with a short series
        const bool small_spin = std::fabs(ctx->geometry.wdot) * span < 1e-3;
        const bool spheroid = y2 > 4.0 && slow && small_spin && !ctx->force_general;
        if (spheroid)
            pm_launch_disc_spheroid(pd, flags, ctx->stream);
        else
            pm_launch_disc(pd, flags, ctx->stream);
    }
    if (plane_mask & kSkyBits) {
        pm::Params ps = p;
        ps.mask = plane_mask & kSkyBits;
        pm_launch_sky(ps, (plane_mask & kLimbBits) != 0, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());

    if (mem != PM_MEM_DEVICE) {
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1)
                PM_HIP(ctx, hipMemcpyAsync(out[i], p.out[i], npx * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PM_OK;
}

int pm_backplanes_map(pm_ctx *ctx, uint64_t plane_mask, const double *lon_deg, const double *lat_deg, int n0, int n1,
                      double alt, double *const *out, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if (!out || !lon_deg || !lat_deg) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (plane_mask & ~kAllBits) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown plane bit in mask");
    if (n0 < 0 || n1 < 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "negative map shape");
    if (!std::isfinite(alt)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "alt must be finite");
    size_t n = (size_t)n0 * n1;
    if (n == 0 || plane_mask == 0) return PM_OK;
    pm::Params p;
    fill_params(ctx, alt, p);
    p.mask = plane_mask;
    p.n0 = n0;
    p.n1 = n1;
    int nreq = 0;
    for (int i = 0; i < PM_NUM_PLANES; i++)
        if ((plane_mask >> i) & 1) {
            if (!out[i]) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "out[%d] is NULL but plane %d is requested", i, i);
            nreq++;
        }
    const double *dlon = lon_deg, *dlat = lat_deg;
    if (mem == PM_MEM_DEVICE) {
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) p.out[i] = out[i];
    } else {
        rc = ensure_scratch(ctx, (size_t)(nreq + 2) * n * sizeof(double));
        if (rc != PM_OK) return rc;
        double *base = (double *)ctx->scratch;
        PM_HIP(ctx, hipMemcpyAsync(base, lon_deg, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipMemcpyAsync(base + n, lat_deg, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dlon = base;
        dlat = base + n;
        int k = 2;
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) p.out[i] = base + (size_t)(k++) * n;
    }
    pm_launch_map(p, dlon, dlat, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    if (mem != PM_MEM_DEVICE) {
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1)
                PM_HIP(ctx, hipMemcpyAsync(out[i], p.out[i], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PM_OK;
}

int pm_transform(pm_ctx *ctx, int from, int to, uint64_t n, const double *a, const double *b, double alt, int flags,
                 double *out_a, double *out_b, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if (from < 0 || from > PM_COORD_LONLAT || to < 0 || to > PM_COORD_LONLAT)
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown coordinate system");
    if (n == 0) return PM_OK;
    if (!a || !b || !out_a || !out_b) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > 0xffffffffull * 256ull) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "too many points");
    // `alt` adjusts the surface for transforms TO lon/lat and is the point's altitude for
    // transforms FROM lon/lat (non-finite alt there gives NaN like the reference)
    double surf_alt = (to == PM_COORD_LONLAT || from == PM_COORD_LONLAT) && std::isfinite(alt) ? alt : 0.0;
    if (to == PM_COORD_LONLAT && !std::isfinite(alt))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "Cannot adjust surface altitude with non-finite alt value");
    pm::Params p;
    fill_params(ctx, surf_alt, p);
    pm::TransformArgs t;
    t.n = n;
    t.alt = alt;
    t.from = from;
    t.to = to;
    t.flags = flags;
    for (int i = 0; i < 3; i++) t.radii0[i] = ctx->geometry.radii[i];
    {
        const pm_geometry &g = ctx->geometry;
        double ks = 1.0 / g.km_per_arcsec, kc = std::cos(g.np_angle_rad), ksn = std::sin(g.np_angle_rad);
        t.Kf[0] = ks * kc; t.Kf[1] = ks * ksn; t.Kf[2] = -ks * ksn; t.Kf[3] = ks * kc;
    }
    if (mem == PM_MEM_DEVICE) {
        t.a = a; t.b = b; t.oa = out_a; t.ob = out_b;
        pm_launch_transform(p, t, ctx->stream);
        PM_HIP(ctx, hipGetLastError());
        return PM_OK;
    }
    rc = ensure_scratch(ctx, (size_t)n * 4 * sizeof(double));
    if (rc != PM_OK) return rc;
    double *base = (double *)ctx->scratch;
    PM_HIP(ctx, hipMemcpyAsync(base, a, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(ctx, hipMemcpyAsync(base + n, b, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    t.a = base; t.b = base + n; t.oa = base + 2 * n; t.ob = base + 3 * n;
    pm_launch_transform(p, t, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    PM_HIP(ctx, hipMemcpyAsync(out_a, t.oa, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(ctx, hipMemcpyAsync(out_b, t.ob, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int pm_xy_map(pm_ctx *ctx, const double *lon_deg, const double *lat_deg, int n0, int n1, double alt, double *x_map,
              double *y_map, int mem)
{
    double *out[PM_NUM_PLANES];
    for (int i = 0; i < PM_NUM_PLANES; i++) out[i] = nullptr;
    out[PM_PIXEL_X] = x_map;
    out[PM_PIXEL_Y] = y_map;
    return pm_backplanes_map(ctx, bit(PM_PIXEL_X) | bit(PM_PIXEL_Y), lon_deg, lat_deg, n0, n1, alt, out, mem);
}

int pm_map_cube(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *x_map, const double *y_map,
                int n0, int n1, int interpolation, int propagate_nan, double *out, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if (!cube || !x_map || !y_map || !out) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "NULL argument");
    size_t esz = dtype_size(dtype);
    if (esz == 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    // BodyXY.map_img body_xy.py:1630: ValueError for unknown interpolation
    int k_rows = 0, k_cols = 0;
    if (interpolation & PM_INTERP_SPLINE_FLAG) {
        k_rows = (interpolation >> 4) & 0xF;
        k_cols = interpolation & 0xF;
        if ((interpolation & ~0x1FF) || k_rows < 1 || k_rows > 5 || k_cols < 1 || k_cols > 5)
            return fail(ctx, PM_ERR_INVALID_ARGUMENT, "Unknown interpolation method %d", interpolation);
        if (k_rows == 1 && k_cols == 1) {
            interpolation = PM_INTERP_LINEAR;
            k_rows = k_cols = 0;
        }
    } else if (interpolation != PM_INTERP_NEAREST && interpolation != PM_INTERP_LINEAR &&
               interpolation != PM_INTERP_SMOOTH) {
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "Unknown interpolation method %d", interpolation);
    }
    const pm_disc &d = ctx->disc;
    if (d.nx <= 0 || d.ny <= 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "image size is empty");
    if (n_planes < 0 || n0 < 0 || n1 < 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "negative shape");
    size_t nmap = (size_t)n0 * n1;
    size_t npx = (size_t)d.nx * d.ny;
    if (n_planes == 0 || nmap == 0) return PM_OK;
    // propagate_nan == 0 almost always needs the statistics pass: complete such calls (and
    // whatever is pending before them) synchronously
    const bool force_sync = (interpolation == PM_INTERP_LINEAR && !propagate_nan);
    if (force_sync && ctx->pending) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }

    pm::ReprojectArgs a;
    a.ny = d.ny;
    a.nx = d.nx;
    a.n_map = (int)nmap;
    a.interpolation = interpolation;
    a.propagate_nan = propagate_nan ? 1 : 0;
    a.plane_stats = nullptr;
    a.seq = 0;
    rc = ensure_flags(ctx, (size_t)n_planes);
    if (rc != PM_OK) return rc;

    if (k_rows) {
        // FITPACK needs more samples than the degree along each axis (scipy raises otherwise)
        if (d.ny <= k_rows || d.nx <= k_cols)
            return fail(ctx, PM_ERR_INVALID_ARGUMENT, "image too small for spline degree (%d, %d)", k_rows, k_cols);
        if (ctx->pending) {
            rc = pm_synchronize(ctx);
            if (rc != PM_OK) return rc;
        }
    }
    // spline_smoothing > 0 (FITPACK smoothing) applies to 'linear' and every spline degree
    const double smoothing = (interpolation == PM_INTERP_NEAREST || interpolation == PM_INTERP_SMOOTH)
                                 ? 0.0 : ctx->spline_smoothing;
    if (smoothing > 0.0) {
        if (!k_rows) k_rows = k_cols = 1;
        if (d.ny <= k_rows || d.nx <= k_cols)
            return fail(ctx, PM_ERR_INVALID_ARGUMENT, "image too small for spline degree (%d, %d)", k_rows, k_cols);
        if (ctx->pending) {
            rc = pm_synchronize(ctx);
            if (rc != PM_OK) return rc;
        }
    }
    const bool smooth = interpolation == PM_INTERP_SMOOTH;
    double limits[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    if (smooth && mem == PM_MEM_DEVICE) {
        // footprint of the map on the image: tiny reduction + 32-byte read-back
        if (!ctx->limits) PM_HIP(ctx, hipMalloc((void **)&ctx->limits, 4 * sizeof(double)));
        pm_launch_map_limits(x_map, y_map, (int)nmap, ctx->limits, ctx->stream);
        PM_HIP(ctx, hipMemcpyAsync(limits, ctx->limits, sizeof(limits), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else if (smooth) {
        for (size_t i = 0; i < nmap; i++) {
            if (!std::isnan(x_map[i])) {
                limits[0] = std::fmin(limits[0], x_map[i]);
                limits[1] = std::fmax(limits[1], x_map[i]);
            }
            if (!std::isnan(y_map[i])) {
                limits[2] = std::fmin(limits[2], y_map[i]);
                limits[3] = std::fmax(limits[3], y_map[i]);
            }
        }
    }
    if (mem == PM_MEM_DEVICE) {
        a.cube = cube;
        a.x_map = x_map;
        a.y_map = y_map;
        a.out = out;
        a.plane_flags = ctx->flags;
        a.n_planes = n_planes;
        if (smooth) return reproject_smooth_resident(ctx, a, dtype, limits);
        if (smoothing > 0.0) return reproject_smoothing_resident(ctx, a, dtype, k_rows, k_cols, smoothing);
        if (k_rows) return reproject_spline_resident(ctx, a, dtype, k_rows, k_cols);
        return reproject_resident(ctx, a, dtype, /*sync_now=*/force_sync);
    }
    // host cube: stream it through the device in chunks of planes
    size_t chunk = (size_t)(1ull << 30) / (npx * esz);
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)n_planes) chunk = (size_t)n_planes;
    size_t cube_bytes = chunk * npx * esz;
    cube_bytes = (cube_bytes + 255) & ~(size_t)255;
    size_t need = cube_bytes + (2 + chunk) * nmap * sizeof(double);
    rc = ensure_scratch(ctx, need);
    if (rc != PM_OK) return rc;
    char *dcube = (char *)ctx->scratch;
    double *dxm = (double *)(dcube + cube_bytes);
    double *dym = dxm + nmap;
    double *dout = dym + nmap;
    PM_HIP(ctx, hipMemcpyAsync(dxm, x_map, nmap * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(ctx, hipMemcpyAsync(dym, y_map, nmap * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    for (size_t p0 = 0; p0 < (size_t)n_planes; p0 += chunk) {
        size_t np = std::min(chunk, (size_t)n_planes - p0);
        PM_HIP(ctx, hipMemcpyAsync(dcube, (const char *)cube + p0 * npx * esz, np * npx * esz, hipMemcpyHostToDevice,
                                   ctx->stream));
        pm::ReprojectArgs b = a;
        b.cube = dcube;
        b.x_map = dxm;
        b.y_map = dym;
        b.out = dout;
        b.plane_flags = ctx->flags;
        b.n_planes = (int)np;
        rc = smooth            ? reproject_smooth_resident(ctx, b, dtype, limits)
             : smoothing > 0.0 ? reproject_smoothing_resident(ctx, b, dtype, k_rows, k_cols, smoothing)
             : k_rows          ? reproject_spline_resident(ctx, b, dtype, k_rows, k_cols)
                      : reproject_resident(ctx, b, dtype, /*sync_now=*/true);
        if (rc != PM_OK) return rc;
        PM_HIP(ctx, hipMemcpyAsync(out + p0 * nmap, dout, np * nmap * sizeof(double), hipMemcpyDeviceToHost,
                                   ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PM_OK;
}

int pm_set_smooth_options(pm_ctx *ctx, int oversample_by, int max_oversampled_img_size)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    ctx->smooth_oversample_by = oversample_by;
    ctx->smooth_max_size = max_oversampled_img_size;
    return PM_OK;
}

int pm_set_spline_smoothing(pm_ctx *ctx, double s)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    if (!(s >= 0.0) || !std::isfinite(s)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "s should be s >= 0.0");
    ctx->spline_smoothing = s;
    return PM_OK;
}

}  // extern "C"
