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
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <atomic>

#include "pm_host.hip.h"

namespace pmh {

int fail(pm_ctx *ctx, int code, const char *fmt, ...)
{
    static std::mutex mu;  // the smoothing-spline workers report from their own threads
    std::lock_guard<std::mutex> lock(mu);
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

// ceil(2^32 / d) for pm::mod_uniform (d >= 2)
static uint32_t mod_magic(uint32_t d) { return d < 2 ? 0u : (uint32_t)(0xFFFFFFFFull / d + 1ull); }

// row / column-block visiting order of the spheroid image kernel for a launch over `rows` rows
void set_row_order(pm::Params &p, int rows)
{
    p.rows = rows;
    p.row_stride = golden_stride(rows);
    p.row_magic = mod_magic((uint32_t)rows);
    p.col_blocks = (uint32_t)((p.nx + pm::kSphBlock - 1) / pm::kSphBlock);
    p.col_magic = mod_magic(p.col_blocks);
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
    {
        // the same map in radians with the RA-like axis negated (body.py:1363: ra = -x), and
        // whether every pixel of the frame stays inside the range of sincos_tiny
        const double k = 3.14159265358979323846 / 180.0 / 3600.0;
        for (int i = 0; i < 3; i++) {
            p.Ar[i] = -(p.A[i] * k);
            p.Ar[3 + i] = p.A[3 + i] * k;
        }
        double amax = 0.0;
        for (int cy = 0; cy < 2; cy++)
            for (int cx = 0; cx < 2; cx++) {
                const double fx = cx ? (double)(d.nx - 1) : 0.0, fy = cy ? (double)(d.ny - 1) : 0.0;
                amax = std::fmax(amax, std::fabs(p.Ar[0] * fx + p.Ar[1] * fy + p.Ar[2]));
                amax = std::fmax(amax, std::fabs(p.Ar[3] * fx + p.Ar[4] * fy + p.Ar[5]));
            }
        p.view_tiny = (amax <= 1e-3) ? 1 : 0;  // NaN compares false
        p.view_direct = (amax <= 1.5) ? 1u : 0u;
        p.lon_k[0] = g.west_positive ? -1.0 : 1.0;
        p.lon_k[1] = p.lon_k[0] * g.wdot;
    }
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
    p.lt_tol = 1e-17 * std::fabs(p.t0);
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
    p.optimize_speed = d.optimize_speed;
    {
        // A pixel whose line of sight passes the body centre by more than the limb's angular radius misses the body
        // whatever the reference's pre-mask (a circle of 1.05 r0 + 1 pixels, body_xy.py:3201-3203) lets through: where
        // that bound is the tighter one it replaces the pre-mask radius - the 5 % annulus (a tenth of a full-disc
        // frame's candidates) never builds a ray. Conservative by construction: rmax, the target's motion over the
        // light-time span of the passes, the offset of the angular origin from the centre direction (1e-16 rad for
        // Body's own matrix, but the block is the caller's), the spherical excess of sqrt(ax^2 + ay^2) over the true
        // separation (cos sep = cos ax cos ay: < sep^2 / 24 relative) and half a pixel on top.
        const double D = std::sqrt(g.T0[0] * g.T0[0] + g.T0[1] * g.T0[1] + g.T0[2] * g.T0[2]);
        const double vt = std::sqrt(g.VT[0] * g.VT[0] + g.VT[1] * g.VT[1] + g.VT[2] * g.VT[2]);
        const double reach = rmax * (1.0 + 1e-9) + 4.0 * vt * rmax / g.clight + 1e-6;
        // |M row 0 x T0| / D = sine of the angle between the angular frame's origin and the centre direction
        const double cx = g.M[1] * g.T0[2] - g.M[2] * g.T0[1], cy = g.M[2] * g.T0[0] - g.M[0] * g.T0[2],
                     cz = g.M[0] * g.T0[1] - g.M[1] * g.T0[0];
        const double off = std::sqrt(cx * cx + cy * cy + cz * cz) / D;
        const double s_rad = std::hypot(p.A[0], p.A[1]) * (3.14159265358979323846 / 648000.0);  // plate scale, rad / pixel
        if (D > 0.0 && reach / D < 0.09 && off < 0.009 && s_rad > 0.0) {
            const double theta = std::asin(reach / D) + 1.001 * off;  // (asin(off) < 1.001 off below 0.009)
            const double rt = theta / s_rad * 1.002 + 0.5;
            if (!d.optimize_speed || rt * rt < p.r2) {
                p.r2 = rt * rt;
                p.optimize_speed = 1;
            }
        }
    }
    rot(g.VT, p.VB, 1.0);
    rot(g.AT, p.AB, 1.0);
    rot(g.S0, p.SB0, 1.0);
    rot(g.VS, p.VSB, 1.0);
    rot(g.AS, p.ASB, 1.0);
    rot(g.VO, p.VOB, 1.0);
    {
        const double vst[3] = {g.VT[0] + g.DVT[0], g.VT[1] + g.DVT[1], g.VT[2] + g.DVT[2]};
        rot(vst, p.VSB_state, 1.0);
        const double ast[3] = {g.AT[0] + g.DAT[0], g.AT[1] + g.DAT[1], g.AT[2] + g.DAT[2]};
        rot(ast, p.ASB_state, 1.0);
        rot(g.WP, p.WPB, 1.0);
    }
    rot(g.ring_n, p.ring_nb, 1.0);
    rot(g.sub_obsvec, p.sub_obs_b, 1.0);
    rot(g.sub_ray, p.sub_ray_b, 1.0);
    p.ira = 1.0 / p.radii[0];
    p.irc = 1.0 / p.radii[2];
    p.inv_c = 1.0 / g.clight;
    p.lat_k = (p.radii[0] / p.radii[2]) * (p.radii[0] / p.radii[2]);
    p.a_over_c = p.radii[0] / p.radii[2];
    for (int i = 0; i < 3; i++) {
        p.O0s[i] = p.O0[i] / p.radii[i];
        p.VBs[i] = p.VB[i] / p.radii[i];
    }
    {
        const double sb = std::sqrt(p.SB0[0] * p.SB0[0] + p.SB0[1] * p.SB0[1] + p.SB0[2] * p.SB0[2]);
        p.sun_ds0 = (p.t0 - g.ts0) - sb / g.clight;
        p.sun_k = 1.0 / (sb * g.clight);
    }
    {
        // phase-angle series of the fast path (Params::ph): expansion point = the body centre
        const double so = std::sqrt(p.O0[0] * p.O0[0] + p.O0[1] * p.O0[1] + p.O0[2] * p.O0[2]);
        const double ss = std::sqrt(p.SB0[0] * p.SB0[0] + p.SB0[1] * p.SB0[1] + p.SB0[2] * p.SB0[2]);
        const double c0 = (p.O0[0] * p.SB0[0] + p.O0[1] * p.SB0[1] + p.O0[2] * p.SB0[2]) / (so * ss);
        const double s2 = 1.0 - c0 * c0, s1 = std::sqrt(std::fmax(s2, 0.0));
        const double rmax_ = std::fmax(p.radii[0], std::fmax(p.radii[1], p.radii[2]));
        // largest excursion of the phase angle over the disc (point anywhere within rmax of the centre, the
        // target's motion over the light-time span included) against sin(g0): the ratio of the series
        const double dg = 1.25 * rmax_ * (1.0 / so + 1.0 / ss) + 1e-7;
        p.phase_series = (s1 > 1e-3 && dg / s1 <= 1.5e-3) ? 1 : 0;
        p.ph[0] = c0;
        p.ph[1] = std::acos(std::fmin(1.0, std::fmax(-1.0, c0)));
        // d/dc acos = -(1 - c^2)^(-1/2); further derivatives in closed form, divided by n!
        const double i1 = s1 > 0.0 ? 1.0 / s1 : 0.0, i2 = i1 * i1;
        p.ph[2] = -i1;
        p.ph[3] = -0.5 * c0 * i1 * i2;
        p.ph[4] = -(1.0 + 2.0 * c0 * c0) * i1 * i2 * i2 / 6.0;
        p.ph[5] = -c0 * (9.0 + 6.0 * c0 * c0) * i1 * i2 * i2 * i2 / 24.0;
    }
    {
        // one quantum of the epoch et - lt, as an angle on the body: beyond 1e-9 deg the fast path keeps to
        // the reference's own sequence of epochs (Params::plain_lt)
        const double quantum = std::nextafter(std::fabs(p.t0), INFINITY) - std::fabs(p.t0);
        const double vt = std::sqrt(g.VT[0] * g.VT[0] + g.VT[1] * g.VT[1] + g.VT[2] * g.VT[2]);
        const double rmin_ = std::fmin(p.radii[0], std::fmin(p.radii[1], p.radii[2]));
        // (... or as a turn of the body: a fast rotator's longitudes move by wdot x quantum - round 4, found by the
        //  fast-spin test of the general kernel: 9e-9 deg per quantum at 30 times Jupiter's spin)
        p.turn_quantum = std::fabs(g.wdot) * quantum > 1.7453292519943296e-11 ? 1 : 0;
        p.plain_lt = (vt * quantum > 1.7453292519943296e-11 * rmin_ || p.turn_quantum) ? 1 : 0;
        p.cf_iter = (p.plain_lt && ctx->lt_mode == 0) ? 1 : 0;
        if (ctx->lt_mode == 1 || (ctx->lt_mode == 2 && !p.plain_lt)) p.plain_lt = ctx->lt_mode;
    }
    {
        // closed-form light time (Params::Y00 ...): in long double, rounded once
        p.lt_c_eff = g.et - p.t0;
        long double v2 = 0.0L;
        for (int i = 0; i < 3; i++) {
            const long double v = (long double)p.VB[i] / (long double)p.radii[i];
            // (O0 = -R0 T0 carries the rounding of its own dot products, 1e-16 relative: taken again here)
            const long double o0 = -((long double)g.R0[3 * i] * g.T0[0] + (long double)g.R0[3 * i + 1] * g.T0[1] +
                                     (long double)g.R0[3 * i + 2] * g.T0[2]);
            const long double vb = (long double)g.R0[3 * i] * g.VT[0] + (long double)g.R0[3 * i + 1] * g.VT[1] +
                                   (long double)g.R0[3 * i + 2] * g.VT[2];
            const long double y = o0 / (long double)p.radii[i] - vb / (long double)p.radii[i] * (long double)p.lt_c_eff;
            p.Y00[i] = (double)y;
            p.Y00lo[i] = (double)(y - (long double)p.Y00[i]);
            p.Wc[i] = (double)(v / (long double)g.clight);
            v2 += v * v;
        }
        // |d| <= R / c between the first pass and the fixed point (observer outside the body); P.P moves by
        // 2 |P| |VBs| |d| + (|VBs| d)^2 over that, |P| <= 1 in the band that matters
        const double rmax_ = std::fmax(p.radii[0], std::fmax(p.radii[1], p.radii[2]));
        const double dv = std::sqrt((double)v2) * 1.05 * rmax_ / g.clight;
        const double band = 1.5 * (2.0 * dv + dv * dv) + 1e-10;
        p.p2_lo = 1.0 - band;
        p.p2_hi = 1.0 + band;
        // ... and for a body that is not a spheroid the shape itself turns under the ray by wdot |d|, which moves its
        // limb by that angle times (a^2 - b^2) / b^2 of a radius; the target's acceleration adds A d^2 / 2 (general kernel)
        const double ab = p.radii[0] / p.radii[1], ba = p.radii[1] / p.radii[0];
        p.tri_k = g.wdot * (ba - ab);
        const double dmax = 1.05 * rmax_ / g.clight;
        const double rmin_ = std::fmin(p.radii[0], std::fmin(p.radii[1], p.radii[2]));
        double a2 = 0.0;
        for (int i = 0; i < 3; i++) a2 += g.AT[i] * g.AT[i];
        const double turn = std::fabs(g.wdot) * dmax * std::fabs(ab * ab - 1.0) + 0.5 * std::sqrt(a2) * dmax * dmax / rmin_;
        p.p2_lo_rot = 1.0 - (band + 4.0 * turn);
        // the closed form for a triaxial body: its shape frozen over the light-time span, the turn put back to first order -
        // while the second-order term (wdot d)^2 r |a^2 / b^2 - 1| stays below 1e-10 km (Io: 1e-12; a Jupiter-sized test
        // body with b = 0.97 a: 4e-6, which keeps the sequence + Newton step)
        {
            const double ang = std::fabs(g.wdot) * dmax;
            p.tri_cf = (p.radii[0] != p.radii[1] && ang * ang * rmax_ * std::fabs(ab * ab - 1.0) < 1e-10 && ctx->lt_mode == 0) ? 1 : 0;
        }
    }
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
    set_row_order(p, d.ny);
    p.n0 = p.n1 = 0;
    p.mask = 0;
    for (int i = 0; i < PM_NUM_PLANES; i++) p.out[i] = nullptr;
}

}  // namespace pmh

using namespace pmh;

namespace {

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

// error return of a host-buffer call that may have copies in flight: drain them first (pm_hostpipe.hip)
int host_fail(pm_ctx *ctx, int rc)
{
    pipe_abort(ctx);
    return rc;
}

// `mem` of an entry point: PM_MEM_HOST and PM_MEM_DEVICE everywhere, PM_MEM_HOST_CUBE only where a
// cube is the one host buffer of the call (pm_map_cube, pm_mapped_data, pm_map_cube_sharded)
int check_mem(pm_ctx *ctx, int mem, bool host_cube_ok)
{
    if (mem == PM_MEM_HOST || mem == PM_MEM_DEVICE || (host_cube_ok && mem == PM_MEM_HOST_CUBE)) return PM_OK;
    if (mem == PM_MEM_HOST_CUBE) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_MEM_HOST_CUBE applies to cube mapping calls only");
    return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown mem value %d", mem);
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
    // defaults from the environment for the A/B tools - only behind PM_DEBUG_ENV=1 (pm_debug_env)
    const char *fg = pm_debug_env("PM_FORCE_GENERAL");
    ctx->force_general = fg && fg[0] == '1';
    if (const char *fp = pm_debug_env("PM_FUSE_PLANES")) ctx->fuse_planes = fp[0] != '0';
    if (const char *m = pm_debug_env("PM_LT_MODE")) ctx->lt_mode = (m[0] == '1') ? 1 : (m[0] == '2') ? 2 : 0;
    if (pm_debug_env("PM_HOSTPIPE_TRACE")) ctx->trace |= 1;
    if (pm_debug_env("PM_SM_DEBUG")) ctx->trace |= 2;
    if (const char *w = pm_debug_env("PM_SM_BATCH_PLANES")) ctx->sm_batch_planes = std::max(0, std::min(std::atoi(w), 4096));
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
    if (ctx->sm_status_host) (void)hipHostFree(ctx->sm_status_host);
    if (ctx->spline_ev) (void)hipEventDestroy(ctx->spline_ev);
    for (auto &ac : ctx->axis) {
        if (ac.t) (void)hipFree(ac.t);
        if (ac.lu) (void)hipFree(ac.lu);
    }
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->hist) (void)hipFree(ctx->hist);
    pipe_destroy(ctx);
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

int pm_set_option(pm_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    switch (option) {
    case PM_OPT_GENERAL_KERNEL:
        ctx->force_general = value != 0;
        return PM_OK;
    case PM_OPT_HOST_CHUNK_BYTES:
        ctx->host_chunk_bytes = (size_t)std::min<int64_t>(std::max<int64_t>(value, (int64_t)1 << 20), (int64_t)1 << 30);
        return PM_OK;
    case PM_OPT_HOST_COPY_THREADS:
        if (value < 0 || value > 64) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "copy threads must be in 0..64");
        ctx->host_copy_threads = (int)value;
        return PM_OK;
    case PM_OPT_ZERO_COPY:
        if (value < -1 || value > 4) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_HOST_CUBE_ROUTE takes -1 .. 4");
        ctx->zero_copy = (int)value;
        return PM_OK;
    case PM_OPT_SPARSE_FRAME:
        if (value < -1 || value > 1) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_SPARSE_FRAME takes -1, 0 or 1");
        ctx->sparse_frame = (int)value;
        return PM_OK;
    case PM_OPT_BLOCK_TABLE_CACHE:
        ctx->table_cache = value != 0;
        return PM_OK;
    case PM_OPT_FUSE_PLANES:
        ctx->fuse_planes = value != 0;
        return PM_OK;
    case PM_OPT_LT_MODE:
        if (value < 0 || value > 2) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_LT_MODE takes 0, 1 or 2");
        ctx->lt_mode = (int)value;
        return PM_OK;
    case PM_OPT_TRACE:
        if (value < 0 || value > 7) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_TRACE takes a mask of 1 (host path), 2 (smoothing splines) and 4 (device-side stage sums)");
        ctx->trace = (int)value;
        return PM_OK;
    case PM_OPT_SM_BATCH_PLANES:
        if (value < 0 || value > 4096) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_SM_BATCH_PLANES takes 0 (the library's choice) .. 4096");
        ctx->sm_batch_planes = (int)value;
        return PM_OK;
    case PM_OPT_SPLINE_SEGMENT:
        if (value < -1 || value > (1 << 20) || (value > 0 && value < 64))
            return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_SPLINE_SEGMENT takes 0 (the library's choice), -1 (never) or a segment length >= 64");
        ctx->spline_segment = (int)value;
        return PM_OK;
    case PM_OPT_FETCH_BLOCK_BYTES:
        if (value != 64 && value != 128 && value != 256) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "PM_OPT_FETCH_BLOCK_BYTES takes 64, 128 or 256");
        ctx->fetch_shift = value == 64 ? 6 : value == 128 ? 7 : 8;
        pipe_reset_route_stats(ctx);
        return PM_OK;
    case PM_OPT_ROUTE_EXPLORE:
        ctx->route_explore = value != 0;
        pipe_reset_route_stats(ctx);  // (what was measured is forgotten: the next large call measures again)
        return PM_OK;
    case PM_OPT_BLOCK_TABLE_HITS:
    case PM_OPT_LAST_CUBE_ROUTE:
    case PM_OPT_LAST_REDO_PLANES:
    case PM_OPT_LAST_DISC_KERNEL:
    case PM_OPT_LAST_LT_PATH:
    case PM_OPT_HOST_COPY_THREADS_IN_USE:
    case PM_OPT_HYBRID_FETCH_PERMILLE:
    case PM_OPT_LAST_SPLINE_SEGMENT:
    case PM_OPT_LAST_SM_KNIFE_EDGES:
    case PM_OPT_LAST_SM_ILL_CONDITIONED:
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "option %d is read-only", option);
    }
    return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown option %d", option);
}

int pm_set_chunk_callback(pm_ctx *ctx, pm_chunk_callback cb, void *user)
{
    if (!ctx) return PM_ERR_INVALID_ARGUMENT;
    ctx->chunk_cb = cb;
    ctx->chunk_user = cb ? user : nullptr;
    return PM_OK;
}

int pm_get_option(pm_ctx *ctx, int option, int64_t *value)
{
    if (!ctx || !value) return PM_ERR_INVALID_ARGUMENT;
    switch (option) {
    case PM_OPT_GENERAL_KERNEL: *value = ctx->force_general ? 1 : 0; return PM_OK;
    case PM_OPT_HOST_CHUNK_BYTES: *value = (int64_t)ctx->host_chunk_bytes; return PM_OK;
    case PM_OPT_HOST_COPY_THREADS: *value = ctx->host_copy_threads; return PM_OK;
    case PM_OPT_ZERO_COPY: *value = ctx->zero_copy; return PM_OK;
    case PM_OPT_SPARSE_FRAME: *value = ctx->sparse_frame; return PM_OK;
    case PM_OPT_LAST_DISC_KERNEL: *value = ctx->last_disc_kernel; return PM_OK;
    case PM_OPT_LAST_LT_PATH: *value = ctx->last_lt_path; return PM_OK;
    case PM_OPT_BLOCK_TABLE_CACHE: *value = ctx->table_cache; return PM_OK;
    case PM_OPT_FUSE_PLANES: *value = ctx->fuse_planes; return PM_OK;
    case PM_OPT_BLOCK_TABLE_HITS: *value = pipe_table_hits(ctx); return PM_OK;
    case PM_OPT_ROUTE_EXPLORE: *value = ctx->route_explore; return PM_OK;
    case PM_OPT_FETCH_BLOCK_BYTES: *value = 1 << ctx->fetch_shift; return PM_OK;
    case PM_OPT_LT_MODE: *value = ctx->lt_mode; return PM_OK;
    case PM_OPT_TRACE: *value = ctx->trace; return PM_OK;
    case PM_OPT_SM_BATCH_PLANES: *value = ctx->sm_batch_planes; return PM_OK;
    case PM_OPT_SPLINE_SEGMENT: *value = ctx->spline_segment; return PM_OK;
    case PM_OPT_LAST_SPLINE_SEGMENT: *value = ctx->last_spline_segment; return PM_OK;
    case PM_OPT_LAST_SM_KNIFE_EDGES: *value = ctx->last_sm_knife_edges; return PM_OK;
    case PM_OPT_LAST_SM_ILL_CONDITIONED: *value = ctx->last_sm_ill_conditioned; return PM_OK;
    case PM_OPT_LAST_CUBE_ROUTE: *value = ctx->last_cube_route; return PM_OK;
    case PM_OPT_LAST_REDO_PLANES: *value = ctx->last_redo_planes; return PM_OK;
    case PM_OPT_HOST_COPY_THREADS_IN_USE: *value = pipe_copy_threads(ctx); return PM_OK;
    case PM_OPT_HYBRID_FETCH_PERMILLE: *value = pipe_hybrid_fetch_permille(ctx); return PM_OK;
    }
    if (option >= PM_OPT_LAST_STAGE_NS && option < PM_OPT_LAST_STAGE_NS + 16) {
        *value = (int64_t)ctx->last_stage_ns[option - PM_OPT_LAST_STAGE_NS];
        return PM_OK;
    }
    if (option >= PM_OPT_ROUTE_NS_PER_PLANE && option < PM_OPT_ROUTE_NS_PER_PLANE + 5) {
        *value = pipe_route_ns_per_plane(ctx, option - PM_OPT_ROUTE_NS_PER_PLANE);
        return PM_OK;
    }
    return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown option %d", option);
}

int pm_host_alloc(pm_ctx *ctx, uint64_t bytes, void **hptr)
{
    if (!ctx || !hptr) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    // non-coherent (coarse-grained) pinned memory: the GPU may cache it, CPU and GPU views meet at
    // kernel / copy boundaries - which is how every entry point here uses host buffers
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocNonCoherent | hipHostMallocPortable);
    if (e != hipSuccess)
        return fail(ctx, PM_ERR_ALLOC, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return PM_OK;
}

int pm_host_free(pm_ctx *ctx, void *hptr)
{
    if (!hptr) return PM_OK;
    // (ctx == NULL: the context that allocated the block has been destroyed before the block's owner let go of it -
    //  no stream of ours can still be working on it)
    if (!ctx) return hipHostFree(hptr) == hipSuccess ? PM_OK : PM_ERR_HIP;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PM_HIP(ctx, hipHostFree(hptr));
    return PM_OK;
}

int pm_host_register(pm_ctx *ctx, void *hptr, uint64_t bytes)
{
    if (!ctx || !hptr) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipHostRegister(hptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return PM_OK;
}

int pm_host_unregister(pm_ctx *ctx, void *hptr)
{
    if (!ctx || !hptr) return PM_ERR_INVALID_ARGUMENT;
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PM_HIP(ctx, hipHostUnregister(hptr));
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

// ------------------------------------------------------------------ DLPack export of device memory
namespace {
// dlpack.h (legacy DLManagedTensor), restated: the layout a 'dltensor' capsule carries
struct DLDevice { int32_t device_type, device_id; };
struct DLDataType { uint8_t code, bits; uint16_t lanes; };
struct DLTensor { void *data; DLDevice device; int32_t ndim; DLDataType dtype; int64_t *shape; int64_t *strides; uint64_t byte_offset; };
struct DLManagedTensor { DLTensor dl_tensor; void *manager_ctx; void (*deleter)(DLManagedTensor *); };
constexpr int kDLROCM = 10;
struct Export { DLManagedTensor m; int64_t shape[8]; };
}  // namespace
struct pm_dlpack_hold {
    std::atomic<int64_t> exports{0};
    std::atomic<int> orphaned{0};  // 1: the owner is gone, memory stays; 2: the owner is gone and the last export frees the memory
    void *dptr = nullptr;
    int device = 0;
};
namespace {
void hold_finish(pm_dlpack_hold *h, int mode)
{
    if (mode == 2 && h->dptr) {
        (void)hipSetDevice(h->device);
        (void)hipFree(h->dptr);
    }
    delete h;
}
void export_deleter(DLManagedTensor *m)
{
    if (!m) return;
    pm_dlpack_hold *h = (pm_dlpack_hold *)m->manager_ctx;
    delete (Export *)m;  // (m is the first member of its Export)
    if (h->exports.fetch_sub(1) == 1) {
        const int mode = h->orphaned.load();
        if (mode) hold_finish(h, mode);
    }
}
}  // namespace

pm_dlpack_hold *pm_dlpack_hold_create(void *dptr, int device)
{
    pm_dlpack_hold *h = new (std::nothrow) pm_dlpack_hold();
    if (h) {
        h->dptr = dptr;
        h->device = device;
    }
    return h;
}

void *pm_dlpack_export(pm_dlpack_hold *hold, int dtype_code, int bits, int ndim, const int64_t *shape)
{
    if (!hold || ndim < 0 || ndim > 8 || (ndim && !shape) || hold->orphaned.load()) return nullptr;
    Export *e = new (std::nothrow) Export();
    if (!e) return nullptr;
    for (int i = 0; i < ndim; i++) e->shape[i] = shape[i];
    e->m.dl_tensor.data = hold->dptr;
    e->m.dl_tensor.device = {kDLROCM, hold->device};
    e->m.dl_tensor.ndim = ndim;
    e->m.dl_tensor.dtype = {(uint8_t)dtype_code, (uint8_t)bits, 1};
    e->m.dl_tensor.shape = e->shape;
    e->m.dl_tensor.strides = nullptr;  // C-contiguous
    e->m.dl_tensor.byte_offset = 0;
    e->m.manager_ctx = hold;
    e->m.deleter = export_deleter;
    hold->exports.fetch_add(1);
    return &e->m;
}

int64_t pm_dlpack_exports(const pm_dlpack_hold *hold) { return hold ? hold->exports.load() : 0; }

int pm_dlpack_release(pm_dlpack_hold *hold, int free_memory)
{
    if (!hold) return 1;
    // (an export that ends between the two steps sees `orphaned` set and finishes the hold itself: the exchange
    //  below makes exactly one side do it)
    hold->exports.fetch_add(1);  // the owner's own guard
    hold->orphaned.store(free_memory ? 2 : 1);
    if (hold->exports.fetch_sub(1) == 1) {
        hold_finish(hold, free_memory ? 2 : 1);
        return 1;
    }
    return 0;
}

void pm_dlpack_delete(void *managed_tensor)
{
    DLManagedTensor *m = (DLManagedTensor *)managed_tensor;
    if (m && m->deleter) m->deleter(m);
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
    // the image kernels address a row with a 32-bit byte offset (rows go on gridDim.y: frames taller than a launch's 65 535
    // rows are mapped in row blocks, pm_backplanes_img_rows)
    if (disc->nx > (1 << 28) || disc->ny > (1 << 30))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "image size is limited to nx <= 2^28, ny <= 2^30");
    ctx->disc = *disc;
    ctx->have_disc = true;
    return PM_OK;
}

// Whether the B0 fast paths apply (k_disc_sph BODY 0 / 1, k_radec_query_b0): an observer well outside the body
// (|O0| > 2 radii in scaled coordinates; anything else, e.g. a lander, goes through the general kernels), the target's /
// Sun's acceleration negligible over the light-time span of a disc intercept (|d| <= R / c: fine while A (R/c)^2 / 2 is
// below 1e-12 of the smallest radius - Jupiter: 6e-9 km of 66 854 km; the ray itself is rounded at 1e-7 km), a spin small
// enough for the short series a triaxial body is turned with per light-time evaluation, and no PM_OPT_GENERAL_KERNEL.
static bool fast_path_geometry(const pm_ctx *ctx, const pm::Params &pd)
{
    double y2 = 0.0;
    for (int i = 0; i < 3; i++) y2 += (pd.O0[i] / pd.radii[i]) * (pd.O0[i] / pd.radii[i]);
    const double rmax = std::fmax(pd.radii[0], std::fmax(pd.radii[1], pd.radii[2]));
    const double rmin = std::fmin(pd.radii[0], std::fmin(pd.radii[1], pd.radii[2]));
    const double span = rmax / ctx->geometry.clight;
    double acc2 = 0.0, accs2 = 0.0;
    for (int i = 0; i < 3; i++) {
        acc2 += ctx->geometry.AT[i] * ctx->geometry.AT[i];
        accs2 += ctx->geometry.AS[i] * ctx->geometry.AS[i];
    }
    const bool slow = 0.5 * std::sqrt(std::fmax(acc2, accs2)) * span * span < 1e-12 * rmin;
    const bool small_spin = std::fabs(ctx->geometry.wdot) * span < 1e-3;
    return y2 > 4.0 && slow && small_spin && !ctx->force_general;
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
    if ((rc = check_mem(ctx, mem, false)) != PM_OK) return rc;
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
    set_row_order(p, n_rows);
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

    // gridDim.y holds at most 65 535 rows: a taller frame (body_xy.py:3166 knows no limit) goes in blocks of rows, each a
    // launch of its own with its own row order - a pixel's value depends on its coordinates only
    constexpr int kRowsPerLaunch = 32768;
    const pm::Params p_all = p;
    for (int b0 = 0; b0 < n_rows; b0 += kRowsPerLaunch) {
    const int nb = std::min(kRowsPerLaunch, n_rows - b0);
    p = p_all;
    p.y_off = row_begin + b0;
    set_row_order(p, nb);
    for (int i = 0; i < PM_NUM_PLANES; i++)
        if ((plane_mask >> i) & 1) p.out[i] = p_all.out[i] + (size_t)b0 * d.nx;
    bool fused_sky = false;
    if (plane_mask & kDiscBits) {
        int flags = 0;
        if (plane_mask & kIllumBits) flags |= 1;
        if (plane_mask & kStateBits) flags |= 2;
        if (plane_mask & kRingBits) flags |= 4;
        pm::Params pd = p;
        pd.mask = plane_mask & kDiscBits;
        // spheroids (every planet in pck00010) take the rotation-free fast path, triaxial bodies
        // (most moons) its variant with one small rotation per light-time evaluation
        const bool spheroid = fast_path_geometry(ctx, pd);
        // every plane of the frame from one launch (PM_OPT_FUSE_PLANES): the sky / limb planes ride along
        // (no fused variant where an epoch quantum is visible: those geometries take the QUANT kernels, two launches)
        fused_sky = spheroid && ctx->fuse_planes && (plane_mask & kSkyBits) && !pd.cf_iter && !pd.turn_quantum;
        if (fused_sky) {
            pd.mask = plane_mask;
            flags |= (plane_mask & kLimbBits) ? (2 << 3) : (1 << 3);
        }
        if (spheroid)
            pm_launch_disc_spheroid(pd, flags, ctx->stream);
        else
            pm_launch_disc(pd, flags, ctx->stream);
        ctx->last_disc_kernel = spheroid ? (pd.radii[0] != pd.radii[1] ? 2 : 1) : 3;
        {
            const bool tri_body = pd.radii[0] != pd.radii[1];
            const bool closed_form = spheroid && (pd.plain_lt == 0 || pd.cf_iter) && (!tri_body || pd.tri_cf);
            ctx->last_lt_path = (closed_form ? 1 : 0) | (closed_form && pd.cf_iter ? 2 : 0) | (pd.plain_lt ? 4 : 0) | (pd.turn_quantum ? 8 : 0);
        }
    }
    if ((plane_mask & kSkyBits) && !fused_sky) {
        pm::Params ps = p;
        ps.mask = plane_mask & kSkyBits;
        pm_launch_sky(ps, (plane_mask & kLimbBits) != 0, ctx->stream);
    }
    }  // (row blocks)
    p = p_all;
    PM_HIP(ctx, hipGetLastError());

    if (mem != PM_MEM_DEVICE) {
        // results -> caller memory at the rate of the link (pm_hostpipe.hip)
        // Planes of the disc (not the ring planes, which every pixel has) are NaN outside the radius
        // pre-mask: when that circle leaves a good part of the frame empty only bands around it are
        // copied and the copy threads write the NaN (PM_OPT_SPARSE_FRAME).
        const uint64_t nan_outside = p.optimize_speed ? (kDiscBits & ~kRingBits) : 0;  // (the pre-mask, or the limb bound of fill_params)
        const double circle = 3.14159265358979323846 * p.r2 / ((double)d.nx * (double)n_rows);
        const bool sparse = ctx->sparse_frame != 0 && (size_t)d.nx * sizeof(double) <= ((size_t)1 << 20) &&
                            (ctx->sparse_frame > 0 || (npx * sizeof(double) >= ((size_t)64 << 20) && circle < 0.85));
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) {
                if (sparse && ((nan_outside >> i) & 1))
                    rc = d2h_issue_disc(ctx, ctx->stream, out[i], p.out[i], (size_t)d.nx, (size_t)n_rows, (double)row_begin, p.x0, p.y0, p.r2);
                else
                    rc = d2h_issue(ctx, ctx->stream, out[i], p.out[i], npx * sizeof(double));
                if (rc != PM_OK) return host_fail(ctx, rc);
            }
        return d2h_finish(ctx, ctx->stream);
    }
    return PM_OK;
}

int pm_backplanes_map(pm_ctx *ctx, uint64_t plane_mask, const double *lon_deg, const double *lat_deg, int n0, int n1,
                      double alt, double *const *out, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if ((rc = check_mem(ctx, mem, false)) != PM_OK) return rc;
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
    // the x/y map alone (what a reprojection asks for) has its own short kernel; PM_OPT_GENERAL_KERNEL
    // keeps it on the general one, like the image planes
    if ((plane_mask & ~(bit(PM_PIXEL_X) | bit(PM_PIXEL_Y))) == 0 && !ctx->force_general)
        pm_launch_map_xy(p, dlon, dlat, ctx->stream);
    else
        pm_launch_map(p, dlon, dlat, ctx->force_general != 0, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    if (mem != PM_MEM_DEVICE) {
        for (int i = 0; i < PM_NUM_PLANES; i++)
            if ((plane_mask >> i) & 1) {
                rc = d2h_issue(ctx, ctx->stream, out[i], p.out[i], n * sizeof(double));
                if (rc != PM_OK) return host_fail(ctx, rc);
            }
        return d2h_finish(ctx, ctx->stream);
    }
    return PM_OK;
}

int pm_transform(pm_ctx *ctx, int from, int to, uint64_t n, const double *a, const double *b, double alt, int flags,
                 double *out_a, double *out_b, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if ((rc = check_mem(ctx, mem, false)) != PM_OK) return rc;
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
    rc = d2h_issue(ctx, ctx->stream, out_a, t.oa, n * sizeof(double));
    if (rc == PM_OK) rc = d2h_issue(ctx, ctx->stream, out_b, t.ob, n * sizeof(double));
    if (rc != PM_OK) return host_fail(ctx, rc);
    return d2h_finish(ctx, ctx->stream);
}

int pm_radec_query(pm_ctx *ctx, uint64_t n, const double *ra_deg, const double *dec_deg, double alt,
                   int ring_only_visible, double *out, int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if ((rc = check_mem(ctx, mem, false)) != PM_OK) return rc;
    if (n == 0) return PM_OK;
    if (!ra_deg || !dec_deg || !out) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > 0xffffffffull * 256ull) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "too many points");
    if (!std::isfinite(alt))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT, "Cannot adjust surface altitude with non-finite alt value");
    pm::Params p;
    fill_params(ctx, alt, p);
    // a spheroid seen from outside: the B0 evaluation (k_radec_query_b0); any other body, PM_OPT_GENERAL_KERNEL: J2000
    const bool b0 = p.radii[0] == p.radii[1] && fast_path_geometry(ctx, p);
    if (mem == PM_MEM_DEVICE) {
        pm_launch_radec_query(p, ra_deg, dec_deg, n, ring_only_visible, out, b0, ctx->stream);
        PM_HIP(ctx, hipGetLastError());
        return PM_OK;
    }
    rc = ensure_scratch(ctx, (size_t)n * 10 * sizeof(double));
    if (rc != PM_OK) return rc;
    double *base = (double *)ctx->scratch;
    PM_HIP(ctx, hipMemcpyAsync(base, ra_deg, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PM_HIP(ctx, hipMemcpyAsync(base + n, dec_deg, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    pm_launch_radec_query(p, base, base + n, n, ring_only_visible, base + 2 * n, b0, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    rc = d2h_issue(ctx, ctx->stream, out, base + 2 * n, n * 8 * sizeof(double));
    if (rc != PM_OK) return host_fail(ctx, rc);
    return d2h_finish(ctx, ctx->stream);
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
    if ((rc = check_mem(ctx, mem, true)) != PM_OK) return rc;
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
    // (PM_OPT_LAST_REDO_PLANES describes THIS call once it has finished: the spline / smoothing / 'smooth'
    //  paths never redo planes and must not leave an earlier call's count behind)
    ctx->last_redo_planes = 0;
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
    // a pending asynchronous call still points at the flag array: finish it (flag check and
    // nanmedian replay) before the array is regrown, which frees and zeroes it
    if (ctx->pending && (size_t)n_planes > ctx->flags_count) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }
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
    if (mem == PM_MEM_HOST_CUBE && smooth)
        return fail(ctx, PM_ERR_UNSUPPORTED, "PM_MEM_HOST_CUBE supports nearest / linear with NaN propagation only");
    // 'smooth': the footprint of the map on the image - a small reduction over the x / y maps where they are (or arrive) on
    // the device + a 32-byte read-back (on the host it was 13 ms of one core for the 6.5 M cells of a 0.1 deg map)
    double limits[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    auto map_limits = [&](const double *dx, const double *dy) -> int {
        if (!ctx->limits) PM_HIP(ctx, hipMalloc((void **)&ctx->limits, 4 * (1 + pm::kMapLimitsBlocks) * sizeof(double)));
        pm_launch_map_limits(dx, dy, (int)nmap, ctx->limits, ctx->stream);
        PM_HIP(ctx, hipMemcpyAsync(limits, ctx->limits, sizeof(limits), hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PM_OK;
    };
    if (smooth && mem == PM_MEM_DEVICE && (rc = map_limits(x_map, y_map)) != PM_OK) return rc;
    if (mem == PM_MEM_HOST_CUBE) {
        // the cube in host memory, maps and result in HBM (the per-rank step of a sharded cube: the
        // mapped planes feed an RCCL all-gather)
        if (smooth || k_rows || smoothing > 0.0 || force_sync)
            return fail(ctx, PM_ERR_UNSUPPORTED, "PM_MEM_HOST_CUBE supports nearest / linear with NaN propagation only");
        const size_t cube_bytes = (size_t)n_planes * npx * esz;
        if (ctx->zero_copy == 1 && host_is_pinned(cube, cube_bytes)) {
            // pinned, in-place gather asked for: enqueue like a device call
            const void *dcube = nullptr;
            PM_HIP(ctx, hipHostGetDevicePointer((void **)&dcube, const_cast<void *>(cube), 0));
            a.cube = dcube;
            a.x_map = x_map;
            a.y_map = y_map;
            a.out = out;
            a.plane_flags = ctx->flags;
            a.n_planes = n_planes;
            return reproject_resident(ctx, a, dtype, /*sync_now=*/false);
        }
        return map_cube_host_pipelined(ctx, cube, dtype, n_planes, x_map, y_map, nmap, a, out, true);
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
    // host cube, nearest / linear with NaN propagation (the default of get_mapped_data): the
    // pipelined / zero-copy path of pm_hostpipe.hip
    if (!smooth && !k_rows && !(smoothing > 0.0) && !force_sync)
        return map_cube_host_pipelined(ctx, cube, dtype, n_planes, x_map, y_map, nmap, a, out, false);
    // the other modes: chunks of planes through the device, one after the other (their fits and
    // statistics passes are synchronous anyway)
    if (ctx->pending) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }
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
    if (smooth && (rc = map_limits(dxm, dym)) != PM_OK) return rc;
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
        rc = d2h_issue(ctx, ctx->stream, out + p0 * nmap, dout, np * nmap * sizeof(double));
        if (rc == PM_OK) rc = d2h_finish(ctx, ctx->stream);
        if (rc != PM_OK) return host_fail(ctx, rc);
    }
    return PM_OK;
}

int pm_mapped_data(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *lon_deg, const double *lat_deg,
                   int n0, int n1, double alt, int interpolation, int propagate_nan, double *x_map, double *y_map, double *out,
                   int mem)
{
    int rc = check_ready(ctx, true);
    if (rc != PM_OK) return rc;
    if ((rc = check_mem(ctx, mem, true)) != PM_OK) return rc;
    if (!cube || !lon_deg || !lat_deg || !x_map || !y_map || !out) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!std::isfinite(alt)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "alt must be finite");
    const size_t esz = dtype_size(dtype);
    if (esz == 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "unknown dtype %d", dtype);
    if (n_planes < 0 || n0 < 0 || n1 < 0) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "negative shape");
    const pm_disc &d = ctx->disc;
    // one launch: device buffers, a handful of planes, nearest / linear, image at least 2 x 2 ... anything
    // else takes the two calls it stands for
    const bool simple = interpolation == PM_INTERP_NEAREST || (interpolation == PM_INTERP_LINEAR && !(ctx->spline_smoothing > 0.0));
    const bool fused = mem == PM_MEM_DEVICE && simple && n_planes >= 1 && n_planes <= 8 && d.nx >= 1 && d.ny >= 1 &&
                       (size_t)n0 * n1 > 0 && !ctx->force_general;
    if (!fused) {
        // (PM_MEM_HOST_CUBE: only the cube is a host buffer - the grids and the maps are device pointers)
        rc = pm_xy_map(ctx, lon_deg, lat_deg, n0, n1, alt, x_map, y_map, mem == PM_MEM_HOST_CUBE ? PM_MEM_DEVICE : mem);
        if (rc != PM_OK) return rc;
        return pm_map_cube(ctx, cube, dtype, n_planes, x_map, y_map, n0, n1, interpolation, propagate_nan, out, mem);
    }
    for (int i = 0; i < 3; i++)
        if (!(ctx->geometry.radii[i] + alt > 0.0)) return fail(ctx, PM_ERR_INVALID_ARGUMENT, "radii + alt must be positive");
    if (ctx->pending && (size_t)n_planes > ctx->flags_count) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }
    rc = ensure_flags(ctx, (size_t)n_planes);
    if (rc != PM_OK) return rc;
    const bool force_sync = interpolation == PM_INTERP_LINEAR && !propagate_nan;
    if (force_sync && ctx->pending) {
        rc = pm_synchronize(ctx);
        if (rc != PM_OK) return rc;
    }
    pm::Params p;
    fill_params(ctx, alt, p);
    p.n0 = n0;
    p.n1 = n1;
    pm::ReprojectArgs a;
    a.cube = cube;
    a.x_map = x_map;
    a.y_map = y_map;
    a.out = out;
    a.plane_flags = ctx->flags;
    a.seq = ++ctx->map_seq;
    a.plane_stats = nullptr;
    a.n_planes = n_planes;
    a.ny = d.ny;
    a.nx = d.nx;
    a.n_map = n0 * n1;
    a.interpolation = interpolation;
    a.propagate_nan = propagate_nan ? 1 : 0;
    pm_launch_mapped_data(p, a, lon_deg, lat_deg, x_map, y_map, dtype, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    if (force_sync) return finish_reproject(ctx, a, dtype);
    ctx->pending = true;
    ctx->pending_args = a;
    ctx->pending_dtype = dtype;
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
