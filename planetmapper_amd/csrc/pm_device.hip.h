// pm_device.hip.h -- per-point device math of the planetmapper hot path (gfx950).
//
// Everything is IEEE binary64. One lane = one pixel (or one map location); all
// geometry constants arrive through the kernel-argument block (scalar loads ->
// SGPRs), so the only vector memory traffic of the image kernels is the final
// coalesced plane stores.
//
// Each function names the reference Python / CSPICE routine it evaluates
// (paths relative to the reference checkout).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/planetmapper_hip.h"
#include "pm_fastmath.hip.h"

namespace pm {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kHalfPi = 0.5 * kPi;
constexpr double kDeg = 180.0 / kPi;  // numpy rad2deg factor
constexpr double kRad = kPi / 180.0;  // numpy deg2rad factor

struct V3 {
    double x, y, z;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V3 ld3(const double *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ bool finite3(V3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

struct M3 {
    double m[9];  // row-major
};
__device__ __forceinline__ V3 mxv(const M3 &R, V3 v)
{
    return {fma(R.m[0], v.x, fma(R.m[1], v.y, R.m[2] * v.z)),
            fma(R.m[3], v.x, fma(R.m[4], v.y, R.m[5] * v.z)),
            fma(R.m[6], v.x, fma(R.m[7], v.y, R.m[8] * v.z))};
}
__device__ __forceinline__ V3 mtxv(const M3 &R, V3 v)
{
    return {fma(R.m[0], v.x, fma(R.m[3], v.y, R.m[6] * v.z)),
            fma(R.m[1], v.x, fma(R.m[4], v.y, R.m[7] * v.z)),
            fma(R.m[2], v.x, fma(R.m[5], v.y, R.m[8] * v.z))};
}
__device__ __forceinline__ V3 mxv(const double *m, V3 v)
{
    return {fma(m[0], v.x, fma(m[1], v.y, m[2] * v.z)), fma(m[3], v.x, fma(m[4], v.y, m[5] * v.z)),
            fma(m[6], v.x, fma(m[7], v.y, m[8] * v.z))};
}
__device__ __forceinline__ V3 mtxv(const double *m, V3 v)
{
    return {fma(m[0], v.x, fma(m[3], v.y, m[6] * v.z)), fma(m[1], v.x, fma(m[4], v.y, m[7] * v.z)),
            fma(m[2], v.x, fma(m[5], v.y, m[8] * v.z))};
}

// Kernel-argument block shared by all kernels (host fills it in pm_capi.hip).
struct Params {
    pm_geometry g;
    double A[6];      // xy -> angular affine            BodyXY._get_xy2angular_matrix body_xy.py:354
    double Ai[6];     // angular -> xy affine            BodyXY._get_angular2xy_matrix body_xy.py:371
    double K[4];      // angular -> km 2x2               Body._get_angular2km_matrix   body.py:1637
    double r2;        // squared radius of the pre-mask  body_xy.py:3201-3203
    double x0, y0;    // disc centre
    double radii[3];  // altitude-adjusted radii         _AdjustedSurfaceAltitude      body.py:172
    double t0;        // et - lt_c
    int32_t nx, ny;
    int32_t optimize_speed;
    int32_t n0, n1;   // map shape (map kernels)
    uint64_t mask;
    double *out[PM_NUM_PLANES];
    // Spheroid fast path (radii[0] == radii[1]): everything expressed in the body-fixed
    // frame frozen at t0 ("B0"); a spheroid is invariant under the spin about its z
    // axis, so the light-time iteration needs no rotations at all and the spin enters
    // only as a longitude offset at the end. Host-precomputed in pm_capi.hip.
    double C[9];    // R0 M^T : angular-frame unit vector -> ray in B0
    double O0[3];   // -R0 T0 : observer position in B0 at t0
    double VB[3];   // R0 VT
    double AB[3];   // R0 AT
    double SB0[3];  // R0 S0
    double VSB[3];  // R0 VS
    double ASB[3];  // R0 AS
    double VOB[3];  // R0 VO
    double VSB_state[3];  // R0 (VT + DVT): the target centre's STATE velocity (radial velocity)
    double ASB_state[3];  // R0 (AT + DAT)
    double WPB[3];        // R0 WP: the drift of the pole (STATE planes)
    double ira, irc;  // 1 / radii[0], 1 / radii[2]
    double inv_c;     // 1 / clight
    double lat_k;     // (radii[0] / radii[2])^2
    double O0s[3];    // O0 / radii, VB / radii: the observer in the scaled frame of surfpt_c is
    double VBs[3];    //   Y(d) = O0s - VBs d (spheroid fast path)
    double a_over_c;  // radii[0] / radii[2]
    double sun_ds0;   // Sun emission epoch of a surface point q (wrt P_T(t0), B0), linearised in q:
    double sun_k;     //   ds = d + sun_ds0 + (SB0 . q) sun_k,  sun_k = 1 / (|SB0| c); see k_disc_sph
    double ir[3];     // 1 / radii[i]
    double limb_n[2]; // surface-normal scalings of recpgr_surface: (m/a)^2, (m/c)^2, m = min(a, c)
    int32_t row_stride;  // row visiting order of the image kernels (coprime with `rows`)
    int32_t y_off;       // first image row of this launch (row-block sharding), normally 0
    int32_t rows;        // rows computed by this launch (<= ny); output row r holds image row y_off + r
    int32_t view_tiny;   // every view angle of the frame is below 1e-3 rad: sincos_tiny applies
    double Ar[6];        // A in radians, first row negated: pixel -> (-ax, ay) of _xy2obsvec_norm
    double lon_k[2];     // {w, w wdot}, w = -1 for west-positive bodies: lon = w (theta - wdot d)
    // n mod d on the scalar unit: q = umulhi(n, ceil(2^32 / d)) is the quotient or one more
    uint32_t row_magic;  // d = rows
    uint32_t col_blocks, col_magic;  // d = column blocks of the spheroid kernel, ceil(nx / kSphBlock)
    uint32_t view_direct;  // every view angle of the frame is below 1.5 rad: recrad(radrec(angles)) = angles (sky_block)
    double ring_nb[3], sub_obs_b[3], sub_ray_b[3];  // R0 ring_n, R0 sub_obsvec, R0 sub_ray: ring block in B0
    double lt_tol;       // CSPICE's light-time stopping rule: 1e-17 |et - lt|  (lt varies by 1e-9 relative over a disc)
    // Phase angle of the spheroid fast path as a series in the cosine: over a disc seen from afar the
    // phase angle g moves by ~R (1/D_obs + 1/D_sun) ~ 1e-4 rad around its value g0 at the body centre, so
    // g = acos(c) is a short Taylor series about c0 = cos g0 (host: pm_capi.hip fill_params, with the
    // guard that decides whether it applies): ph[0] = c0, ph[1] = g0, ph[2..5] = acos^(n)(c0) / n!
    double ph[6];
    int32_t phase_series;  // 1: the series holds to < 1e-15 rad over this frame's disc
    // 1: the spheroid fast path walks the reference's own sequence of light-time epochs (no Newton step on
    // the seed). Epochs et - lt are doubles: one quantum of them (6e-8 s at et = 3.8e8 s) moves the target
    // by |VT| ulp(t0), and two light times 1e-10 s apart round to different quanta in 0.1 % of the pixels.
    // Where that jump is visible on the body (Mars seen from Earth in 2012: 2.4e-8 deg per quantum) only the
    // reference's own iterates reproduce its choice of quantum; where it is not (Jupiter 2005: 3e-10 deg),
    // the shorter sequence is taken. Host: pm_capi.hip fill_params.
    // (2, tools/ only: the Newton step on the seed without the closed form in front of it)
    int32_t plain_lt;
    // Closed-form light time of the spheroid fast path (k_disc_sph): the observer in the scaled frame when the
    // target is taken at the epoch the light left it, Y = Y00 + s Wc for an intercept s km down the ray.
    double Y00[3];    // O0s - VBs (et - t0)
    double Y00lo[3];  // ... and what its rounding to binary64 dropped
    double Wc[3];     // VBs / c
    double lt_c_eff;  // et - t0 (exact: t0 = fl(et - lt_c) is within a factor 2 of et)
    double p2_lo, p2_hi;  // 1 -+ the change of the squared impact parameter P.P over the light-time span of a disc (with margin)
    // triaxial / general variants of k_disc_sph (Newton step on the light-time seed)
    double tri_k;      // wdot (b / a - a / b): the turn of the shape under the ray, per unit Xf_x Xf_y
    double p2_lo_rot;  // p2_lo widened by that turn (and the target's acceleration) over a light-time span
    int32_t turn_quantum;  // 1: one quantum of the epoch et - lt TURNS the body by more than 1e-9 deg (a fast rotator, or any
                           // planet late enough in the century): illumination and state are placed at the epoch of illumf_c's /
                           // spkcpt_c's own light-time solution (BODY 1 / 2: in full; a spheroid: its normal and point turned)
    int32_t cf_iter;  // 1: one epoch quantum is visible on the body and the library is left to choose (PM_OPT_LT_MODE 0): the
                      // closed form steps through the reference's sequence of iterates to land on ITS final epoch
    int32_t pad_cf_;
    int32_t tri_cf;  // 1: a triaxial body takes the closed-form light time with a first-order turn of its shape (k_disc_sph, BODY 1)
};

// n mod d for wave-uniform operands without the VALU float-reciprocal sequence hipcc expands
// `%` into: magic = ceil(2^32 / d) from the host (pmh::mod_magic), n < 2^32, d >= 2.
__device__ __forceinline__ uint32_t mod_uniform(uint32_t n, uint32_t d, uint32_t magic)
{
    const int32_t r = (int32_t)(n - __umulhi(n, magic) * d);  // in [-d, d)
    return (uint32_t)(r < 0 ? r + (int32_t)d : r);
}

constexpr int kBlock = 256;
// the spheroid image kernel runs one wave per workgroup: finer-grained dispatch mixes the
// cheap and the expensive row segments better (measured 0.269 vs 0.275 ms at 256, 0.291 at 512)
constexpr int kSphBlock = 64;

// Arguments of the point-transform kernel (pm_transform).
struct TransformArgs {
    const double *a, *b;
    double *oa, *ob;
    unsigned long long n;
    double alt;
    double radii0[3];  // radii WITHOUT the altitude adjustment (p.radii holds radii + alt)
    double Kf[4];      // km -> angular (p.K is angular -> km)
    int from, to, flags;
};

// Per-plane statistics for the NaN pre-clean of map_img (k_median_* kernels).
struct PlaneStats {
    unsigned long long prefix[2];  // radix-select state of the two middle ranks
    unsigned long long rank[2];
    unsigned long long n_finite;
    unsigned long long n_nan;
    double median;                 // np.nanmedian of the plane (0.0 if nothing finite)
    int all_nan;                   // np.all(np.isnan(plane))
    int needs_median;              // lazy form (pm_launch_clean_lazy): some pixel's clean value IS the median - compute it
};

// Spline (RectBivariateSpline, s = 0) reprojection: per-axis knots + banded LU of the
// collocation matrix, built on the host (pm_capi.hip), resident in device memory.
struct SplineAxis {
    const double *t;   // n + k + 1 knots
    const double *lu;  // n x (2k + 1) banded LU (unit lower), entry (i, j) at lu[i*(2k+1) + (j-i+k)]; the RECIPROCAL of U's diagonal at j = i
    int n, k;
};
struct SplineArgs {
    double *work;  // n_planes x ny x nx cleaned image -> B-spline coefficients (in place)
    SplineAxis rows, cols;  // axis 0 (image y) / axis 1 (image x)
    // the segmented solves of few, large planes (k_spline_seg_*): samples per segment along axis 0 / axis 1 (0: the
    // one-lane-per-line solves, in place), and the second buffer their forward passes write into
    int seg_rows = 0, seg_cols = 0;
    double *work2 = nullptr;
};
// How far before (beyond) a segment its forward (backward) substitution starts: the homogeneous recursion of the factors decays
// by the symbol's root per step (degree 2: 0.172, 3: 0.268, 4: 0.361, 5: 0.431) - what is left at the segment is < 1e-34 of the
// values. (1e-22 is not enough for the SAME BITS: a perturbation of that size changes the rounding of one operation in ~2e5,
// and a cube has millions of segment starts - the first lengths, 32 / 48 / 48 / 64, left one sample of 1.1 million an ulp off.)
constexpr int kSolveLines = 64;        // lines per wave of the spline solves
constexpr long kSolveFillWaves = 1024;  // one wave per SIMD of the chip (256 CUs x 4)
__host__ __device__ inline int spline_warm(int k) { return k <= 1 ? 16 : k == 2 ? 48 : k == 3 ? 64 : k == 4 ? 80 : 96; }

// Arguments of the reprojection kernel (pm_map_cube).
struct ReprojectArgs {
    const void *cube;     // n_planes x ny x nx elements
    const double *x_map;  // n_map
    const double *y_map;
    double *out;          // n_planes x n_map
    int *plane_flags;     // n_planes: set to `seq` (atomicMax) when a sampled pixel of the plane
                          // needs the plane's nanmedian, i.e. the call must be finished with
                          // plane statistics; never cleared, seq only grows
    int seq;              // sequence number of this pm_map_cube call (> 0)
    const PlaneStats *plane_stats;  // n_planes (CLEAN kernels only)
    int n_planes, ny, nx;
    int n_map;
    int interpolation;
    int propagate_nan;
};

// The sparse host path of pm_map_cube (pm_hostpipe.hip; k_mark_blocks / k_fetch_blocks /
// k_reproject_blocks): the blocks of a plane that the map samples - the same in every plane of the
// cube - and the table they are collected into. Blocks are 16 bytes when CPU threads collect them
// into pinned staging (the link then carries little more than the sampled pixels), 128 bytes when
// the GPU fetches them from pinned host memory itself (PM_OPT_FETCH_BLOCK_BYTES: the PCIe read granularity).
constexpr int kBlkShiftHost = 4;  // (the GPU's own fetch blocks: PM_OPT_FETCH_BLOCK_BYTES, 128 bytes by default)
struct BlockTable {
    const int *blkmap;   // [plane_bytes >> shift] block of the plane -> row of the table, -1 = not in it
    const int *blklist;  // [n_list] row of the table -> block of the plane (k_fetch_blocks only)
    char *table;         // [planes of the chunk][n_list][1 << shift] the blocks
    unsigned n_list;
    int shift;
    size_t plane_bytes;  // a multiple of the block size
};

// One axis of the oversampled grid of 'smooth' interpolation (get_xy_pchip
// body_xy.py:1724-1741): the original pixel coordinates first..last, optionally refined
// to `num` points with numpy.linspace arithmetic (i * step + first, last point exact).
constexpr int kMapLimitsBlocks = 1024;  // pm_launch_map_limits: at most this many partial results
struct SmoothAxis {
    int first, last;  // trimmed original range [first, last] (pixels within 5 of the map's footprint)
    int num;          // grid points
    int oversampled;  // 0: grid = first, first+1, ..., last
    double step;
};
struct SmoothArgs {
    SmoothAxis x, y;
    int general;  // PM_OPT_GENERAL_KERNEL: every cell takes the gap-aware form (the cross-check of the 4 x 4 form)
    int planes_per_lane;  // set by the launcher
};
__device__ __forceinline__ double smooth_grid(const SmoothAxis &ax, int i)
{
    if (!ax.oversampled) return (double)(ax.first + i);
    if (i == ax.num - 1) return (double)ax.last;
    return __dadd_rn(__dmul_rn((double)i, ax.step), (double)ax.first);  // no FMA: numpy rounds twice
}
// interval i with grid[i] <= v < grid[i+1] (last interval closed), grid[0] <= v <= grid[num-1]
__device__ __forceinline__ int smooth_interval(const SmoothAxis &ax, double v)
{
    int i = ax.oversampled ? (int)floor(div_fast(v - (double)ax.first, ax.step)) : (int)floor(v) - ax.first;  // (a first guess)
    i = i < 0 ? 0 : (i > ax.num - 2 ? ax.num - 2 : i);
    while (i > 0 && v < smooth_grid(ax, i)) i--;
    while (i < ax.num - 2 && v >= smooth_grid(ax, i + 1)) i++;
    return i;
}

// scipy.interpolate.PchipInterpolator restated piecewise (oracle: pchip_1d): Fritsch-Carlson
// derivative at an interior sample / Moler's three-point rule at an end sample
__device__ __forceinline__ double pchip_sign(double v) { return (double)((v > 0.0) - (v < 0.0)); }
// (The kernel that calls these is bound by their arithmetic. Samples without gaps are one pixel apart: a division by 1.0
//  is the identity and is skipped, bit for bit; the others use the Newton division of pm_fastmath.hip.h, within an ulp of
//  the IEEE quotient - the oracle, which restates scipy's arithmetic to the bit, stays the checker at 1e-9 relative.)
__device__ __forceinline__ double pchip_over(double a, double h) { return h == 1.0 ? a : div_fast(a, h); }
__device__ __forceinline__ double pchip_interior(double h0, double h1, double m0, double m1)
{
    // zero unless the slopes are both positive or both negative (scipy: sign(m0) != sign(m1) | m0 == 0 | m1 == 0; slopes of
    // finite samples are never NaN) - four compares instead of two sign() values and three
    if (!(m0 > 0.0 ? m1 > 0.0 : (m0 < 0.0 && m1 < 0.0))) return 0.0;
    const double w1 = 2.0 * h1 + h0, w2 = h1 + 2.0 * h0;
    // the weighted harmonic mean (w1 + w2) / (w1 / m0 + w2 / m1) with ONE division: m0 / (w1 m1 + w2 m0) is below 1 / w2
    // (the slopes have one sign), so nothing overflows that the four-division form would not. (The fma is spelt out: which
    // of the two products the compiler fuses must not depend on the caller - the two smooth kernels give the same bits.)
    return (w1 + w2) * div_fast(m0, __builtin_fma(w1, m1, w2 * m0)) * m1;
}
__device__ __forceinline__ double pchip_edge(double h0, double h1, double m0, double m1)
{
    const double d = div_fast((2.0 * h0 + h1) * m0 - h0 * m1, h0 + h1);
    if (pchip_sign(d) != pchip_sign(m0)) return 0.0;
    if (pchip_sign(m0) != pchip_sign(m1) && fabs(d) > 3.0 * fabs(m0)) return 3.0 * m0;
    return d;
}
// Value at v in [xb, xc] of the PCHIP interpolant whose samples around v are
// (xa, ya)?, (xb, yb), (xc, yc), (xd, yd)?: the piece on [xb, xc] depends on nothing else.
__device__ __forceinline__ double pchip_piece(bool has_a, double xa, double ya, double xb, double yb, double xc,
                                              double yc, bool has_d, double xd, double yd, double v)
{
    const double h = xc - xb, slope = pchip_over(yc - yb, h);
    double db = slope, dc = slope;  // two samples in all: straight line
    if (has_a || has_d) {
        const double hab = xb - xa, hcd = xd - xc;
        const double mab = has_a ? pchip_over(yb - ya, hab) : 0.0, mcd = has_d ? pchip_over(yd - yc, hcd) : 0.0;
        db = has_a ? pchip_interior(hab, h, mab, slope) : pchip_edge(h, hcd, slope, mcd);
        dc = has_d ? pchip_interior(h, hcd, slope, mcd) : pchip_edge(h, hab, slope, mab);
    }
    const double t = pchip_over(db + dc - 2.0 * slope, h);
    const double c0 = pchip_over(t, h), c1 = pchip_over(slope - db, h) - t;
    const double s = v - xb;
    double z = s, res = yb;
    res += db * z;
    z *= s;
    res += c1 * z;
    z *= s;
    res += c0 * z;
    return res;
}
// The piece on [b, b+1] of four consecutive finite samples one pixel apart (ya, yb, yc, yd): pchip_piece's arithmetic for
// has_a = has_d = true and h = hab = hcd = 1, split into the coefficients (the same for every point of the piece) and the
// evaluation at s = v - b.
struct PchipUnitPiece {
    double yb, db, c1, c0;
    __device__ __forceinline__ PchipUnitPiece(double ya, double yb_, double yc, double yd) : yb(yb_)
    {
        const double slope = yc - yb, mab = yb - ya, mcd = yd - yc;
        db = pchip_interior(1.0, 1.0, mab, slope);
        const double dc = pchip_interior(1.0, 1.0, slope, mcd);
        c0 = db + dc - 2.0 * slope;
        c1 = (slope - db) - c0;
    }
    __device__ __forceinline__ double operator()(double s) const
    {
        double z = s, res = yb;
        res += db * z;
        z *= s;
        res += c1 * z;
        z *= s;
        res += c0 * z;
        return res;
    }
};
// PCHIP through the finite samples val(lo..hi) (integer abscissae), evaluated at v;
// NaN outside the first..last finite sample (extrapolate=False) or with < 2 of them.
// Only the <= 4 finite samples around v are looked up.
template <typename F>
__device__ __forceinline__ double pchip_gappy(F val, int lo, int hi, double v)
{
    const double nan = __builtin_nan("");
    if (!(v >= (double)lo && v <= (double)hi)) return nan;
    const int fl = (int)floor(v);
    int ib = fl, ic = fl + 1;
    double yb = nan, yc = nan;
    for (;; ib--) {
        if (ib < lo) return nan;
        yb = val(ib);
        if (isfinite(yb)) break;
    }
    for (; ic <= hi; ic++) {
        yc = val(ic);
        if (isfinite(yc)) break;
    }
    if (ic > hi) {
        // v is at or beyond the last finite sample: the last interval is closed
        if (v != (double)ib) return nan;
        ic = ib;
        yc = yb;
        for (ib = ic - 1;; ib--) {
            if (ib < lo) return nan;
            yb = val(ib);
            if (isfinite(yb)) break;
        }
    }
    int ia = ib - 1, id = ic + 1;
    double ya = nan, yd = nan;
    for (; ia >= lo; ia--) {
        ya = val(ia);
        if (isfinite(ya)) break;
    }
    for (; id <= hi; id++) {
        yd = val(id);
        if (isfinite(yd)) break;
    }
    return pchip_piece(ia >= lo, (double)ia, ya, (double)ib, yb, (double)ic, yc, id <= hi, (double)id, yd, v);
}

// ------------------------------------------------------------------ CSPICE basics
// radrec_c
__device__ __forceinline__ V3 radrec(double ra, double dec)
{
    double sr, cr, sd, cd;
    sincos(ra, &sr, &cr);
    sincos(dec, &sd, &cd);
    return {cr * cd, sr * cd, sd};
}
// recrad_c: RA in [0, 2pi), Dec
__device__ __forceinline__ void recrad(V3 v, double &ra, double &dec)
{
    dec = atan2(v.z, sqrt(fma(v.x, v.x, v.y * v.y)));
    ra = (v.x == 0.0 && v.y == 0.0) ? 0.0 : atan2(v.y, v.x);
    if (ra < 0.0) ra += kTwoPi;
}
// radrec / recrad for the all-pixel kernels: same arithmetic, with the range-aware sincos,
// one-division atan2 and Newton square root of pm_fastmath.hip.h (v must be finite, non-zero)
__device__ __forceinline__ V3 radrec_f(double ra, double dec)
{
    double sr, cr, sd, cd;
    sincos_auto(ra, sr, cr);
    sincos_auto(dec, sd, cd);
    return {cr * cd, sr * cd, sd};
}
__device__ __forceinline__ void recrad_f(V3 v, double &ra, double &dec)
{
    dec = atan2_fast(v.z, sqrt_fast(fma(v.x, v.x, v.y * v.y)));
    ra = atan2_fast(v.y, v.x);
    if (ra < 0.0) ra += kTwoPi;
}
__device__ __forceinline__ double norm_f(V3 a) { return sqrt_fast(dot(a, a)); }
// vsep_c for two UNIT vectors
__device__ __forceinline__ double vsep_unit(V3 u, V3 v)
{
    double d = dot(u, v);
    if (d > 0.0) return 2.0 * asin(0.5 * norm(u - v));
    if (d < 0.0) return kPi - 2.0 * asin(0.5 * norm(u + v));
    return kHalfPi;
}
__device__ __forceinline__ V3 unit(V3 a)
{
    double n = norm(a);
    double s = (n > 0.0) ? 1.0 / n : 0.0;
    return s * a;
}
// a / |a| for a finite non-zero vector (reciprocal square root + Newton, pm_fastmath.hip.h)
__device__ __forceinline__ V3 unit_f(V3 a) { return rsqrt_fast(dot(a, a)) * a; }
// vsep_c of two UNIT vectors with asin on |x| <= 0.5 only: 2 asin(|u - v| / 2) below 60 deg (and the
// supplement form above 120 deg) like CSPICE's vsep_c, pi/2 - asin(u . v) in between
__device__ __forceinline__ double vsep_fast(V3 u, V3 v)
{
    const double d = dot(u, v);
    const bool mid = fabs(d) < 0.5;
    // Neighbouring pixels have neighbouring angles: most waves sit wholly inside (or wholly outside)
    // the 60..120 deg band, and then the half-chord, its square root and every select below are dead
    // weight. Same operations on the same operands as the general form: the same bits whatever the
    // wave's other lanes hold (the wave votes below only skip work, they never change a lane's result).
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(true), mid_m = __builtin_amdgcn_ballot_w64(mid);
    if (mid_m == lanes) return kHalfPi - asin_half(d);
    // sin^2 of half the angle to the nearer of v and -v: |u -+ v|^2 / 4 = (1 - |u . v|) / 2 for unit
    // vectors. The short form loses relative accuracy as the angle closes (the 1e-16 of the dot
    // product against 1 - |d|): a lane takes it while its own 1 - |d| > 1e-4 (angles beyond 0.8 deg
    // from 0 / 180: the error stays below 2e-14 rad) and the difference form of CSPICE's vsep_c
    // otherwise. The choice is PER LANE - a point's bits do not depend on which points share its
    // wave; the difference form is only evaluated in waves where some lane needs it.
    const double h = fma(-0.5, fabs(d), 0.5);
    const unsigned long long close_m = ~mid_m & __builtin_amdgcn_ballot_w64(!(h > 5e-5));  // (votes on the masks: wave_any)
    double s = sqrt_fast(h);
    if (close_m != 0) {
        const double sg = (d > 0.0) ? -1.0 : 1.0;
        const V3 w = {fma(sg, v.x, u.x), fma(sg, v.y, u.y), fma(sg, v.z, u.z)};
        const double sd = 0.5 * sqrt_fast(dot(w, w));
        s = __builtin_amdgcn_inverse_ballot_w64(close_m) ? sd : s;
    }
    if (mid_m == 0) {
        const double r = asin_half(s);
        return d > 0.0 ? r + r : fma_m2_c(r, kPi);
    }
    const double r = asin_half(mid ? d : s);
    return mid ? kHalfPi - r : (d > 0.0 ? r + r : fma_m2_c(r, kPi));
}


// ------------------------------------------------------------------ time-dependent state
// R(t) = rot3(wdot (t - t0)) R0   (pxform / pxfrm2 of the reference, body.py:940-1000)
// `small` selects the series form, exact to < 1e-20 for |angle| < 1e-3 rad (the <= 1 s
// light-time spans across a disc); otherwise full sincos.
template <bool SMALL>
__device__ __forceinline__ void rot_at(const Params &p, double t, M3 &R)
{
    double ang = p.g.wdot * (t - p.t0);
    double s, c;
    // (the form is the lane's own - a point's bits do not depend on its wave's other lanes -, the wave only
    //  decides which forms are evaluated)
    const bool tiny = fabs(ang) < 1e-3;
    const unsigned long long tiny_m = __builtin_amdgcn_ballot_w64(tiny);
    if (SMALL ? tiny : tiny_m != 0) {
        double a2 = ang * ang;
        s = ang * fma(a2, fma(a2, 1.0 / 120.0, -1.0 / 6.0), 1.0);
        c = fma(a2, fma(a2, fma(a2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    } else if (SMALL) {
        sincos(ang, &s, &c);
    }
    if (!SMALL && tiny_m != __builtin_amdgcn_ballot_w64(true)) {
        double s2, c2;
        sincos_auto(ang, s2, c2);  // (range-aware polynomials by lane, libm beyond 1e5)
        s = tiny ? s : s2;
        c = tiny ? c : c2;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        R.m[j] = fma(c, p.g.R0[j], s * p.g.R0[3 + j]);
        R.m[3 + j] = fma(c, p.g.R0[3 + j], -s * p.g.R0[j]);
        R.m[6 + j] = p.g.R0[6 + j];
    }
}
// target centre wrt observer(et) at epoch t
__device__ __forceinline__ V3 target_at(const Params &p, double t)
{
    double d = t - p.t0;
    double h = 0.5 * d * d;
    return {fma(p.g.AT[0], h, fma(p.g.VT[0], d, p.g.T0[0])), fma(p.g.AT[1], h, fma(p.g.VT[1], d, p.g.T0[1])),
            fma(p.g.AT[2], h, fma(p.g.VT[2], d, p.g.T0[2]))};
}
// displacement of the target centre between t0 and t
__device__ __forceinline__ V3 target_shift(const Params &p, double t)
{
    double d = t - p.t0;
    double h = 0.5 * d * d;
    return {fma(p.g.AT[0], h, p.g.VT[0] * d), fma(p.g.AT[1], h, p.g.VT[1] * d), fma(p.g.AT[2], h, p.g.VT[2] * d)};
}
__device__ __forceinline__ V3 sun_at(const Params &p, double t)
{
    double d = t - p.g.ts0;
    double h = 0.5 * d * d;
    return {fma(p.g.AS[0], h, fma(p.g.VS[0], d, p.g.S0[0])), fma(p.g.AS[1], h, fma(p.g.VS[1], d, p.g.S0[1])),
            fma(p.g.AS[2], h, fma(p.g.VS[2], d, p.g.S0[2]))};
}

// ------------------------------------------------------------------ ellipsoid
// surfpt_c: nearest intersection of ray (o, u) with the ellipsoid, observer outside or
// inside. Returns false if the ray misses.
__device__ __forceinline__ bool surfpt(V3 o, V3 u, const double *radii, V3 &pt)
{
    V3 X = {u.x / radii[0], u.y / radii[1], u.z / radii[2]};
    V3 Y = {o.x / radii[0], o.y / radii[1], o.z / radii[2]};
    double xx = dot(X, X);
    if (xx == 0.0) return false;
    double yx = dot(Y, X);
    double k = yx / xx;
    V3 P = {fma(-k, X.x, Y.x), fma(-k, X.y, Y.y), fma(-k, X.z, Y.z)};
    double p2 = dot(P, P), y2 = dot(Y, Y);
    double sign;
    if (y2 > 1.0) {
        if (p2 > 1.0 || yx > 0.0) return false;
        sign = -1.0;
    } else if (y2 == 1.0) {
        pt = o;
        return true;
    } else {
        sign = 1.0;
    }
    double s = sign * sqrt(fmax(0.0, 1.0 - p2)) * rsqrt(xx);
    pt = {fma(s, X.x, P.x) * radii[0], fma(s, X.y, P.y) * radii[1], fma(s, X.z, P.z) * radii[2]};
    return true;
}

// surfpt_c with the host's reciprocals of the radii (ir[i] = 1 / radii[i], as the spheroid fast path
// scales its vectors) and the Newton square roots: six IEEE divisions and two libm roots fewer per
// light-time evaluation of the general kernels
__device__ __forceinline__ bool surfpt_ir(V3 o, V3 u, const double *radii, const double *ir, V3 &pt)
{
    V3 X = {u.x * ir[0], u.y * ir[1], u.z * ir[2]};
    V3 Y = {o.x * ir[0], o.y * ir[1], o.z * ir[2]};
    double xx = dot(X, X);
    if (xx == 0.0) return false;
    double yx = dot(Y, X);
    double k = div_fast(yx, xx);
    V3 P = {fma(-k, X.x, Y.x), fma(-k, X.y, Y.y), fma(-k, X.z, Y.z)};
    double p2 = dot(P, P), y2 = dot(Y, Y);
    double sign;
    if (y2 > 1.0) {
        if (p2 > 1.0 || yx > 0.0) return false;
        sign = -1.0;
    } else if (y2 == 1.0) {
        pt = o;
        return true;
    } else {
        sign = 1.0;
    }
    double s = sign * sqrt_fast(fmax(0.0, 1.0 - p2)) * rsqrt_fast(xx);
    pt = {fma(s, X.x, P.x) * radii[0], fma(s, X.y, P.y) * radii[1], fma(s, X.z, P.z) * radii[2]};
    return true;
}

// sincpt_c('ELLIPSOID', ..., 'CN', ..., ray): Body._obsvec_norm2targvec body.py:1008-1020.
// Converged-Newtonian light time: repeat the intercept at te = et - lt until the light
// time moves by <= 1e-17 |te| (CSPICE's rule), at most 10 evaluations. The contraction
// factor is (v/c) / cos(emission): ~4e-5 over most of the disc (3 evaluations), but
// it approaches 1 at grazing incidence, so limb lanes need more - a fixed count is
// not enough there. Outputs the body-fixed point and its light time.
__device__ __forceinline__ bool sincpt(const Params &p, V3 ray, V3 &sp, double &lt)
{
    lt = p.g.lt_c;
#pragma unroll 1
    for (int it = 0; it < 10; it++) {
        double te = p.g.et - lt;
        M3 R;
        rot_at<true>(p, te, R);
        V3 obs = neg(mxv(R, target_at(p, te)));
        V3 u = mxv(R, ray);
        if (!surfpt_ir(obs, u, p.radii, p.ir, sp)) return false;
        double nlt = norm_f(sp - obs) * p.inv_c;
        double err = fabs(nlt - lt);
        lt = nlt;
        if (err <= 1e-17 * fabs(p.g.et - lt)) break;
    }
    return true;
}

// recpgr_c for a point ON the surface (body.py:1030-1035): planetographic lon [0, 2pi),
// geodetic latitude from the surface normal (x/a^2, y/a^2, z/c^2).
__device__ __forceinline__ void recpgr_surface(const Params &p, V3 v, double &lon, double &lat)
{
    // (limb_n = (m / a)^2, (m / c)^2 with m = min(a, c): the host's copy of this scaling)
    double nx = v.x * p.limb_n[0], ny = v.y * p.limb_n[0], nz = v.z * p.limb_n[1];
    if (nx == 0.0 && ny == 0.0 && nz == 0.0) {
        lon = 0.0;
        lat = kHalfPi;
        return;
    }
    lat = atan2_fast<true>(nz, sqrt_fast(fma(nx, nx, ny * ny)));
    double l = (v.x == 0.0 && v.y == 0.0) ? 0.0 : atan2_fast(v.y, v.x);
    if (p.g.west_positive) l = -l;
    if (l < 0.0) l += kTwoPi;
    lon = l;
}

// nearpt_c restated for a spheroid (a, a, c): signed altitude of `v` and the geodetic
// latitude of its near point; Newton iteration on the Lagrange multiplier.
__device__ __forceinline__ void recpgr_general(const Params &p, V3 v, double &lon, double &lat, double &alt)
{
    const double a = p.radii[0], c = p.radii[2];
    const double a2 = a * a, c2 = c * c;
    double rho2 = fma(v.x, v.x, v.y * v.y);
    double rho = sqrt(rho2);
    double q = rho2 / a2 + (v.z * v.z) / c2;
    double bx, bz;  // near point in the meridian plane
    if (q == 0.0) {
        bool pole = c <= a;
        bx = pole ? 0.0 : a;
        bz = pole ? c : 0.0;
        alt = -fmin(a, c);
    } else {
        double lam;
        if (q >= 1.0) {
            lam = 0.0;
        } else {
            double l1 = (rho != 0.0) ? -a2 + a * rho : -1e300;
            double l2 = (v.z != 0.0) ? -c2 + c * fabs(v.z) : -1e300;
            lam = fmax(l1, l2);
        }
#pragma unroll 1
        for (int it = 0; it < 60; it++) {
            double da = a2 + lam, dc = c2 + lam;
            double ta = a * rho / da, tc = c * v.z / dc;
            double f = fma(ta, ta, fma(tc, tc, -1.0));
            double df = -2.0 * (ta * ta / da + tc * tc / dc);
            if (df == 0.0) break;
            double step = f / df;
            double nl = lam - step;
            if (nl == lam) break;
            lam = nl;
            if (fabs(step) <= 1e-16 * fabs(lam)) break;
        }
        bx = a2 * rho / (a2 + lam);
        bz = c2 * v.z / (c2 + lam);
        double s = sqrt(bx * bx / a2 + bz * bz / c2);
        bx /= s;
        bz /= s;
        double dx = rho - bx, dz = v.z - bz;
        alt = sqrt(fma(dx, dx, dz * dz));
        if (q < 1.0) alt = -alt;
    }
    double m = fmin(a, c);
    double a1 = m / a, c1 = m / c;
    double nr = bx * (a1 * a1), nz = bz * (c1 * c1);
    lat = (nr == 0.0 && nz == 0.0) ? kHalfPi : atan2(nz, nr);
    double l = (v.x == 0.0 && v.y == 0.0) ? 0.0 : atan2(v.y, v.x);
    if (p.g.west_positive) l = -l;
    if (l < 0.0) l += kTwoPi;
    lon = l;
}

// recpgr_c of an arbitrary point for a spheroid, fast form for the ring planes: signed
// altitude and EAST longitude only. Same Lagrange-multiplier equation as recpgr_general,
// but Newton starts from the tightest lower bound lam0 = max(a rho - a^2, c|z| - c^2, 0 if
// outside) - for ring-plane points (z ~ 0) that IS the root, so 1-2 steps suffice - and each
// step uses reciprocals instead of five divisions.
__device__ __forceinline__ void recpgr_alt_lon(const Params &p, V3 v, double &lon_east, double &alt, bool care = true)
{
    const double a = p.radii[0], c = p.radii[2];
    const double a2 = a * a, c2 = c * c;
    const double rho2 = fma(v.x, v.x, v.y * v.y);
    const double rho = sqrt_fast(rho2);
    lon_east = (rho2 == 0.0) ? 0.0 : atan2_fast(v.y, v.x);
    // Ring-plane intercepts lie within a few km of the equatorial plane (PM's obsvec -> targvec
    // transform is not exactly plane-preserving) and outside the body. For such a point the
    // multiplier of the near point is lam0 = a (rho - a) up to eps^2 = (c z / (c^2 + lam0))^2 ~ 1e-11
    // relative, which gives the near point (a (1 - eps^2 / 2), c eps) and the altitude in closed
    // form: dx sqrt(1 + (dz / dx)^2) = dx (1 + (dz / dx)^2 / 2) with (dz / dx)^2 <= 1e-6 under the
    // guard below (next term 1e-13 relative). The two reciprocals only scale 1e-5-sized
    // corrections: raw 2^-24 seeds do. This replaces the Newton iteration and its finish - ten
    // reciprocal / square-root seeds and ~90 FP64 operations per pixel of a ring frame.
    // (care: lanes whose result is used; the others must not veto the wave-uniform choice)
    // The form is each LANE's own (a point gets the closed form iff IT lies in the plane's vicinity, whatever
    // the other lanes of its wave hold); the wave only decides what is evaluated: the closed form alone when every
    // lane that matters takes it (ring frames), both with a select per lane otherwise.
    const bool near_plane = rho > a && fabs(v.z) <= 1e-3 * (rho - a);
    const unsigned long long plane_m = __builtin_amdgcn_ballot_w64(near_plane);
    const unsigned long long other_m = __builtin_amdgcn_ballot_w64(care) & ~plane_m;
    double alt_plane = 0.0;
    if (plane_m != 0) {
        const double dxa = rho - a;
        const double rc = __builtin_amdgcn_rcp(fma(a, dxa, c2));  // 1 / (c^2 + lam0)
        const double e = (c * v.z) * rc;
        const double dx = fma(0.5 * a * e, e, dxa);  // rho - a (1 - e^2 / 2)
        const double dz = fma(-c, e, v.z);
        const double t = dz * __builtin_amdgcn_rcp(dx);
        alt_plane = dx * fma(0.5 * t, t, 1.0);
    }
    if (other_m == 0) {
        alt = alt_plane;
        return;
    }
    const double q = rho2 * (p.ira * p.ira) + (v.z * v.z) * (p.irc * p.irc);
    const double l1 = (rho != 0.0) ? fma(a, rho, -a2) : -1e300;
    const double l2 = (v.z != 0.0) ? fma(c, fabs(v.z), -c2) : -1e300;
    double lam = fmax(fmax(l1, l2), (q >= 1.0) ? 0.0 : -1e300);
    const double ar = a * rho, cz = c * v.z;
    // Newton from a lower bound is monotone and quadratic; once the step falls to the rounding
    // floor (a few ulps of lam) it only dithers, so stop there: 1e-15 |lam| moves the near
    // point by < 1e-9 km.
    // (a lane that has stopped keeps its value while the wave goes on for the others: further steps would dither
    //  it by an ulp, and how many there are depends on the company)
    bool done = !(q > 0.0);  // (the centre itself, or NaN: nothing to iterate)
#pragma unroll 1
    for (int it = 0; it < 12; it++) {
        const double ra = rcp_fast(a2 + lam), rc = rcp_fast(c2 + lam);
        const double ta = ar * ra, tc = cz * rc;
        const double f = fma(ta, ta, fma(tc, tc, -1.0));
        const double df = -2.0 * fma(ta * ta, ra, tc * tc * rc);
        const double step = div_fast(f, df);
        const double nl = lam - step;
        const bool stop = (nl == lam) || fabs(step) <= 1e-15 * fabs(nl);
        lam = done ? lam : nl;
        done = done || stop;
        if (wave_all(done)) break;
    }
    double bx = a2 * rho * rcp_fast(a2 + lam), bz = c2 * v.z * rcp_fast(c2 + lam);
    const double s = rsqrt_fast(bx * bx * (p.ira * p.ira) + bz * bz * (p.irc * p.irc));
    bx *= s;
    bz *= s;
    const double dx = rho - bx, dz = v.z - bz;
    alt = sqrt_fast(fma(dx, dx, dz * dz));
    if (q < 1.0) alt = -alt;
    if (q == 0.0) alt = -fmin(a, c);
    alt = near_plane ? alt_plane : alt;
}

// pgrrec_c (body.py:903-910): surface point of the spheroid `radii` + alt along its normal
__device__ __forceinline__ V3 pgrrec_alt(const Params &p, const double *radii, double lon, double lat, double alt)
{
    double a = radii[0], b = radii[2];
    double le = p.g.west_positive ? -lon : lon;
    double sl, cl, so, co;
    sincos_auto(lat, sl, cl);  // degree-sized angles: pi/2-reduced polynomials, libm beyond 1e5 / NaN
    sincos_auto(le, so, co);
    double big = fmax(fabs(a * cl), fabs(b * sl));
    double x = a * cl / big, y = b * sl / big;
    double scale = 1.0 / (big * sqrt(fma(x, x, y * y)));
    return {fma(alt, co * cl, scale * a * a * co * cl), fma(alt, so * cl, scale * a * a * so * cl),
            fma(alt, sl, scale * b * b * sl)};
}

// pgrrec_c (body.py:903-910) at altitude 0
__device__ __forceinline__ V3 pgrrec_surface(const Params &p, double lon, double lat)
{
    double a = p.radii[0], b = p.radii[2];
    double le = p.g.west_positive ? -lon : lon;
    double sl, cl, so, co;
    sincos_auto(lat, sl, cl);  // degree-sized angles: pi/2-reduced polynomials, libm beyond 1e5 / NaN
    sincos_auto(le, so, co);
    double big = fmax(fabs(a * cl), fabs(b * sl));
    double x = a * cl / big, y = b * sl / big;
    double scale = 1.0 / (big * sqrt(fma(x, x, y * y)));
    return {scale * a * a * co * cl, scale * a * a * so * cl, scale * b * b * sl};
}

// ------------------------------------------------------------------ light time of a body-fixed point
// spkcpt_c / first half of illumf_c. `lt` is an initial guess on entry; NITER fixed-point
// passes (the contraction factor is v/c ~ 4e-5), then pos and R are evaluated at the
// final epoch. pos = point wrt observer, J2000.
template <int NITER>
__device__ __forceinline__ void point_lt(const Params &p, V3 sp, double &lt, V3 &pos, M3 &R)
{
#pragma unroll 1
    for (int it = 0; it < NITER; it++) {
        double te = p.g.et - lt;
        rot_at<true>(p, te, R);
        pos = target_at(p, te) + mtxv(R, sp);
        lt = norm_f(pos) * p.inv_c;
    }
    double te = p.g.et - lt;
    rot_at<true>(p, te, R);
    pos = target_at(p, te) + mtxv(R, sp);
}

// illumf_c(..., 'SUN', ..., 'CN', ...): Body._illumf_from_targvec_radians body.py:1915.
// Given the converged pos/R/lt of the point. Angles in radians.
__device__ __forceinline__ void illum_angles(const Params &p, V3 sp, double lt, V3 pos, const M3 &R,
                                             double &phase, double &inc, double &emi)
{
    double te = p.g.et - lt;
    V3 obsv = neg(mxv(R, pos));
    V3 q = target_shift(p, te) + mtxv(R, sp);
    // Sun light time (spkcpo_c, CN): two passes from the centre value; the Sun moves
    // 0.01 km/s so the second pass is already converged to 1e-9 km.
    double lts = te - p.g.ts0;
    V3 sv;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        sv = sun_at(p, te - lts) - q;
        lts = norm_f(sv) * p.inv_c;
    }
    sv = sun_at(p, te - lts) - q;
    V3 sunb = unit_f(mxv(R, sv));
    V3 ob = unit_f(obsv);
    // surfnm_c: sp / radii^2, normalised (the common scale of CSPICE's form drops out)
    V3 n = unit_f(v3(sp.x * (p.ir[0] * p.ir[0]), sp.y * (p.ir[1] * p.ir[1]), sp.z * (p.ir[2] * p.ir[2])));
    phase = vsep_fast(sunb, ob);
    inc = vsep_fast(n, sunb);
    emi = vsep_fast(n, ob);
}

// emission angle alone (the `visibl` flag of illumf_c: emission < 90 deg), for callers that
// need no Sun geometry: same operations as in illum_angles
__device__ __forceinline__ double emission_angle(const Params &p, V3 sp, V3 pos, const M3 &R)
{
    V3 ob = unit_f(neg(mxv(R, pos)));
    V3 n = unit_f(v3(sp.x * (p.ir[0] * p.ir[0]), sp.y * (p.ir[1] * p.ir[1]), sp.z * (p.ir[2] * p.ir[2])));
    return vsep_fast(n, ob);
}

// Body._azimuth_angle_from_gie_radians body.py:2319 on degree images (body_xy.py:3742)
__device__ __forceinline__ double azimuth_deg(double ph_deg, double in_deg, double em_deg)
{
    double phr = ph_deg * kRad, inr = in_deg * kRad, emr = em_deg * kRad;
    double ce = cos(emr), ci = cos(inr);
    double a = cos(phr) - ce * ci;
    double b = sqrt(1.0 - ce * ce) * sqrt(1.0 - ci * ci);
    return (kPi - acos(a / b)) * kDeg;
}

// acos(x) for |x| <= 1 with asin on |x| <= 0.5 only (vsep_fast's scheme): pi/2 - asin(x) in the middle, the
// half-angle form 2 asin(sqrt((1 - |x|) / 2)) towards the ends (1 - |x| is exact there)
__device__ __forceinline__ double acos_fast(double x)
{
    const bool mid = fabs(x) <= 0.5;
    const double s = sqrt_fast(fma(-0.5, fabs(x), 0.5));
    const double r = asin_half(mid ? x : s);
    return mid ? kHalfPi - r : (x > 0.0 ? 2.0 * r : kPi - 2.0 * r);
}

// Body._azimuth_angle_from_gie_radians body.py:2319-2332 from the COSINES of the three angles - the dot
// products the angles themselves were taken from - instead of cos() of the angles: pi - acos(q) = acos(-q),
// q = (cos g - cos e cos i) / (sin e sin i). NaN where the reference's arccos is: |q| > 1 or 0 / 0.
// (Three libm cosines, an arccos, two square roots and an IEEE division cost this one plane 0.09 ms of a
//  4096^2 frame - two thirds of the whole headline set.)
__device__ __forceinline__ double azimuth_from_cosines(double cg, double ci, double ce)
{
    const double a = fma(-ce, ci, cg);
    const double b2 = fma(-ce, ce, 1.0) * fma(-ci, ci, 1.0);
    const double q = a * rsqrt_fast(fmax(b2, 1e-300));
    const double az = acos_fast(-fmax(-1.0, fmin(1.0, q)));
    return (fabs(q) <= 1.0 && b2 > 0.0) ? az * kDeg : __builtin_nan("");
}

// spkcpt_c velocity with the light-time rate; Body._radial_velocity_from_state body.py:2847
__device__ __forceinline__ double radial_velocity(const Params &p, V3 sp, double lt, V3 pos, const M3 &R)
{
    double te = p.g.et - lt;
    double d = te - p.t0;
    V3 off = mtxv(R, sp);
    V3 z = {R.m[6], R.m[7], R.m[8]};
    V3 w = p.g.wdot * cross(z, off) + cross(ld3(p.g.WP), off);  // (+ the drift of the pole: pm_geometry.WP)
    // (a STATE's velocity: VT + DVT, see pm_geometry.DVT)
    V3 vp = {fma(p.g.AT[0] + p.g.DAT[0], d, p.g.VT[0] + p.g.DVT[0]) + w.x, fma(p.g.AT[1] + p.g.DAT[1], d, p.g.VT[1] + p.g.DVT[1]) + w.y,
             fma(p.g.AT[2] + p.g.DAT[2], d, p.g.VT[2] + p.g.DVT[2]) + w.z};
    V3 rh = unit_f(pos);
    V3 vo = ld3(p.g.VO);
    double dlt = div_fast(dot(rh, vp - vo) * p.inv_c, fma(dot(rh, vp), p.inv_c, 1.0));
    V3 vel = (1.0 - dlt) * vp - vo;
    return dot(vel, rh);
}

// ------------------------------------------------------------------ PM's own transforms
// Body._targvec2obsvec body.py:917-948
__device__ __forceinline__ V3 targvec2obsvec(const Params &p, V3 tv)
{
    V3 off = tv - ld3(p.g.sub_sp);
    double dist = norm_f(ld3(p.g.sub_ray) + off) - p.g.sub_dist;
    double t = p.g.sub_et - dist * p.inv_c;
    M3 R;
    rot_at<false>(p, t, R);
    return ld3(p.g.sub_obsvec) + mtxv(R, off);
}
// Body._obsvec2targvec body.py:972-1006
__device__ __forceinline__ V3 obsvec2targvec(const Params &p, V3 ov)
{
    V3 off = ov - ld3(p.g.sub_obsvec);
    double dist = norm_f(off - ld3(p.g.sub_ray)) - p.g.sub_dist;
    double t = p.g.sub_et - dist * p.inv_c;
    M3 R;
    rot_at<false>(p, t, R);
    return ld3(p.g.sub_sp) + mxv(R, off);
}

// Body._obsvec2angular body.py:1345-1361 (arcsec)
__device__ __forceinline__ void obsvec2angular(const Params &p, V3 ov, double &ax, double &ay)
{
    V3 w = mxv(p.g.M, ov);
    double ra, dec;
    recrad(w, ra, dec);
    double x = fmod(-(ra * kDeg), 360.0);
    if (x < 0.0) x += 360.0;
    if (x > 180.0) x -= 360.0;
    ax = x * 3600.0;
    ay = (dec * kDeg) * 3600.0;
}

// obsvec2angular for finite non-zero `ov`. RA from recrad is in [0, 2pi], so the reference's
// `-ra % 360` needs no fmod: one conditional add.
__device__ __forceinline__ void obsvec2angular_f(const Params &p, V3 ov, double &ax, double &ay)
{
    V3 w = mxv(p.g.M, ov);
    double ra, dec;
    recrad_f(w, ra, dec);
    double x = -(ra * kDeg);
    if (x < 0.0) x += 360.0;
    if (x > 180.0) x -= 360.0;
    ax = x * 3600.0;
    ay = (dec * kDeg) * 3600.0;
}

// Body._ring_coordinates_from_obsvec(only_visible=False) body.py:2577-2615
__device__ __forceinline__ void ring_coords(const Params &p, V3 ov, double &radius, double &lon_deg, double &dist)
{
    const double nan = __builtin_nan("");
    radius = lon_deg = dist = nan;
    if (!finite3(ov)) return;
    double n = norm(ov);
    if (n == 0.0) return;
    V3 u = (1.0 / n) * ov;
    double k = p.g.ring_k;
    double pd = dot(u, ld3(p.g.ring_n));
    V3 ip;
    if (k == 0.0) {
        if (pd == 0.0) return;
        ip = {0.0, 0.0, 0.0};
    } else {
        if (!(pd > 0.0)) return;
        if (k >= pd * (1.7976931348623157e308 / 3.0)) return;
        ip = (k / pd) * u;
    }
    V3 tv = obsvec2targvec(p, ip);
    double lon, lat, alt;
    recpgr_general(p, tv, lon, lat, alt);
    dist = norm(ip);
    lon_deg = lon * kDeg;
    radius = alt + p.radii[0];
}

// Body._limb_coordinates_from_obsvec body.py:2081-2110
__device__ __forceinline__ void limb_coords(const Params &p, V3 ray, double &lon_deg, double &lat_deg, double &dist)
{
    const double nan = __builtin_nan("");
    lon_deg = lat_deg = dist = nan;
    if (!finite3(ray)) return;
    V3 T0 = ld3(p.g.T0);
    double k = dot(T0, ray) / dot(ray, ray);
    V3 near = k * ray;
    double nd = norm(near - T0);
    V3 tv = obsvec2targvec(p, near);
    V3 s;
    if (!surfpt(v3(0.0, 0.0, 0.0), tv, p.radii, s)) return;
    double lon, lat;
    recpgr_surface(p, s, lon, lat);
    lon_deg = lon * kDeg;
    lat_deg = lat * kDeg;
    dist = nd - norm(s);
}


// Body.local_solar_time_from_lon body.py:2376-2398 (et2lst_c, truncated to whole seconds)
// fmod(x, 86400) for |x| < 3 * 86400 is at most two exact subtractions (Sterbenz), the divisions are
// div_fast (agrees with the IEEE quotient in every probed sample): the libm fmod and five IEEE divisions of
// the plain form cost this plane a fifth of the whole headline set.
__device__ __forceinline__ double local_solar_time(const Params &p, double lon_deg)
{
    if (!isfinite(lon_deg)) return __builtin_nan("");
    double lon = lon_deg * kRad;
    double le = p.g.west_positive ? -lon : lon;
    double angle = le - p.g.lst_sun_lon;
    double frac = div_fast(angle, kTwoPi) + 0.5;
    double secnds = 86400.0 * frac;  // |angle| < 4 pi: |secnds| < 3 * 86400
    if (!(fabs(secnds) < 259200.0)) {
        secnds = fmod(secnds, 86400.0);
    } else {
        const double m = fabs(secnds);
        const double r = m >= 172800.0 ? m - 172800.0 : (m >= 86400.0 ? m - 86400.0 : m);
        secnds = secnds < 0.0 ? -r : r;
    }
    if (secnds < 0.0) secnds += 86400.0;
    double hr = floor(div_fast(secnds, 3600.0));
    double rem = secnds - 3600.0 * hr;
    double mn = floor(div_fast(rem, 60.0));
    double sc = floor(rem - 60.0 * mn);
    return hr + div_fast(mn, 60.0) + div_fast(sc, 3600.0);
}



// ------------------------------------------------------------------ one map cell -> pixel coordinates
// Kernel-argument block read in place: constants are loaded where they are used instead of at kernel
// entry (hipcc otherwise parks them in VGPR lanes once the scalar registers run out). Valid in kernels
// whose FIRST argument is the Params block (offset 0 of the kernel-argument segment).
typedef const __attribute__((address_space(4))) Params *KParams;
__device__ __forceinline__ KParams kernarg_params()
{
    KParams kp = (KParams)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// Pixel coordinates (px, py) of the map cell (lon_deg, lat_deg), NaN where the cell is not visible or
// falls outside the frame: the chain BodyXY._get_targvec_map -> _get_illumf_map -> _get_obsvec_map ->
// _get_radec_map -> _get_xy_map (body_xy.py:3227-3300, 3419-3491, 3667) as k_map_xy / k_mapped_data
// evaluate it (derivation: pm_kernels.hip, "x/y map alone").
__device__ __forceinline__ void map_cell_xy(KParams kp, double lon_deg, double lat_deg, double &px_out, double &py_out)
{
    const double nan = __builtin_nan("");
    const bool have = isfinite(lon_deg) && isfinite(lat_deg);
    const double lon = have ? lon_deg * kRad : 0.0, lat = have ? lat_deg * kRad : 0.0;
    // pgrrec_c, altitude 0: (a^2 cos(lat) cos(l), a^2 cos(lat) sin(l), c^2 sin(lat)) / sqrt(a^2 cos^2 + c^2 sin^2)
    const double a = kp->radii[0], c = kp->radii[2];
    double sl, cl, so, co;
    sincos_auto(lat, sl, cl);
    sincos_auto(kp->g.west_positive ? -lon : lon, so, co);
    const double acl = a * cl, csl = c * sl;
    const double den = rsqrt_fast(fma(acl, acl, csl * csl));
    const double ha = a * acl * den;
    const V3 tv = {ha * co, ha * so, c * csl * den};

    // light time of the point, two passes from the centre value: pos(te) = T(te) + R(te)^T tv, in B0
    // w(d) = VB d + AB d^2 / 2 - O0 + Rz(wdot d)^T tv
    const double wdot = kp->g.wdot;
    // (the first pass, from the centre's light time, sits at t0 itself - d = (et - lt_c) - t0 = 0 exactly: the point as it
    //  stands, the bits of the general expression without its sincos and nine FMAs)
    V3 q = tv, w = {tv.x - kp->O0[0], tv.y - kp->O0[1], tv.z - kp->O0[2]};
    const double lt = sqrt_fast(dot(w, w)) * kp->inv_c;
    {
        const double d = (kp->g.et - lt) - kp->t0;
        const double h = 0.5 * d * d;
        const double ang = wdot * d;
        double sa, ca;
        sincos_tiered<true>(ang, sa, ca);
        q = {fma(ca, tv.x, -sa * tv.y), fma(sa, tv.x, ca * tv.y), tv.z};
        w = {fma(kp->AB[0], h, fma(kp->VB[0], d, q.x - kp->O0[0])), fma(kp->AB[1], h, fma(kp->VB[1], d, q.y - kp->O0[1])),
             fma(kp->AB[2], h, fma(kp->VB[2], d, q.z - kp->O0[2]))};
    }
    // visible <=> the outward normal (q / radii^2, turned like q) faces the observer (at -w from the point)
    const double facing = -fma(q.x * kp->ir[0] * kp->ir[0], w.x, fma(q.y * kp->ir[1] * kp->ir[1], w.y, q.z * kp->ir[2] * kp->ir[2] * w.z));
    const bool vis = have && facing > 0.0;

    // Body._targvec2obsvec: the point's own light-time offset from the sub-observer point
    const V3 off = {tv.x - kp->g.sub_sp[0], tv.y - kp->g.sub_sp[1], tv.z - kp->g.sub_sp[2]};
    const V3 sr = {kp->g.sub_ray[0] + off.x, kp->g.sub_ray[1] + off.y, kp->g.sub_ray[2] + off.z};
    const double dist = sqrt_fast(dot(sr, sr)) - kp->g.sub_dist;
    const double t = kp->g.sub_et - dist * kp->inv_c;
    const double ang2 = wdot * (t - kp->t0);
    double s2, c2;
    sincos_tiered<true>(ang2, s2, c2);
    // R0 ov = R0 sub_obsvec + Rz(ang2)^T off
    const V3 b = {kp->sub_obs_b[0] + fma(c2, off.x, -s2 * off.y), kp->sub_obs_b[1] + fma(s2, off.x, c2 * off.y),
                  kp->sub_obs_b[2] + off.z};
    // Body._obsvec2angular: M ov = C^T (R0 ov); recrad_c is scale free
    const V3 m = {fma(kp->C[0], b.x, fma(kp->C[3], b.y, kp->C[6] * b.z)), fma(kp->C[1], b.x, fma(kp->C[4], b.y, kp->C[7] * b.z)),
                  fma(kp->C[2], b.x, fma(kp->C[5], b.y, kp->C[8] * b.z))};
    double ra, dec;
    recrad_f(m, ra, dec);
    double xx = -(ra * kDeg);
    if (xx < 0.0) xx += 360.0;
    if (xx > 180.0) xx -= 360.0;
    const double ax = xx * 3600.0, ay = (dec * kDeg) * 3600.0;
    const double px = fma(kp->Ai[0], ax, fma(kp->Ai[1], ay, kp->Ai[2]));
    const double py = fma(kp->Ai[3], ax, fma(kp->Ai[4], ay, kp->Ai[5]));
    // BodyXY._xy_in_image_frame body_xy.py:1868
    const bool in_frame = vis && -0.5 < px && px < kp->nx - 0.5 && -0.5 < py && py < kp->ny - 0.5;
    px_out = in_frame ? px : nan;
    py_out = in_frame ? py : nan;
}

}  // namespace pm
