// pm_kernels.hip -- gfx950 kernels of the planetmapper hot path.
//
//   k_disc<FLAGS>   image-space planes that need the ray/ellipsoid intercept
//                   (lon/lat, centric, illumination, azimuth, LST, state, ring)
//   k_sky<LIMB>     image-space planes defined for every pixel
//                   (RA/Dec, pixel x/y, km, angular, limb)
//   k_map           map-space planes + x_map/y_map for a lon/lat grid
//   k_reproject<T>  bilinear / nearest reprojection of a cube onto the map grid
//   k_reproject_smooth<T>  'smooth' (PCHIP-oversampled) reprojection, evaluated on the fly
//
// Launch geometry of the image kernels: one lane per pixel, a wave covers 64
// consecutive x of one row, so every plane store is one 512-byte fully coalesced
// write per wave; blockIdx.y = row. No loads besides the kernel-argument block.
#include "pm_device.hip.h"

namespace pm {

constexpr int kBlock = 256;
// the spheroid image kernel runs one wave per workgroup: finer-grained dispatch mixes the
// cheap and the expensive row segments better (measured 0.269 vs 0.275 ms at 256, 0.291 at 512)
constexpr int kSphBlock = 64;

enum DiscFlags : int {
    DF_ILLUM = 1,  // PHASE / INCIDENCE / EMISSION / AZIMUTH
    DF_STATE = 2,  // DISTANCE / RADIAL-VELOCITY / DOPPLER
    DF_RING = 4,   // RING-RADIUS / RING-LON-GRAPHIC / RING-DISTANCE
};

#define PM_WANT(pl) ((p.mask >> (pl)) & 1ull)
#define PM_PUT(pl, val)                            \
    do {                                           \
        if (PM_WANT(pl)) p.out[pl][idx] = (val);   \
    } while (0)

// Store through a wave-uniform row pointer (SGPR pair) + 32-bit lane offset: the
// global_store saddr form, one address dword per lane instead of a 64-bit VGPR pointer.
#define PM_PUT_ROW(pl, val)                                               \
    do {                                                                  \
        if (PM_WANT(pl))                                                  \
            *reinterpret_cast<double *>(reinterpret_cast<char *>(p.out[pl] + row_base) + lane_off) = (val); \
    } while (0)

// Reference loops fused here: BodyXY._get_targvec_img body_xy.py:3195, _get_lonlat_img
// :3281, _get_lonlat_centric_img :3346, _get_illumination_gie_img :3658,
// get_azimuth_angle_img :3742, get_local_solar_time_img :3787, _get_state_imgs :3830,
// get_radial_velocity_img :3895, get_doppler_img :3938,
// _get_ring_plane_coordinate_imgs :4059.
template <int FLAGS>
__global__ __launch_bounds__(kBlock) void k_disc(const Params p)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = p.y_off + (int)blockIdx.y;
    const bool inside = x < p.nx;
    const size_t row_base = (size_t)blockIdx.y * p.nx;  // wave-uniform: stores use the saddr form
    const unsigned lane_off = (unsigned)x * 8u;
    const double nan = __builtin_nan("");

    // radius pre-mask of _get_targvec_img (only with optimize_speed)
    const double dx = (double)x - p.x0, dy = (double)y - p.y0;
    bool cand = inside && !(p.optimize_speed && (dx * dx + dy * dy) > p.r2);

    V3 ray = xy2ray(p, (double)x, (double)y);

    // rays rebuilt from the RA/Dec degree images for the ring planes
    // (_get_obsvec_norm_img body_xy.py:3262-3271)
    V3 ray2 = ray;
    if (FLAGS & DF_RING) {
        double ra, dec;
        recrad(ray, ra, dec);
        ray2 = radrec((ra * kDeg) * kRad, (dec * kDeg) * kRad);
    }

    V3 sp = {nan, nan, nan};
    double lt = 0.0;
    bool on_disc = false;
    // wave-uniform skip of the intercept when no lane of the wave can be on the disc
    if (__any(cand)) {
        if (cand) on_disc = sincpt(p, ray, sp, lt);
    }

    double lon_deg = nan, lat_deg = nan;
    if (on_disc) {
        double lon, lat;
        recpgr_surface(p, sp, lon, lat);
        lon_deg = lon * kDeg;
        lat_deg = lat * kDeg;
    }
    if (inside) {
        PM_PUT_ROW(PM_LON_GRAPHIC, lon_deg);
        PM_PUT_ROW(PM_LAT_GRAPHIC, lat_deg);
    }
    if (PM_WANT(PM_LON_CENTRIC) || PM_WANT(PM_LAT_CENTRIC)) {
        double lc = nan, bc = nan;
        if (on_disc) {
            // reclat_c: Body._targvec2lonlat_centric body.py:2905
            bc = atan2(sp.z, sqrt(fma(sp.x, sp.x, sp.y * sp.y))) * kDeg;
            lc = ((sp.x == 0.0 && sp.y == 0.0) ? 0.0 : atan2(sp.y, sp.x)) * kDeg;
        }
        if (inside) {
            PM_PUT_ROW(PM_LON_CENTRIC, lc);
            PM_PUT_ROW(PM_LAT_CENTRIC, bc);
        }
    }
    if (PM_WANT(PM_LOCAL_SOLAR_TIME)) {
        double v = local_solar_time(p, lon_deg);
        if (inside) PM_PUT_ROW(PM_LOCAL_SOLAR_TIME, v);
    }

    double surf_dist = nan;
    if (FLAGS & (DF_ILLUM | DF_STATE | DF_RING)) {
        double ph = nan, in = nan, em = nan, az = nan, rv = nan, dop = nan;
        if (on_disc) {
            // the intercept's light time is already the fixed point of the point's own
            // light-time equation to ~4e-10 s: one more pass converges it
            V3 pos;
            M3 R;
            point_lt<1>(p, sp, lt, pos, R);
            if (FLAGS & DF_ILLUM) {
                illum_angles(p, sp, lt, pos, R, ph, in, em);
                ph *= kDeg;
                in *= kDeg;
                em *= kDeg;
                if (PM_WANT(PM_AZIMUTH)) az = azimuth_deg(ph, in, em);
            }
            if (FLAGS & (DF_STATE | DF_RING)) surf_dist = lt * p.g.clight;
            if (FLAGS & DF_STATE) {
                rv = radial_velocity(p, sp, lt, pos, R);
                double beta = rv / p.g.clight;  // SpiceBase.calculate_doppler_factor base.py:550
                dop = sqrt((1.0 + beta) / (1.0 - beta));
            }
        }
        if (inside) {
            if (FLAGS & DF_ILLUM) {
                PM_PUT_ROW(PM_PHASE, ph);
                PM_PUT_ROW(PM_INCIDENCE, in);
                PM_PUT_ROW(PM_EMISSION, em);
                PM_PUT_ROW(PM_AZIMUTH, az);
            }
            if (FLAGS & DF_STATE) {
                PM_PUT_ROW(PM_DISTANCE, surf_dist);
                PM_PUT_ROW(PM_RADIAL_VELOCITY, rv);
                PM_PUT_ROW(PM_DOPPLER, dop);
            }
        }
    }

    if (FLAGS & DF_RING) {
        double rr, rl, rd;
        ring_coords(p, ray2, rr, rl, rd);
        // hidden behind the disc (NaN compares false): body_xy.py:4077-4080
        if (rd > surf_dist) rr = rl = rd = nan;
        if (inside) {
            PM_PUT_ROW(PM_RING_RADIUS, rr);
            PM_PUT_ROW(PM_RING_LON_GRAPHIC, rl);
            PM_PUT_ROW(PM_RING_DISTANCE, rd);
        }
    }
}

// ------------------------------------------------------------------ spheroid fast path
// Same planes as k_disc (without the ring planes) for bodies with radii[0] == radii[1].
//
// All vectors live in B0, the body-fixed frame frozen at t0 = et - lt_c. Because a
// spheroid is invariant under rotation about its spin axis, the ray/ellipsoid intercept
// at the light-time corrected epoch te can be evaluated in B0 with the un-rotated body:
// only the target's translation VB (te - t0) enters the light-time iteration, and the
// spin shows up once, as the longitude offset wdot (te - t0). No 3x3 products inside the
// loop, no separate light-time solve for the illumination (the point lies ON the ray, so
// its position relative to the observer is tau * ray), one sqrt per intercept.
// vsep angles use asin on |x| <= 0.5 only: 2 asin(|u-v|/2) for angles < 60 deg (and the
// supplement form > 120 deg) like CSPICE's vsep_c, pi/2 - asin(u.v) in between.
__device__ __forceinline__ double vsep_fast(V3 u, V3 v)
{
    const double d = dot(u, v);
    const double sg = (d > 0.0) ? -1.0 : 1.0;
    const V3 w = {fma(sg, v.x, u.x), fma(sg, v.y, u.y), fma(sg, v.z, u.z)};
    const double s = 0.5 * sqrt_fast(dot(w, w));
    const bool mid = fabs(d) < 0.5;
    const double r = asin_half(mid ? d : s);
    return mid ? kHalfPi - r : (d > 0.0 ? 2.0 * r : kPi - 2.0 * r);
}

// TRI: triaxial ellipsoid (a != b). The shape is no longer invariant under the spin, so each
// light-time evaluation first turns ray and observer by the spin angle of its epoch (a few
// 1e-5 rad: series) into the body-fixed frame and rescales the ray; the intercept is then
// body-fixed, and is turned back to B0 for the illumination geometry.
template <int FLAGS, bool TRI>
__global__ __launch_bounds__(kSphBlock) void k_disc_sph(const Params p)
{
    // Workgroups are dealt round-robin to the 8 XCDs (linear id % 8); with a row-major grid
    // each XCD would always get the same image columns, and the columns through the disc
    // centre cost far more than the ones at the frame edge. Rotating the column block by the
    // row index gives every XCD the same mix.
    const int x = (int)((blockIdx.x + blockIdx.y) % gridDim.x) * kSphBlock + threadIdx.x;
    // Rows are visited in a golden-ratio stride order (a bijection: gcd(row_stride, ny) = 1)
    // so that store-only rows off the disc and FP64-heavy rows through it are resident on the
    // chip at the same time: HBM writes of the former overlap the VALU work of the latter.
    const int yl = (int)(((long long)blockIdx.y * p.row_stride) % p.rows);  // row within this launch
    const int y = p.y_off + yl;
    const bool inside = x < p.nx;
    const size_t row_base = (size_t)yl * p.nx;  // wave-uniform
    const unsigned lane_off = (unsigned)x * 8u;  // byte offset in the row (< 4 GiB, checked by the host)
    const double nan = __builtin_nan("");

    const double dx = (double)x - p.x0, dy = (double)y - p.y0;
    const bool cand = inside && !(p.optimize_speed && (dx * dx + dy * dy) > p.r2);

    double lon_deg = nan, lat_deg = nan, lc_deg = nan, bc_deg = nan;
    double ph = nan, in = nan, em = nan, az = nan, dist = nan, rv = nan, dop = nan;
    double rr = nan, rl = nan, rd = nan;
    double dist_lt = nan;  // observer -> surface distance (lt * c) of on-disc pixels

    const bool any_cand = __any(cand);
    V3 va = {0.0, 0.0, 0.0};  // unit vector of the pixel in the angular frame
    if (any_cand || (FLAGS & DF_RING)) {
        // pixel -> unit ray (BodyXY._xy2obsvec_norm body_xy.py:375)
        const double fx = (double)x, fy = (double)y;
        const double ax = fma(p.A[0], fx, fma(p.A[1], fy, p.A[2]));
        const double ay = fma(p.A[3], fx, fma(p.A[4], fy, p.A[5]));
        double sr, cr, sd, cd;
        constexpr double kArcsec = kRad / 3600.0;  // arcsec -> rad (1 ulp from (a / 3600) * kRad)
        sincos_auto(-(ax * kArcsec), sr, cr);
        sincos_auto(ay * kArcsec, sd, cd);
        va = v3(cr * cd, sr * cd, sd);
    }

    if (any_cand) {
        const V3 u = mxv(p.C, va);  // ray in B0

        // surfpt_c in scaled coordinates; for a spheroid X and 1/(X.X) are fixed for the pixel
        V3 X = {u.x * p.ir[0], u.y * p.ir[1], u.z * p.ir[2]};
        double ixx = rcp_fast(dot(X, X));
        double cz = 1.0, sz = 0.0;  // spin since t0 at the epoch of the current evaluation (TRI)

        // sincpt_c 'CN': converged light time, CSPICE stopping rule, <= 10 evaluations
        // CSPICE's rule is |dlt| <= 1e-17 |et - lt|; lt varies by 1e-9 relative over a disc
        const double lt_tol = 1e-17 * fabs(p.t0);
        double lt = p.g.lt_c, d = 0.0, k = 0.0, root = 0.0;
        V3 P = {0.0, 0.0, 0.0};
        bool hit = cand;
#pragma unroll 1
        for (int it = 0; it < 10; it++) {
            // wave-uniform exit once no lane is still iterating
            const double te = p.g.et - lt;
            d = te - p.t0;
            // (the target's acceleration moves it by A d^2 / 2 < 1e-8 km over the |d| <= R / c of a
            //  disc intercept, 10x below the rounding of the ray itself: not carried here)
            const V3 obs = {fma(-p.VB[0], d, p.O0[0]), fma(-p.VB[1], d, p.O0[1]), fma(-p.VB[2], d, p.O0[2])};
            V3 Y;
            if (TRI) {
                const double dl = p.g.wdot * d, d2 = dl * dl;  // |dl| < 1e-3 (host check)
                cz = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
                sz = dl * fma(d2, -1.0 / 6.0, 1.0);
                const V3 ub = {fma(cz, u.x, sz * u.y), fma(cz, u.y, -sz * u.x), u.z};
                X = {ub.x * p.ir[0], ub.y * p.ir[1], ub.z * p.ir[2]};
                ixx = rcp_fast(dot(X, X));
                Y = {fma(cz, obs.x, sz * obs.y) * p.ir[0], fma(cz, obs.y, -sz * obs.x) * p.ir[1], obs.z * p.ir[2]};
            } else {
                Y = {obs.x * p.ir[0], obs.y * p.ir[1], obs.z * p.ir[2]};
            }
            const double yx = dot(Y, X);
            k = yx * ixx;
            P = {fma(-k, X.x, Y.x), fma(-k, X.y, Y.y), fma(-k, X.z, Y.z)};
            const double p2 = dot(P, P);
            // (an observer inside the body, Y.Y <= 1, never reaches this kernel: the
            //  launcher requires |O0| scaled > 1 and the target moves km, not radii)
            if (p2 > 1.0 || yx > 0.0) hit = false;
            root = sqrt_fast(fmax(0.0, 1.0 - p2) * ixx);
            const double nlt = (-k - root) * p.inv_c;
            const double err = fabs(nlt - lt);
            const bool done = !hit || err <= lt_tol;
            if (hit) lt = nlt;
            if (__all(done)) break;
        }

        if (hit) {
            // intercept in B0; body-fixed = Rz_frame(delta) * B0 with delta = wdot d
            const double tau = -k - root;  // distance observer -> point along the ray
            if (FLAGS & (DF_RING | DF_STATE)) dist_lt = lt * p.g.clight;
            const V3 Xf = {fma(-root, X.x, P.x), fma(-root, X.y, P.y), fma(-root, X.z, P.z)};
            // body-fixed at te for TRI, B0 otherwise (body-fixed = Rz_frame(delta) * B0)
            const V3 sp = {Xf.x * p.radii[0], Xf.y * p.radii[1], Xf.z * p.radii[2]};
            const double delta = TRI ? 0.0 : p.g.wdot * d;
            const double rho = sqrt_fast(fma(sp.x, sp.x, sp.y * sp.y));
            const bool polar = (sp.x == 0.0 && sp.y == 0.0);
            // recpgr_c body.py:1030: east longitude in the frame at te = B0 longitude - delta
            const double le = polar ? 0.0 : atan2_fast(sp.y, sp.x) - delta;
            double l = p.g.west_positive ? -le : le;
            if (l < 0.0) l += kTwoPi;
            if (l >= kTwoPi) l -= kTwoPi;
            lon_deg = l * kDeg;
            lat_deg = ((polar && sp.z == 0.0) ? kHalfPi : atan2_fast(sp.z * p.lat_k, rho)) * kDeg;
            if (PM_WANT(PM_LON_CENTRIC) || PM_WANT(PM_LAT_CENTRIC)) {
                // reclat_c body.py:2905: east-positive, (-pi, pi]
                double lc = le;
                if (lc <= -kPi) lc += kTwoPi;
                if (lc > kPi) lc -= kTwoPi;
                lc_deg = lc * kDeg;
                bc_deg = ((polar && sp.z == 0.0) ? 0.0 : atan2_fast(sp.z, rho)) * kDeg;
            }
            // the point in B0 (for the Sun / observer geometry, which lives there)
            const V3 sp0 = TRI ? v3(fma(cz, sp.x, -sz * sp.y), fma(sz, sp.x, cz * sp.y), sp.z) : sp;
            if (FLAGS & DF_ILLUM) {
                // illumf_c body.py:1915: point wrt P_T(t0) in B0; Sun light time: two passes
                const V3 q = {fma(p.VB[0], d, sp0.x), fma(p.VB[1], d, sp0.y), fma(p.VB[2], d, sp0.z)};
                const double te = p.g.et - lt;
                // Sun light time (spkcpo_c 'CN'): at lts0 = te - ts0 the Sun sits at S0 exactly;
                // one correction pass leaves |d lts| ~ (v_sun / c) * 0.25 s = 1e-8 s, i.e. 1e-10 km
                V3 sv = ld3(p.SB0) - q;
                const double s2 = dot(sv, sv);
                const double lts = s2 * rsqrt_fast(s2) * p.inv_c;
                const double ds = (te - lts) - p.g.ts0;
                sv = v3(fma(p.VSB[0], ds, p.SB0[0]) - q.x, fma(p.VSB[1], ds, p.SB0[1]) - q.y,
                        fma(p.VSB[2], ds, p.SB0[2]) - q.z);
                const V3 sunb = rsqrt_fast(dot(sv, sv)) * sv;
                const V3 ob = neg(u);  // observer seen from the point: -ray (unit)
                V3 n = {sp.x * (p.ir[0] * p.ir[0]), sp.y * (p.ir[1] * p.ir[1]), sp.z * (p.ir[2] * p.ir[2])};  // surfnm_c
                if (TRI) n = {fma(cz, n.x, -sz * n.y), fma(sz, n.x, cz * n.y), n.z};
                n = rsqrt_fast(dot(n, n)) * n;
                ph = vsep_fast(sunb, ob) * kDeg;
                in = vsep_fast(n, sunb) * kDeg;
                em = vsep_fast(n, ob) * kDeg;
                if (PM_WANT(PM_AZIMUTH)) az = azimuth_deg(ph, in, em);
            }
            if (FLAGS & DF_STATE) {
                // spkcpt_c body.py:2830: distance = lt c; velocity with the light-time rate
                dist = dist_lt;
                const V3 vp = {fma(p.AB[0], d, p.VB[0]) - p.g.wdot * sp0.y, fma(p.AB[1], d, p.VB[1]) + p.g.wdot * sp0.x,
                               fma(p.AB[2], d, p.VB[2])};
                const V3 vo = ld3(p.VOB);
                const double dlt = (dot(u, vp - vo) * p.inv_c) / (1.0 + dot(u, vp) * p.inv_c);
                rv = dot((1.0 - dlt) * vp - vo, u);
                const double beta = rv / p.g.clight;
                dop = sqrt((1.0 + beta) / (1.0 - beta));
                (void)tau;
            }
        }
    }

    if (FLAGS & DF_RING) {
        // Body._ring_coordinates_from_obsvec(only_visible=False) body.py:2577-2615 for EVERY
        // pixel: inrypl_c with the J2000 ray, PM's _obsvec2targvec (body.py:972-1006; it mixes
        // J2000 and body-fixed components by design) and recpgr_c of the in-plane point. (The
        // reference rebuilds the ray from RA/Dec in degrees, body_xy.py:3262; that round trip
        // perturbs it by < 1 ulp - below the rounding of the ray itself - and is not replayed.)
        const V3 ray = mtxv(p.g.M, va);
        const double pd = dot(ray, ld3(p.g.ring_n));
        const double kk = p.g.ring_k;
        const bool ok = (kk == 0.0) ? (pd != 0.0) : (pd > 0.0 && kk < pd * (1.7976931348623157e308 / 3.0));
        if (__any(ok)) {
            // lanes without an intersection carry a harmless finite point through the math
            const double s = !ok ? 1.0 : ((kk == 0.0) ? 0.0 : div_fast(kk, pd));
            const V3 ip = s * ray;
            const V3 off = ip - ld3(p.g.sub_obsvec);
            const V3 w = off - ld3(p.g.sub_ray);
            const double dd = sqrt_fast(dot(w, w)) - p.g.sub_dist;
            const double t = p.g.sub_et - dd * p.inv_c;
            double sa, ca;
            sincos_auto(p.g.wdot * (t - p.t0), sa, ca);
            const V3 ob = mxv(p.g.R0, off);  // R(t) off = Rz_frame(ang) (R0 off)
            const V3 tv = {fma(ca, ob.x, sa * ob.y) + p.g.sub_sp[0], fma(ca, ob.y, -sa * ob.x) + p.g.sub_sp[1],
                           ob.z + p.g.sub_sp[2]};
            double le, alt;
            recpgr_alt_lon(p, tv, le, alt);
            double l = p.g.west_positive ? -le : le;
            if (l < 0.0) l += kTwoPi;
            // hidden behind the disc (NaN compares false): body_xy.py:4077-4080
            const double rdist = sqrt_fast(dot(ip, ip));
            if (ok && !(rdist > dist_lt)) {
                rr = alt + p.radii[0];
                rl = l * kDeg;
                rd = rdist;
            }
        }
    }

    if (inside) {
        if (FLAGS & DF_RING) {
            PM_PUT_ROW(PM_RING_RADIUS, rr);
            PM_PUT_ROW(PM_RING_LON_GRAPHIC, rl);
            PM_PUT_ROW(PM_RING_DISTANCE, rd);
        }
        PM_PUT_ROW(PM_LON_GRAPHIC, lon_deg);
        PM_PUT_ROW(PM_LAT_GRAPHIC, lat_deg);
        PM_PUT_ROW(PM_LON_CENTRIC, lc_deg);
        PM_PUT_ROW(PM_LAT_CENTRIC, bc_deg);
        if (PM_WANT(PM_LOCAL_SOLAR_TIME)) PM_PUT_ROW(PM_LOCAL_SOLAR_TIME, local_solar_time(p, lon_deg));
        if (FLAGS & DF_ILLUM) {
            PM_PUT_ROW(PM_PHASE, ph);
            PM_PUT_ROW(PM_INCIDENCE, in);
            PM_PUT_ROW(PM_EMISSION, em);
            PM_PUT_ROW(PM_AZIMUTH, az);
        }
        if (FLAGS & DF_STATE) {
            PM_PUT_ROW(PM_DISTANCE, dist);
            PM_PUT_ROW(PM_RADIAL_VELOCITY, rv);
            PM_PUT_ROW(PM_DOPPLER, dop);
        }
    }
}

// Reference loops fused here: BodyXY._get_radec_img body_xy.py:3409, get_x_img :3494,
// get_y_img :3519, _get_km_xy_img :3545, get_angular_x_img :3610,
// _get_limb_coordinate_imgs :3964.
template <bool LIMB>
__global__ __launch_bounds__(kBlock) void k_sky(const Params p)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = p.y_off + (int)blockIdx.y;
    if (x >= p.nx) return;
    const size_t row_base = (size_t)blockIdx.y * p.nx;  // wave-uniform: stores use the saddr form
    const unsigned lane_off = (unsigned)x * 8u;
    // xy2ray with the fast elementary functions (rays are finite for every pixel)
    const double xd = (double)x, yd = (double)y;
    const double ax0 = fma(p.A[0], xd, fma(p.A[1], yd, p.A[2]));
    const double ay0 = fma(p.A[3], xd, fma(p.A[4], yd, p.A[5]));
    V3 ray = mtxv(p.g.M, radrec_f(-(div_fast(ax0, 3600.0) * kRad), div_fast(ay0, 3600.0) * kRad));
    if (PM_WANT(PM_RA) || PM_WANT(PM_DEC)) {
        double ra, dec;
        recrad_f(ray, ra, dec);
        PM_PUT_ROW(PM_RA, ra * kDeg);
        PM_PUT_ROW(PM_DEC, dec * kDeg);
    }
    PM_PUT_ROW(PM_PIXEL_X, xd);
    PM_PUT_ROW(PM_PIXEL_Y, yd);
    const bool km = PM_WANT(PM_KM_X) || PM_WANT(PM_KM_Y) || PM_WANT(PM_ANGULAR_X) || PM_WANT(PM_ANGULAR_Y);
    if (km || LIMB) {
        // The reference rebuilds the ray from RA/Dec in degrees (radec2obsvec_norm, body_xy.py:3262);
        // that round trip moves it by < 1 ulp - below the rounding of the ray itself, 1e4 times
        // below the parity bar after the D/R amplification - and is not replayed.
        const V3 ray2 = ray;
        if (km) {
            double ax, ay;
            obsvec2angular_f(p, ray2, ax, ay);
            double kx = fma(p.K[0], ax, p.K[1] * ay), ky = fma(p.K[2], ax, p.K[3] * ay);
            PM_PUT_ROW(PM_KM_X, kx);
            PM_PUT_ROW(PM_KM_Y, ky);
            if (PM_WANT(PM_ANGULAR_X) || PM_WANT(PM_ANGULAR_Y)) {
                const double ik = rcp_fast(p.g.km_per_arcsec);
                const double qx = kx * ik, qy = ky * ik;
                PM_PUT_ROW(PM_ANGULAR_X, fma(fma(-p.g.km_per_arcsec, qx, kx), ik, qx));
                PM_PUT_ROW(PM_ANGULAR_Y, fma(fma(-p.g.km_per_arcsec, qy, ky), ik, qy));
            }
        }
        if (LIMB) {
            double ll, lb, ld;
            limb_coords_f(p, ray2, ll, lb, ld);
            PM_PUT_ROW(PM_LIMB_LON_GRAPHIC, ll);
            PM_PUT_ROW(PM_LIMB_LAT_GRAPHIC, lb);
            PM_PUT_ROW(PM_LIMB_DISTANCE, ld);
        }
    }
}

// Map-space chain: BodyXY._get_targvec_map body_xy.py:3227, _get_illumf_map :3667,
// _get_obsvec_map :3273, _get_radec_map :3419, _get_xy_map :3478 and the get_*_map
// planes. One lane per map location; lon/lat grids are read coalesced.
// SUN: phase / incidence / azimuth or the `lit` gate of the limb / ring planes are wanted;
// STATE: distance / radial velocity / doppler. The x/y-map request of a reprojection needs
// neither: only the emission angle (visibility) is evaluated then.
template <bool SUN, bool STATE>
__global__ __launch_bounds__(kBlock) void k_map(const Params p, const double *__restrict__ lon_in,
                                                const double *__restrict__ lat_in)
{
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t n = (size_t)p.n0 * p.n1;
    if (idx >= n) return;
    const double nan = __builtin_nan("");
    double lon_deg = lon_in[idx], lat_deg = lat_in[idx];
    if (!isfinite(lon_deg) || !isfinite(lat_deg)) lon_deg = lat_deg = nan;
    PM_PUT(PM_LON_GRAPHIC, lon_deg);
    PM_PUT(PM_LAT_GRAPHIC, lat_deg);
    const bool have = !isnan(lon_deg);
    V3 tv = {nan, nan, nan};
    if (have) tv = pgrrec_surface(p, lon_deg * kRad, lat_deg * kRad);

    double ph = nan, in = nan, em = nan, surf_dist = nan, rv = nan, dop = nan;
    bool vis = false, lit = false;
    if (have) {
        double lt = p.g.lt_c;
        V3 pos;
        M3 R;
        point_lt<3>(p, tv, lt, pos, R);
        if (SUN) {
            illum_angles(p, tv, lt, pos, R, ph, in, em);
            lit = in < kHalfPi;
        } else {
            em = emission_angle(p, tv, pos, R);
        }
        vis = em < kHalfPi;
        ph *= kDeg;
        in *= kDeg;
        em *= kDeg;
        surf_dist = lt * p.g.clight;
        if (STATE) {
            rv = radial_velocity(p, tv, lt, pos, R);
            double beta = rv / p.g.clight;
            dop = sqrt((1.0 + beta) / (1.0 - beta));
        }
    }
    PM_PUT(PM_PHASE, ph);
    PM_PUT(PM_INCIDENCE, in);
    PM_PUT(PM_EMISSION, em);
    if (PM_WANT(PM_AZIMUTH)) PM_PUT(PM_AZIMUTH, have ? azimuth_deg(ph, in, em) : nan);
    PM_PUT(PM_DISTANCE, surf_dist);
    PM_PUT(PM_RADIAL_VELOCITY, rv);
    PM_PUT(PM_DOPPLER, dop);
    if (PM_WANT(PM_LON_CENTRIC) || PM_WANT(PM_LAT_CENTRIC)) {
        double lc = nan, bc = nan;
        if (have) {
            bc = atan2(tv.z, sqrt(fma(tv.x, tv.x, tv.y * tv.y))) * kDeg;
            lc = ((tv.x == 0.0 && tv.y == 0.0) ? 0.0 : atan2(tv.y, tv.x)) * kDeg;
        }
        PM_PUT(PM_LON_CENTRIC, lc);
        PM_PUT(PM_LAT_CENTRIC, bc);
    }
    if (PM_WANT(PM_LOCAL_SOLAR_TIME)) PM_PUT(PM_LOCAL_SOLAR_TIME, local_solar_time(p, lon_deg));

    V3 ov = {nan, nan, nan};
    if (have) ov = targvec2obsvec(p, tv);
    double ra_deg = nan, dec_deg = nan;
    if (have && vis) {
        double ra, dec;
        recrad_f(ov, ra, dec);  // (ov is finite and non-zero here)
        ra_deg = ra * kDeg;
        dec_deg = dec * kDeg;
    }
    PM_PUT(PM_RA, ra_deg);
    PM_PUT(PM_DEC, dec_deg);

    double px = nan, py = nan, kx = nan, ky = nan;
    if (!isnan(ra_deg)) {
        V3 u = radrec_f(ra_deg * kRad, dec_deg * kRad);
        double ax, ay;
        obsvec2angular_f(p, u, ax, ay);
        double xx = fma(p.Ai[0], ax, fma(p.Ai[1], ay, p.Ai[2]));
        double yy = fma(p.Ai[3], ax, fma(p.Ai[4], ay, p.Ai[5]));
        // BodyXY._xy_in_image_frame body_xy.py:1868
        if (-0.5 < xx && xx < p.nx - 0.5 && -0.5 < yy && yy < p.ny - 0.5) {
            px = xx;
            py = yy;
        }
        kx = fma(p.K[0], ax, p.K[1] * ay);
        ky = fma(p.K[2], ax, p.K[3] * ay);
    }
    PM_PUT(PM_PIXEL_X, px);
    PM_PUT(PM_PIXEL_Y, py);
    PM_PUT(PM_KM_X, kx);
    PM_PUT(PM_KM_Y, ky);
    PM_PUT(PM_ANGULAR_X, kx / p.g.km_per_arcsec);
    PM_PUT(PM_ANGULAR_Y, ky / p.g.km_per_arcsec);

    const bool need_limb =
        PM_WANT(PM_LIMB_DISTANCE) || PM_WANT(PM_LIMB_LON_GRAPHIC) || PM_WANT(PM_LIMB_LAT_GRAPHIC);
    const bool need_ring =
        PM_WANT(PM_RING_RADIUS) || PM_WANT(PM_RING_LON_GRAPHIC) || PM_WANT(PM_RING_DISTANCE);
    if (need_limb || need_ring) {
        // gated on illumf column 4 (lit) like the reference (body_xy.py:3981, 4097)
        double ll = nan, lb = nan, ld = nan, rr = nan, rl = nan, rd = nan;
        if (have && lit) {
            if (need_limb) limb_coords(p, ov, ll, lb, ld);
            if (need_ring) ring_coords(p, ov, rr, rl, rd);
        }
        if (rd > surf_dist) rr = rl = rd = nan;
        PM_PUT(PM_LIMB_LON_GRAPHIC, ll);
        PM_PUT(PM_LIMB_LAT_GRAPHIC, lb);
        PM_PUT(PM_LIMB_DISTANCE, ld);
        PM_PUT(PM_RING_RADIUS, rr);
        PM_PUT(PM_RING_LON_GRAPHIC, rl);
        PM_PUT(PM_RING_DISTANCE, rd);
    }
}

// ------------------------------------------------------------------ coordinate transforms
// Array-valued coordinate transforms (reference: SpiceBase._maybe_transform_as_arrays
// base.py:719-757 around BodyXY.xy2radec ... angular2xy body_xy.py:385-561 and
// Body.lonlat2radec ... angular2km body.py:1083-1217, 1375-1800). One lane per point.

// Body._test_if_targvec_visible body.py:2112-2150 (p.radii must be the unadjusted radii)
__device__ __forceinline__ bool targvec_visible(const Params &p, V3 tv, bool on_surface)
{
    if (on_surface) {
        double lt = p.g.lt_c, ph, in, em;
        V3 pos;
        M3 R;
        point_lt<3>(p, tv, lt, pos, R);
        illum_angles(p, tv, lt, pos, R, ph, in, em);
        return em < kHalfPi;
    }
    V3 ov = targvec2obsvec(p, tv), sp;
    double lt_i;
    if (!sincpt(p, ov, sp, lt_i)) return true;
    V3 pos;
    M3 R;
    point_lt<1>(p, sp, lt_i, pos, R);
    double lt_p = p.g.lt_c;
    point_lt<3>(p, tv, lt_p, pos, R);
    return lt_p < lt_i;
}

__global__ __launch_bounds__(kBlock) void k_transform(const Params p, const TransformArgs t)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= t.n) return;
    const double nan = __builtin_nan("");
    enum { CS_XY = 0, CS_RADEC = 1, CS_ANGULAR = 2, CS_KM = 3, CS_LONLAT = 4 };
    const bool nvn = t.flags & 1, centric = t.flags & 2;
    // a copy of the block whose `radii` are the unadjusted ones (source side of lon/lat)
    double pa = t.a[i], pb = t.b[i], ra_out = nan, rb_out = nan;
    V3 ov = {nan, nan, nan};
    if (t.from == CS_LONLAT) {
        Params p0 = p;
        p0.radii[0] = t.radii0[0];
        p0.radii[1] = t.radii0[1];
        p0.radii[2] = t.radii0[2];
        double lon = pa, lat = pb;
        if (centric) {
            // centric2graphic_lonlat body.py:2970: latsrf_c + targvec2lonlat(alt)
            if (!(isfinite(lon) && isfinite(lat))) {
                lon = lat = nan;
            } else {
                V3 dir = radrec(lon * kRad, lat * kRad), s;
                surfpt(v3(0.0, 0.0, 0.0), dir, p0.radii, s);
                double lo, la, al;
                if (t.alt == 0.0)
                    recpgr_surface(p, s, lo, la);
                else
                    recpgr_general(p, s, lo, la, al);  // p.radii = radii + alt
                lon = lo * kDeg;
                lat = la * kDeg;
            }
        }
        const double lr = lon * kRad, br = lat * kRad;
        if (isfinite(lr) && isfinite(br) && isfinite(t.alt)) {
            V3 tv = pgrrec_alt(p0, p0.radii, lr, br, t.alt);
            if (!nvn || targvec_visible(p0, tv, t.alt == 0.0)) ov = targvec2obsvec(p0, tv);
        }
    } else if (t.from == CS_RADEC) {
        const double ra = pa * kRad, dec = pb * kRad;
        if (isfinite(ra) && isfinite(dec)) ov = radrec(ra, dec);
    } else {
        double ax = pa, ay = pb;
        if (t.from == CS_XY) {
            ax = p.A[0] * pa + p.A[1] * pb + p.A[2];
            ay = p.A[3] * pa + p.A[4] * pb + p.A[5];
        } else if (t.from == CS_KM) {
            ax = t.Kf[0] * pa + t.Kf[1] * pb;
            ay = t.Kf[2] * pa + t.Kf[3] * pb;
        }
        V3 v = radrec(-((ax / 3600.0) * kRad), (ay / 3600.0) * kRad);
        ov = mtxv(p.g.M, v);
    }
    const bool have = finite3(ov);
    if (t.to == CS_RADEC) {
        if (have) {
            double ra, dec;
            recrad(ov, ra, dec);
            ra_out = ra * kDeg;
            rb_out = dec * kDeg;
        }
    } else if (t.to == CS_LONLAT) {
        V3 sp;
        double lt;
        if (have && sincpt(p, ov, sp, lt)) {  // p.radii = radii + alt
            double lo, la;
            recpgr_surface(p, sp, lo, la);
            double lon = lo * kDeg, lat = la * kDeg;
            if (centric) {
                // graphic2centric_lonlat(lon, lat, alt=alt) inside the altitude context
                V3 tv = pgrrec_alt(p, p.radii, lon * kRad, lat * kRad, t.alt);
                lat = atan2(tv.z, sqrt(fma(tv.x, tv.x, tv.y * tv.y))) * kDeg;
                lon = ((tv.x == 0.0 && tv.y == 0.0) ? 0.0 : atan2(tv.y, tv.x)) * kDeg;
            }
            ra_out = lon;
            rb_out = lat;
        }
    } else if (have) {
        double ax, ay;
        obsvec2angular(p, ov, ax, ay);
        if (t.to == CS_ANGULAR) {
            ra_out = ax;
            rb_out = ay;
        } else if (t.to == CS_KM) {
            ra_out = p.K[0] * ax + p.K[1] * ay;
            rb_out = p.K[2] * ax + p.K[3] * ay;
        } else {
            ra_out = p.Ai[0] * ax + p.Ai[1] * ay + p.Ai[2];
            rb_out = p.Ai[3] * ax + p.Ai[4] * ay + p.Ai[5];
        }
    }
    t.oa[i] = ra_out;
    t.ob[i] = rb_out;
}

// ------------------------------------------------------------------ reprojection
template <typename T>
__device__ __forceinline__ double load_as_f64(const T *p, size_t i)
{
    return (double)p[i];
}

// Value the reference would interpolate from at pixel (i, j): the pixel itself if finite,
// else the mean of the finite pixels of its clipped 3x3 window, else the plane's nanmedian
// (BodyXY._replace_nans_with_interpolated_values body_xy.py:1871-1904; the reflect-mode
// `uniform_filter(bad, size=3)` test there is equivalent to "the clipped window holds no
// finite pixel" because reflection only repeats pixels of the window).
template <typename T>
__device__ __forceinline__ double cleaned_at(const T *img, long i, long j, int ny, int nx, double median,
                                             bool &needs_median)
{
    const double v = load_as_f64(img, (size_t)i * nx + j);
    if (isfinite(v)) return v;
    double sum = 0.0;
    int cnt = 0;
    for (long ii = (i > 0 ? i - 1 : 0); ii <= i + 1 && ii < ny; ii++)
        for (long jj = (j > 0 ? j - 1 : 0); jj <= j + 1 && jj < nx; jj++) {
            const double w = load_as_f64(img, (size_t)ii * nx + jj);
            if (isfinite(w)) {
                sum += w;
                cnt++;
            }
        }
    if (cnt > 0) return sum / (double)cnt;
    needs_median = true;
    return median;
}

// One lane per (map location, plane): BodyXY.map_img body_xy.py:1414 for every plane of
// Observation._get_mapped_data observation.py:876. blockIdx.y = plane, so the 64 lanes
// of a wave gather from one plane around neighbouring (x, y) -> the 4-point footprints
// overlap in L2; the store is coalesced along the map row.
//
// The reference interpolates a NaN-cleaned copy of each plane; here the cleaned value of a
// non-finite corner is computed on the fly from its 3x3 window. Only a corner whose whole
// window is non-finite needs the plane's nanmedian: with plane_stats == NULL (first pass)
// such a plane is flagged (plane_flags[pl] = call sequence number) and the host reruns it
// after k_median_*.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject(const ReprojectArgs a)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const T *img = (const T *)a.cube + (size_t)pl * ny * nx;
    double *o = a.out + (size_t)pl * a.n_map;
    double x = a.x_map[m], y = a.y_map[m];
    double val = nan;
    if (!isnan(x)) {
        if (a.interpolation == PM_INTERP_NEAREST) {
            // _do_nearest_interpolation body_xy.py:1633: np.round = half to even
            long xi = (long)rint(x), yi = (long)rint(y);
            if (xi < 0) xi += nx;
            if (yi < 0) yi += ny;
            // maps made by pm_xy_map are always inside the frame; a caller-supplied map that is
            // not (the reference raises IndexError there) must not read outside the plane
            if (xi >= 0 && xi < nx && yi >= 0 && yi < ny && !isnan(y)) val = load_as_f64(img, (size_t)yi * nx + xi);
        } else {
            const bool have_stats = a.plane_stats != nullptr;
            bool skip = have_stats && a.plane_stats[pl].all_nan;  // body_xy.py:1668-1670
            if (a.propagate_nan && !skip) {
                // _should_propagate_nan_to_map body_xy.py:1855-1866
                if (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1) {
                    skip = true;
                } else {
                    long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
                    long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
                    double t0 = load_as_f64(img, (size_t)ja * nx + ia), t1 = load_as_f64(img, (size_t)ja * nx + ib);
                    double t2 = load_as_f64(img, (size_t)jb * nx + ia), t3 = load_as_f64(img, (size_t)jb * nx + ib);
                    skip = isnan(t0) || isnan(t1) || isnan(t2) || isnan(t3);
                }
            }
            if (!skip) {
                // RectBivariateSpline(kx=ky=1, s=0).ev == bilinear; FITPACK clamps the
                // evaluation point to the knot range.
                double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
                long x0 = (long)floor(xc), y0 = (long)floor(yc);
                if (x0 > nx - 2) x0 = nx - 2;
                if (y0 > ny - 2) y0 = ny - 2;
                if (x0 < 0) x0 = 0;
                if (y0 < 0) y0 = 0;
                long x1 = x0 + 1 < nx ? x0 + 1 : x0, y1 = y0 + 1 < ny ? y0 + 1 : y0;
                double fx = xc - (double)x0, fy = yc - (double)y0;
                const double med = have_stats ? a.plane_stats[pl].median : 0.0;
                bool nm = false;
                // corners with zero weight contribute exactly 0 in the reference (their cleaned
                // value is finite), so they are not evaluated at all
                const double w00 = (1.0 - fy) * (1.0 - fx), w01 = (1.0 - fy) * fx, w10 = fy * (1.0 - fx), w11 = fy * fx;
                const double v00 = (fx != 1.0 && fy != 1.0) ? cleaned_at(img, y0, x0, ny, nx, med, nm) : 0.0;
                const double v01 = (fx != 0.0 && fy != 1.0) ? cleaned_at(img, y0, x1, ny, nx, med, nm) : 0.0;
                const double v10 = (fx != 1.0 && fy != 0.0) ? cleaned_at(img, y1, x0, ny, nx, med, nm) : 0.0;
                const double v11 = (fx != 0.0 && fy != 0.0) ? cleaned_at(img, y1, x1, ny, nx, med, nm) : 0.0;
                (void)w00; (void)w01; (void)w10; (void)w11;
                val = (1.0 - fy) * ((1.0 - fx) * v00 + fx * v01) + fy * ((1.0 - fx) * v10 + fx * v11);
                if (nm && !have_stats) atomicMax(&a.plane_flags[pl], a.seq);
            }
        }
    }
    o[m] = val;
}

// ------------------------------------------------------------------ spline reprojection
// scipy RectBivariateSpline(kx, ky, s=0).ev of BodyXY._do_spline_interpolation
// (body_xy.py:1651-1702) for 'quadratic', 'cubic' and (k0, k1): interpolating tensor-product
// B-spline. Pipeline per chunk of planes: k_median_* (plane statistics) -> k_spline_clean
// (NaN-cleaned float64 copy) -> k_spline_solve axis 0, axis 1 (banded LU substitution, in
// place: samples -> coefficients) -> k_spline_eval.

// 'smooth' interpolation (BodyXY._do_smooth_interpolation / _pchip_grid_interp2d
// body_xy.py:1704-1853). The reference materialises the whole oversampled image (up to
// 10000 x 10000 per plane) and then samples it bilinearly at the map cells. PCHIP is local
// (a piece depends on four samples), so here each (cell, plane) lane evaluates just the four
// fine-grid nodes around its sample: node (r, k) = column PCHIP at ys[r] over the rows whose
// row PCHIP at xs[k] is finite, each of those a PCHIP over the finite pixels of the row.
// Work scales with the map, not with the oversampled image, and nothing is staged in HBM.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject_smooth(const ReprojectArgs a, const SmoothArgs sm)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const T *img = (const T *)a.cube + (size_t)pl * ny * nx;
    const double x = a.x_map[m], y = a.y_map[m];
    double val = nan;
    bool skip = isnan(x);
    if (!skip && a.propagate_nan) {
        if (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1) {
            skip = true;
        } else {
            long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
            long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
            skip = isnan(load_as_f64(img, (size_t)ja * nx + ia)) || isnan(load_as_f64(img, (size_t)ja * nx + ib)) ||
                   isnan(load_as_f64(img, (size_t)jb * nx + ia)) || isnan(load_as_f64(img, (size_t)jb * nx + ib));
        }
    }
    // RegularGridInterpolator(bounds_error=False, fill_value=nan)
    const double x_lo = (double)sm.x.first, x_hi = (double)sm.x.last;
    const double y_lo = (double)sm.y.first, y_hi = (double)sm.y.last;
    if (!skip && x >= x_lo && x <= x_hi && y >= y_lo && y <= y_hi) {
        const int k = smooth_interval(sm.x, x), r = smooth_interval(sm.y, y);
        const double xk0 = smooth_grid(sm.x, k), xk1 = smooth_grid(sm.x, k + 1);
        const double yr0 = smooth_grid(sm.y, r), yr1 = smooth_grid(sm.y, r + 1);
        auto node = [&](double xq, double yq) {
            auto column = [&](int i) {
                auto row = [&](int j) { return load_as_f64(img, (size_t)i * nx + j); };
                return pchip_gappy(row, sm.x.first, sm.x.last, xq);
            };
            return pchip_gappy(column, sm.y.first, sm.y.last, yq);
        };
        const double f00 = node(xk0, yr0), f01 = node(xk1, yr0), f10 = node(xk0, yr1), f11 = node(xk1, yr1);
        const double fx = (x - xk0) / (xk1 - xk0), fy = (y - yr0) / (yr1 - yr0);
        val = f00 * (1.0 - fy) * (1.0 - fx) + f01 * (1.0 - fy) * fx + f10 * fy * (1.0 - fx) + f11 * fy * fx;
    }
    a.out[(size_t)pl * a.n_map + m] = val;
}

// nanmin / nanmax of the x and y maps (one block): limits[0..3] = xmin, xmax, ymin, ymax;
// +inf / -inf when no cell is visible.
__global__ __launch_bounds__(kBlock) void k_map_limits(const double *x_map, const double *y_map, int n, double *limits)
{
    __shared__ double sh[4][kBlock];
    double xmin = __builtin_inf(), xmax = -__builtin_inf(), ymin = __builtin_inf(), ymax = -__builtin_inf();
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const double x = x_map[i], y = y_map[i];
        if (!isnan(x)) {
            xmin = fmin(xmin, x);
            xmax = fmax(xmax, x);
        }
        if (!isnan(y)) {
            ymin = fmin(ymin, y);
            ymax = fmax(ymax, y);
        }
    }
    sh[0][threadIdx.x] = xmin;
    sh[1][threadIdx.x] = xmax;
    sh[2][threadIdx.x] = ymin;
    sh[3][threadIdx.x] = ymax;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sh[0][threadIdx.x] = fmin(sh[0][threadIdx.x], sh[0][threadIdx.x + st]);
            sh[1][threadIdx.x] = fmax(sh[1][threadIdx.x], sh[1][threadIdx.x + st]);
            sh[2][threadIdx.x] = fmin(sh[2][threadIdx.x], sh[2][threadIdx.x + st]);
            sh[3][threadIdx.x] = fmax(sh[3][threadIdx.x], sh[3][threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) limits[threadIdx.x] = sh[threadIdx.x][0];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_spline_clean(const T *cube, double *work, const PlaneStats *stats, int ny, int nx)
{
    const size_t npx = (size_t)ny * nx;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (i >= npx) return;
    bool nm = false;
    work[(size_t)pl * npx + i] = cleaned_at(cube + (size_t)pl * npx, (long)(i / nx), (long)(i % nx), ny, nx, stats[pl].median, nm);
}

// One lane per (plane, line): solve B c = v along `axis` in place with the banded LU.
// axis 0: lines are image columns (lanes adjacent in x read one image row per step: coalesced);
// axis 1: lines are image rows (each lane walks its own row).
__global__ __launch_bounds__(kBlock) void k_spline_solve(double *work, int n_planes, int ny, int nx, int axis, SplineAxis ax)
{
    const size_t npx = (size_t)ny * nx;
    const int lines = axis == 0 ? nx : ny;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (tid >= (size_t)n_planes * lines) return;
    const int pl = (int)(tid / lines), line = (int)(tid % lines);
    double *v = work + (size_t)pl * npx + (axis == 0 ? (size_t)line : (size_t)line * nx);
    const size_t stride = axis == 0 ? (size_t)nx : 1;
    const int n = ax.n, k = ax.k, w = 2 * k + 1;
    // forward substitution (unit lower triangle), the last k results kept in registers. Samples
    // and LU rows are fetched eight at a time so that their load latencies overlap (a line along
    // image rows then also uses every 64-byte sector it touches in full).
    double prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < n; i0 += 8) {
        double vb[8], lb[8][5];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = (i0 + u < n) ? i0 + u : n - 1;
            vb[u] = v[(size_t)i * stride];
            const double *row = ax.lu + (size_t)i * w;
#pragma unroll
            for (int q = 1; q <= 5; q++) lb[u][q - 1] = (q <= k) ? row[k - q] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u;
            if (i < n) {
                double s = vb[u];
#pragma unroll
                for (int q = 1; q <= 5; q++)
                    if (q <= k && i - q >= 0) s -= lb[u][q - 1] * prev[q - 1];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[(size_t)i * stride] = s;
            }
        }
    }
    // back substitution
#pragma unroll
    for (int q = 0; q < 5; q++) prev[q] = 0.0;
    for (int i0 = n - 1; i0 >= 0; i0 -= 8) {
        double vb[8], ub[8][6];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = (i0 - u >= 0) ? i0 - u : 0;
            vb[u] = v[(size_t)i * stride];
            const double *row = ax.lu + (size_t)i * w;
#pragma unroll
            for (int q = 0; q <= 5; q++) ub[u][q] = (q <= k) ? row[k + q] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 - u;
            if (i >= 0) {
                double s = vb[u];
#pragma unroll
                for (int q = 1; q <= 5; q++)
                    if (q <= k && i + q < n) s -= ub[u][q] * prev[q - 1];
                s /= ub[u][0];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[(size_t)i * stride] = s;
            }
        }
    }
}
__device__ __forceinline__ int spline_interval(const SplineAxis &ax, double x)
{
    // knots are samples (odd k) or sample midpoints (even k): the span follows from floor(x)
    int l = ax.k;
    const int hi = ax.n - 1;
    // t[k+1+j] = j + k/2 + 1 (odd k) or j + k/2 + 0.5 (even k); find the largest l with t[l] <= x
    const double off = (ax.k & 1) ? (double)(ax.k / 2 + 1) : (double)(ax.k / 2) + 0.5;
    int j = (int)floor(x - off) + 1;  // number of interior knots <= x
    if (j < 0) j = 0;
    l = ax.k + j;
    if (l > hi) l = hi;
    while (l < hi && x >= ax.t[l + 1]) l++;  // guard against rounding at knot values
    while (l > ax.k && x < ax.t[l]) l--;
    return l;
}
__device__ __forceinline__ void spline_basis(const SplineAxis &ax, double x, int l, double *h)
{
    double hh[6];
    h[0] = 1.0;
    for (int j = 1; j <= ax.k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            const int li = l + i, lj = li - j;
            const double f = hh[i - 1] / (ax.t[li] - ax.t[lj]);
            h[i - 1] += f * (ax.t[li] - x);
            h[i] = f * (x - ax.t[lj]);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_spline_eval(const ReprojectArgs a, const SplineArgs sa)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const T *img = (const T *)a.cube + (size_t)pl * ny * nx;
    const double *c = sa.work + (size_t)pl * ny * nx;
    double x = a.x_map[m], y = a.y_map[m];
    double val = nan;
    bool skip = isnan(x) || a.plane_stats[pl].all_nan;
    if (!skip && a.propagate_nan) {
        if (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1) {
            skip = true;
        } else {
            long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
            long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
            skip = isnan(load_as_f64(img, (size_t)ja * nx + ia)) || isnan(load_as_f64(img, (size_t)ja * nx + ib)) ||
                   isnan(load_as_f64(img, (size_t)jb * nx + ia)) || isnan(load_as_f64(img, (size_t)jb * nx + ib));
        }
    }
    if (!skip) {
        const double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
        double hy[6], hx[6];
        const int ly = spline_interval(sa.rows, yc), lx = spline_interval(sa.cols, xc);
        spline_basis(sa.rows, yc, ly, hy);
        spline_basis(sa.cols, xc, lx, hx);
        double s = 0.0;
        for (int p = 0; p <= sa.rows.k; p++) {
            double r = 0.0;
            for (int q = 0; q <= sa.cols.k; q++) r += hx[q] * c[(size_t)(ly - sa.rows.k + p) * nx + (lx - sa.cols.k + q)];
            s += hy[p] * r;
        }
        val = s;
    }
    a.out[(size_t)pl * a.n_map + m] = val;
}

// ------------------------------------------------------------------ per-plane nanmedian
// np.nanmedian of each plane (+-inf treated as NaN, body_xy.py:1882-1890) by an 8-pass
// radix select over the order-preserving 64-bit key of the doubles. Two ranks are tracked
// at once (the two middle elements of an even count). All planes are processed by the
// same launches; one pass = one streaming read of the cube.
__device__ __forceinline__ unsigned long long sortable_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// ------------------------------------------------------------------ smoothing splines
// Least squares  [A; B/p] c = [d; 0]  for every right-hand side q of one direction (one lane
// each) by the corrected semi-normal equations: R'R c = A'd with the host's QR factor R, then
// one refinement step with the residual (restores the accuracy the normal equations lose:
// error ~ cond(A) eps instead of cond(A)^2 eps). d(i, q) = in[i * si + q * sq]; g and c are
// nc x nrhs work / result arrays (coalesced over q); the substitutions keep their band of
// previous values in registers.
__global__ __launch_bounds__(kBlock) void k_sm_solve(const SmoothFitAxis a, const double *__restrict__ in, size_t si,
                                                     size_t sq, int nrhs, double *__restrict__ g_glob,
                                                     double *__restrict__ c_glob, int use_lds)
{
    extern __shared__ double sm_lds[];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrhs) return;
    const int band = a.k + 2, nc = a.nc;
    // work vectors g, c of this right-hand side: element j at v[j * st + o]. With few knots (the
    // usual outcome of smoothing) both live in LDS, lane-contiguous and conflict-free: the
    // substitutions are chains of dependent read-modify-writes, i.e. latency-bound in HBM.
    double *g = use_lds ? sm_lds : g_glob;
    double *c = use_lds ? sm_lds + (size_t)nc * blockDim.x : c_glob;
    const size_t st = use_lds ? (size_t)blockDim.x : (size_t)nrhs;
    const size_t o = use_lds ? (size_t)threadIdx.x : (size_t)q;
    for (int pass = 0; pass < 2; pass++) {
        // g = A' r, r = d (first pass) or d - A c (refinement). Consecutive samples share their
        // k + 1 B-splines or move on by one: the partial sums (and the coefficients they need)
        // sit in a register window that slides with the knot interval, and the samples are
        // fetched eight at a time so that their load latencies overlap.
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, c4 = 0.0, c5 = 0.0;
        int cur = 0;
        if (pass) {
            c0 = c[0 * st + o];
            c1 = c[1 * st + o];
            if (a.k >= 2) c2 = c[2 * st + o];
            if (a.k >= 3) c3 = c[3 * st + o];
            if (a.k >= 4) c4 = c[4 * st + o];
            if (a.k >= 5) c5 = c[5 * st + o];
        }
        for (int i0 = 0; i0 < a.m; i0 += 8) {
            double buf[8], hv[8][6];
            int lv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {  // all loads of the batch are issued before any is used
                const int i = (i0 + u < a.m) ? i0 + u : a.m - 1;
                buf[u] = in[(size_t)i * si + q * sq];
                lv[u] = a.lb[i];
#pragma unroll
                for (int e = 0; e < 6; e++) hv[u][e] = a.hb[(size_t)i * 6 + e];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u;
                if (i < a.m) {
                    const int l0 = lv[u];
                    while (cur < l0) {  // slide the window: retire the leading partial sum
                        g[cur * st + o] = s0;
                        s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = 0.0;
                        cur++;
                        if (pass) {
                            c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5;
                            const double nv = c[(cur + a.k) * st + o];
                            if (a.k == 1) c1 = nv;
                            else if (a.k == 2) c2 = nv;
                            else if (a.k == 3) c3 = nv;
                            else if (a.k == 4) c4 = nv;
                            else c5 = nv;
                        }
                    }
                    const double *h = hv[u];
                    double r = buf[u];
                    if (pass) {
                        r -= h[0] * c0 + h[1] * c1;
                        if (a.k >= 2) r -= h[2] * c2;
                        if (a.k >= 3) r -= h[3] * c3;
                        if (a.k >= 4) r -= h[4] * c4;
                        if (a.k >= 5) r -= h[5] * c5;
                    }
                    s0 += h[0] * r;
                    s1 += h[1] * r;
                    if (a.k >= 2) s2 += h[2] * r;
                    if (a.k >= 3) s3 += h[3] * r;
                    if (a.k >= 4) s4 += h[4] * r;
                    if (a.k >= 5) s5 += h[5] * r;
                }
            }
        }
        // flush the last window (it ends at coefficient nc - 1)
        g[cur * st + o] = s0;
        g[(cur + 1) * st + o] = s1;
        if (a.k >= 2) g[(cur + 2) * st + o] = s2;
        if (a.k >= 3) g[(cur + 3) * st + o] = s3;
        if (a.k >= 4) g[(cur + 4) * st + o] = s4;
        if (a.k >= 5) g[(cur + 5) * st + o] = s5;
        if (pass)  // ... minus (B/p)' (B/p) c: the jump rows have a zero right-hand side
            for (int r = 0; r < a.nb; r++) {
                const double *b = a.Bp + (size_t)r * kSmBand;
                double v = 0.0;
                for (int e = 0; e < band; e++) v += b[e] * c[(r + e) * st + o];
                for (int e = 0; e < band; e++) g[(r + e) * st + o] -= b[e] * v;
            }
        // forward substitution R' w = g (w overwrites g)
        double w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0, w5 = 0.0, w6 = 0.0;  // w[j-1] .. w[j-6]
        for (int j = 0; j < nc; j++) {
            double sv = g[j * st + o];
            const double *Rj = a.R + (size_t)j * kSmBand;
            if (j >= 1) sv -= (Rj - 1 * kSmBand)[1] * w1;
            if (band > 2 && j >= 2) sv -= (Rj - 2 * kSmBand)[2] * w2;
            if (band > 3 && j >= 3) sv -= (Rj - 3 * kSmBand)[3] * w3;
            if (band > 4 && j >= 4) sv -= (Rj - 4 * kSmBand)[4] * w4;
            if (band > 5 && j >= 5) sv -= (Rj - 5 * kSmBand)[5] * w5;
            if (band > 6 && j >= 6) sv -= (Rj - 6 * kSmBand)[6] * w6;
            sv /= Rj[0];
            g[j * st + o] = sv;
            w6 = w5; w5 = w4; w4 = w3; w3 = w2; w2 = w1; w1 = sv;
        }
        // back substitution R x = w, then c = x (first pass) or c += x
        double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0, x5 = 0.0, x6 = 0.0;  // x[j+1] .. x[j+6]
        for (int j = nc - 1; j >= 0; j--) {
            double sv = g[j * st + o];
            const double *Rj = a.R + (size_t)j * kSmBand;
            sv -= Rj[1] * x1;  // (entries beyond the matrix are stored as zeros)
            if (band > 2) sv -= Rj[2] * x2;
            if (band > 3) sv -= Rj[3] * x3;
            if (band > 4) sv -= Rj[4] * x4;
            if (band > 5) sv -= Rj[5] * x5;
            if (band > 6) sv -= Rj[6] * x6;
            sv /= Rj[0];
            x6 = x5; x5 = x4; x4 = x3; x3 = x2; x2 = x1; x1 = sv;
            c[j * st + o] = pass ? c[j * st + o] + sv : sv;
        }
    }
    if (use_lds)
        for (int j = 0; j < nc; j++) c_glob[(size_t)j * nrhs + q] = c[j * st + o];
}

// out[j * rows + i] = in[i * cols + j] (LDS-tiled)
__global__ __launch_bounds__(kBlock) void k_transpose(const double *__restrict__ in, double *__restrict__ out, int rows,
                                                     int cols)
{
    __shared__ double tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    if (i < rows && j < cols) tile[ty][tx] = in[(size_t)i * cols + j];
    __syncthreads();
    i = blockIdx.y * 16 + tx;
    j = blockIdx.x * 16 + ty;
    if (i < rows && j < cols) out[(size_t)j * rows + i] = tile[tx][ty];
}

// Squared residuals of the fitted spline at the image pixels, summed per image row and per
// image column (the host turns them into fp and the per-knot-interval sums of FITPACK).
__global__ __launch_bounds__(kBlock) void k_sm_resid(const SmoothFitAxis ay, const SmoothFitAxis ax,
                                                     const double *__restrict__ z, const double *__restrict__ ct,
                                                     double *__restrict__ rowsum, double *__restrict__ colsum)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;  // image column
    const int i = blockIdx.y;                         // image row
    double term = 0.0;
    if (j < ax.m) {
        const int la = ay.lb[i], lb = ax.lb[j], nr = ay.nc;
        const double *hy = ay.hb + (size_t)i * 6, *hx = ax.hb + (size_t)j * 6;
        double sv = 0.0;
        for (int b = 0; b <= ax.k; b++) {
            double r = 0.0;
            for (int e = 0; e <= ay.k; e++) r += hy[e] * ct[(size_t)(lb + b) * nr + (la + e)];
            sv += hx[b] * r;
        }
        const double d = z[(size_t)i * ax.m + j] - sv;
        term = d * d;
        atomicAdd(&colsum[j], term);
    }
    // one atomic per wave for the row
    double rs = term;
    for (int off = 32; off > 0; off >>= 1) rs += __shfl_down(rs, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&rowsum[i], rs);
}

// bispev of a fitted smoothing spline at the map cells of one plane
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sm_eval(const ReprojectArgs a, const SmoothEvalArgs e)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const T *img = (const T *)a.cube + (size_t)e.plane * ny * nx;
    const double x = a.x_map[m], y = a.y_map[m];
    double val = nan;
    bool skip = isnan(x) || isnan(y);
    if (!skip && a.propagate_nan) {
        if (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1) {
            skip = true;
        } else {
            long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
            long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
            skip = isnan(load_as_f64(img, (size_t)ja * nx + ia)) || isnan(load_as_f64(img, (size_t)ja * nx + ib)) ||
                   isnan(load_as_f64(img, (size_t)jb * nx + ia)) || isnan(load_as_f64(img, (size_t)jb * nx + ib));
        }
    }
    if (!skip) {
        const double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
        // the knots are integer abscissae: the span of x is the span of floor(x)
        const int ly = e.span_rows[(int)yc], lx = e.span_cols[(int)xc];
        double hy[6], hx[6];
        SplineAxis ry = {e.t_rows, nullptr, e.nr, e.k_rows}, rx = {e.t_cols, nullptr, e.nc, e.k_cols};
        spline_basis(ry, yc, ly, hy);
        spline_basis(rx, xc, lx, hx);
        double sv = 0.0;
        for (int q = 0; q <= e.k_cols; q++) {
            double r = 0.0;
            for (int p = 0; p <= e.k_rows; p++) r += hy[p] * e.ct[(size_t)(lx - e.k_cols + q) * e.nr + (ly - e.k_rows + p)];
            sv += hx[q] * r;
        }
        val = sv;
    }
    a.out[(size_t)e.plane * a.n_map + m] = val;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_median_hist(const T *cube, size_t plane_elems, int shift, PlaneStats *stats,
                                                        unsigned int *hist /* [P][2][256] */)
{
    __shared__ unsigned int h[2][256];
    const int pl = blockIdx.y;
    h[0][threadIdx.x] = 0;
    h[1][threadIdx.x] = 0;
    __syncthreads();
    const T *img = cube + (size_t)pl * plane_elems;
    const unsigned long long mask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
    const unsigned long long pa = stats[pl].prefix[0], pb = stats[pl].prefix[1];
    unsigned int n_nan = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < plane_elems; i += (size_t)gridDim.x * kBlock) {
        const double v = (double)img[i];
        if (!isfinite(v)) {
            n_nan += isnan(v) ? 1u : 0u;
            continue;
        }
        const unsigned long long key = sortable_key(v);
        const unsigned int bin = (unsigned int)(key >> shift) & 255u;
        if ((key & mask) == pa) atomicAdd(&h[0][bin], 1u);
        if ((key & mask) == pb) atomicAdd(&h[1][bin], 1u);
    }
    __syncthreads();
    unsigned int *g = hist + (size_t)pl * 512;
    if (h[0][threadIdx.x]) atomicAdd(&g[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&g[256 + threadIdx.x], h[1][threadIdx.x]);
    if (shift == 56 && n_nan) atomicAdd(&stats[pl].n_nan, (unsigned long long)n_nan);
}

// one 256-thread block per plane: pick the bin holding each tracked rank, extend the prefix
__global__ __launch_bounds__(kBlock) void k_median_pick(int shift, size_t plane_elems, PlaneStats *stats, unsigned int *hist)
{
    const int pl = blockIdx.x;
    unsigned int *g = hist + (size_t)pl * 512;
    __shared__ unsigned long long cum[2][256];
    cum[0][threadIdx.x] = g[threadIdx.x];
    cum[1][threadIdx.x] = g[256 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 2) {
        const int s = threadIdx.x;
        PlaneStats &st = stats[pl];
        if (shift == 56) {
            unsigned long long n = 0;
            for (int b = 0; b < 256; b++) n += cum[s][b];
            st.n_finite = n;
            st.rank[s] = (n == 0) ? 0 : (s == 0 ? (n - 1) / 2 : n / 2);
        }
        unsigned long long k = st.rank[s], acc = 0;
        int bin = 0;
        for (int b = 0; b < 256; b++) {
            if (acc + cum[s][b] > k) {
                bin = b;
                break;
            }
            acc += cum[s][b];
        }
        st.rank[s] = k - acc;
        st.prefix[s] |= ((unsigned long long)bin) << shift;
    }
    __syncthreads();
    g[threadIdx.x] = 0;
    g[256 + threadIdx.x] = 0;
    if (shift == 0 && threadIdx.x == 0) {
        PlaneStats &st = stats[pl];
        // np.nanmedian: mean of the two middle values; 0.0 if nothing is finite (:1887-1890)
        st.median = st.n_finite ? 0.5 * (key_to_double(st.prefix[0]) + key_to_double(st.prefix[1])) : 0.0;
        st.all_nan = (st.n_nan == (unsigned long long)plane_elems) ? 1 : 0;
    }
}

}  // namespace pm

// ------------------------------------------------------------------ launchers (called from pm_capi.hip)
extern "C++" {

void pm_launch_disc(const pm::Params &p, int flags, hipStream_t s)
{
    dim3 grid((p.nx + pm::kBlock - 1) / pm::kBlock, p.rows);
    dim3 block(pm::kBlock);
    switch (flags & 7) {
    case 0: hipLaunchKernelGGL(pm::k_disc<0>, grid, block, 0, s, p); break;
    case 1: hipLaunchKernelGGL(pm::k_disc<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(pm::k_disc<2>, grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL(pm::k_disc<3>, grid, block, 0, s, p); break;
    case 4: hipLaunchKernelGGL(pm::k_disc<4>, grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL(pm::k_disc<5>, grid, block, 0, s, p); break;
    case 6: hipLaunchKernelGGL(pm::k_disc<6>, grid, block, 0, s, p); break;
    case 7: hipLaunchKernelGGL(pm::k_disc<7>, grid, block, 0, s, p); break;
    }
}

void pm_launch_disc_spheroid(const pm::Params &p, int flags, hipStream_t s)
{
    dim3 grid((p.nx + pm::kSphBlock - 1) / pm::kSphBlock, p.rows);
    dim3 block(pm::kSphBlock);
    const bool tri = p.radii[0] != p.radii[1];
#define PM_SPH_CASE(F)                                                                  \
    case F:                                                                             \
        if (tri) hipLaunchKernelGGL((pm::k_disc_sph<F, true>), grid, block, 0, s, p);   \
        else hipLaunchKernelGGL((pm::k_disc_sph<F, false>), grid, block, 0, s, p);      \
        break;
    switch (flags & 7) {
        PM_SPH_CASE(0)
        PM_SPH_CASE(1)
        PM_SPH_CASE(2)
        PM_SPH_CASE(3)
        PM_SPH_CASE(4)
        PM_SPH_CASE(5)
        PM_SPH_CASE(6)
        PM_SPH_CASE(7)
    }
#undef PM_SPH_CASE
}

void pm_launch_sky(const pm::Params &p, bool limb, hipStream_t s)
{
    dim3 grid((p.nx + pm::kBlock - 1) / pm::kBlock, p.rows);
    dim3 block(pm::kBlock);
    if (limb)
        hipLaunchKernelGGL(pm::k_sky<true>, grid, block, 0, s, p);
    else
        hipLaunchKernelGGL(pm::k_sky<false>, grid, block, 0, s, p);
}

void pm_launch_transform(const pm::Params &p, const pm::TransformArgs &t, hipStream_t s)
{
    unsigned long long blocks = (t.n + pm::kBlock - 1) / pm::kBlock;
    hipLaunchKernelGGL(pm::k_transform, dim3((unsigned)blocks), dim3(pm::kBlock), 0, s, p, t);
}

void pm_launch_map(const pm::Params &p, const double *lon, const double *lat, hipStream_t s)
{
    size_t n = (size_t)p.n0 * p.n1;
    dim3 grid((unsigned)((n + pm::kBlock - 1) / pm::kBlock));
    const uint64_t sun_bits = PM_PLANE_BIT(PM_PHASE) | PM_PLANE_BIT(PM_INCIDENCE) | PM_PLANE_BIT(PM_AZIMUTH) |
                              PM_PLANE_BIT(PM_LIMB_LON_GRAPHIC) | PM_PLANE_BIT(PM_LIMB_LAT_GRAPHIC) |
                              PM_PLANE_BIT(PM_LIMB_DISTANCE) | PM_PLANE_BIT(PM_RING_RADIUS) |
                              PM_PLANE_BIT(PM_RING_LON_GRAPHIC) | PM_PLANE_BIT(PM_RING_DISTANCE);
    const uint64_t state_bits = PM_PLANE_BIT(PM_RADIAL_VELOCITY) | PM_PLANE_BIT(PM_DOPPLER);
    const bool sun = (p.mask & sun_bits) != 0, state = (p.mask & state_bits) != 0;
    if (sun && state) hipLaunchKernelGGL((pm::k_map<true, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else if (sun) hipLaunchKernelGGL((pm::k_map<true, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else if (state) hipLaunchKernelGGL((pm::k_map<false, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else hipLaunchKernelGGL((pm::k_map<false, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
}

template <typename T>
static void launch_reproject_t(const pm::ReprojectArgs &a, hipStream_t s)
{
    dim3 grid((a.n_map + pm::kBlock - 1) / pm::kBlock, a.n_planes);
    hipLaunchKernelGGL(pm::k_reproject<T>, grid, dim3(pm::kBlock), 0, s, a);
}

template <typename T>
static void launch_median_t(const void *cube, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                            unsigned int *hist, hipStream_t s)
{
    unsigned gx = (unsigned)((plane_elems + pm::kBlock * 16 - 1) / (pm::kBlock * 16));
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(pm::k_median_hist<T>, dim3(gx, n_planes), dim3(pm::kBlock), 0, s, (const T *)cube,
                           plane_elems, shift, stats, hist);
        hipLaunchKernelGGL(pm::k_median_pick, dim3(n_planes), dim3(pm::kBlock), 0, s, shift, plane_elems, stats, hist);
    }
}

template <typename T>
static void launch_smooth_t(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, hipStream_t s)
{
    dim3 grid((a.n_map + pm::kBlock - 1) / pm::kBlock, a.n_planes);
    hipLaunchKernelGGL(pm::k_reproject_smooth<T>, grid, dim3(pm::kBlock), 0, s, a, sm);
}

void pm_launch_reproject_smooth(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_smooth_t<double>(a, sm, s); break;
    case PM_F32: launch_smooth_t<float>(a, sm, s); break;
    case PM_I16: launch_smooth_t<int16_t>(a, sm, s); break;
    case PM_I32: launch_smooth_t<int32_t>(a, sm, s); break;
    case PM_U8: launch_smooth_t<uint8_t>(a, sm, s); break;
    case PM_U16: launch_smooth_t<uint16_t>(a, sm, s); break;
    }
}

void pm_launch_map_limits(const double *x_map, const double *y_map, int n, double *limits, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_map_limits, dim3(1), dim3(pm::kBlock), 0, s, x_map, y_map, n, limits);
}

void pm_launch_reproject(const pm::ReprojectArgs &a, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_reproject_t<double>(a, s); break;
    case PM_F32: launch_reproject_t<float>(a, s); break;
    case PM_I16: launch_reproject_t<int16_t>(a, s); break;
    case PM_I32: launch_reproject_t<int32_t>(a, s); break;
    case PM_U8: launch_reproject_t<uint8_t>(a, s); break;
    case PM_U16: launch_reproject_t<uint16_t>(a, s); break;
    }
}

template <typename T>
static void launch_spline_t(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, hipStream_t s)
{
    const size_t npx = (size_t)a.ny * a.nx;
    hipLaunchKernelGGL(pm::k_spline_clean<T>, dim3((unsigned)((npx + pm::kBlock - 1) / pm::kBlock), a.n_planes),
                       dim3(pm::kBlock), 0, s, (const T *)a.cube, sa.work, a.plane_stats, a.ny, a.nx);
    size_t l0 = (size_t)a.n_planes * a.nx, l1 = (size_t)a.n_planes * a.ny;
    hipLaunchKernelGGL(pm::k_spline_solve, dim3((unsigned)((l0 + pm::kBlock - 1) / pm::kBlock)), dim3(pm::kBlock), 0, s,
                       sa.work, a.n_planes, a.ny, a.nx, 0, sa.rows);
    hipLaunchKernelGGL(pm::k_spline_solve, dim3((unsigned)((l1 + pm::kBlock - 1) / pm::kBlock)), dim3(pm::kBlock), 0, s,
                       sa.work, a.n_planes, a.ny, a.nx, 1, sa.cols);
    hipLaunchKernelGGL(pm::k_spline_eval<T>, dim3((a.n_map + pm::kBlock - 1) / pm::kBlock, a.n_planes), dim3(pm::kBlock),
                       0, s, a, sa);
}

template <typename T>
static void launch_clean_t(const pm::ReprojectArgs &a, double *work, hipStream_t s)
{
    const size_t npx = (size_t)a.ny * a.nx;
    hipLaunchKernelGGL(pm::k_spline_clean<T>, dim3((unsigned)((npx + pm::kBlock - 1) / pm::kBlock), a.n_planes),
                       dim3(pm::kBlock), 0, s, (const T *)a.cube, work, a.plane_stats, a.ny, a.nx);
}
// NaN-cleaned f64 copy of a.n_planes planes into `work` (a.plane_stats must hold the medians)
void pm_launch_clean(const pm::ReprojectArgs &a, double *work, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_clean_t<double>(a, work, s); break;
    case PM_F32: launch_clean_t<float>(a, work, s); break;
    case PM_I16: launch_clean_t<int16_t>(a, work, s); break;
    case PM_I32: launch_clean_t<int32_t>(a, work, s); break;
    case PM_U8: launch_clean_t<uint8_t>(a, work, s); break;
    case PM_U16: launch_clean_t<uint16_t>(a, work, s); break;
    }
}

void pm_launch_sm_solve(const pm::SmoothFitAxis &ax, const double *in, size_t si, size_t sq, int nrhs, double *g,
                        double *c, hipStream_t s)
{
    // both work vectors of a 64-lane workgroup in LDS when they fit (nc <= 146 of the 160 KB;
    // more than 64 KB of dynamic LDS has to be enabled per kernel)
    static const size_t lds_limit = [] {
        const int want = 150 * 1024;
        return hipFuncSetAttribute((const void *)pm::k_sm_solve, hipFuncAttributeMaxDynamicSharedMemorySize, want) ==
                       hipSuccess
                   ? (size_t)want
                   : (size_t)(64 * 1024);
    }();
    const size_t lds = (size_t)2 * ax.nc * 64 * sizeof(double);
    if (lds <= lds_limit)
        hipLaunchKernelGGL(pm::k_sm_solve, dim3((nrhs + 63) / 64), dim3(64), lds, s, ax, in, si, sq, nrhs, g, c, 1);
    else
        hipLaunchKernelGGL(pm::k_sm_solve, dim3((nrhs + pm::kBlock - 1) / pm::kBlock), dim3(pm::kBlock), 0, s, ax, in, si,
                           sq, nrhs, g, c, 0);
}
void pm_launch_transpose(const double *in, double *out, int rows, int cols, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_transpose, dim3((cols + 15) / 16, (rows + 15) / 16), dim3(pm::kBlock), 0, s, in, out, rows,
                       cols);
}
void pm_launch_sm_resid(const pm::SmoothFitAxis &ay, const pm::SmoothFitAxis &ax, const double *z, const double *ct,
                        double *rowsum, double *colsum, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_sm_resid, dim3((ax.m + pm::kBlock - 1) / pm::kBlock, ay.m), dim3(pm::kBlock), 0, s, ay, ax, z,
                       ct, rowsum, colsum);
}
void pm_launch_sm_eval(const pm::ReprojectArgs &a, const pm::SmoothEvalArgs &e, int dtype, hipStream_t s)
{
    dim3 grid((a.n_map + pm::kBlock - 1) / pm::kBlock), block(pm::kBlock);
    switch (dtype) {
    case PM_F64: hipLaunchKernelGGL(pm::k_sm_eval<double>, grid, block, 0, s, a, e); break;
    case PM_F32: hipLaunchKernelGGL(pm::k_sm_eval<float>, grid, block, 0, s, a, e); break;
    case PM_I16: hipLaunchKernelGGL(pm::k_sm_eval<int16_t>, grid, block, 0, s, a, e); break;
    case PM_I32: hipLaunchKernelGGL(pm::k_sm_eval<int32_t>, grid, block, 0, s, a, e); break;
    case PM_U8: hipLaunchKernelGGL(pm::k_sm_eval<uint8_t>, grid, block, 0, s, a, e); break;
    case PM_U16: hipLaunchKernelGGL(pm::k_sm_eval<uint16_t>, grid, block, 0, s, a, e); break;
    }
}

// a.plane_stats must already hold the plane statistics (pm_launch_plane_medians)
void pm_launch_spline(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_spline_t<double>(a, sa, s); break;
    case PM_F32: launch_spline_t<float>(a, sa, s); break;
    case PM_I16: launch_spline_t<int16_t>(a, sa, s); break;
    case PM_I32: launch_spline_t<int32_t>(a, sa, s); break;
    case PM_U8: launch_spline_t<uint8_t>(a, sa, s); break;
    case PM_U16: launch_spline_t<uint16_t>(a, sa, s); break;
    }
}

// stats / hist must be zero-filled by the caller (hipMemsetAsync) before this call
void pm_launch_plane_medians(const void *cube, int dtype, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                             unsigned int *hist, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_median_t<double>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_F32: launch_median_t<float>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_I16: launch_median_t<int16_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_I32: launch_median_t<int32_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_U8: launch_median_t<uint8_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_U16: launch_median_t<uint16_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    }
}
}
