// pm_kernels.hip -- gfx950 kernels of the planetmapper hot path.
//
//   k_disc_sph<FLAGS, BODY, SKY, MASK>  image-space planes that need the ray/ellipsoid intercept
//                   (lon/lat, centric, illumination, azimuth, LST, state, ring); BODY: spheroid / triaxial / general
//   k_sky<LIMB>     image-space planes defined for every pixel
//                   (RA/Dec, pixel x/y, km, angular, limb)
//   k_map_b0<SUN, STATE>  map-space planes of a lon/lat grid, in B0 (k_map: the J2000 evaluation, PM_OPT_GENERAL_KERNEL)
//   k_map_xy        the x/y map of a reprojection alone
//   k_transform     array-valued coordinate transforms
// (reprojection kernels: pm_kernels_reproject.hip)
//
// Launch geometry of the image kernels: one lane per pixel, a wave covers 64
// consecutive x of one row, so every plane store is one 512-byte fully coalesced
// write per wave; blockIdx.y = row. No loads besides the kernel-argument block.
#include "pm_device.hip.h"

namespace pm {


enum DiscFlags : int {
    DF_ILLUM = 1,  // PHASE / INCIDENCE / EMISSION / AZIMUTH
    DF_STATE = 2,  // DISTANCE / RADIAL-VELOCITY / DOPPLER
    DF_RING = 4,   // RING-RADIUS / RING-LON-GRAPHIC / RING-DISTANCE
};

// plane sets with their own instantiation of k_disc_sph (its MASK parameter)
constexpr unsigned long long plane_bit(int pl) { return 1ull << pl; }
constexpr unsigned long long kMaskHeadline =
    plane_bit(PM_LON_GRAPHIC) | plane_bit(PM_LAT_GRAPHIC) | plane_bit(PM_PHASE) | plane_bit(PM_INCIDENCE) | plane_bit(PM_EMISSION);
constexpr unsigned long long kMaskRings =
    kMaskHeadline | plane_bit(PM_RING_RADIUS) | plane_bit(PM_RING_LON_GRAPHIC) | plane_bit(PM_RING_DISTANCE);
constexpr unsigned long long kMaskDisc =
    kMaskRings | plane_bit(PM_LON_CENTRIC) | plane_bit(PM_LAT_CENTRIC) | plane_bit(PM_AZIMUTH) | plane_bit(PM_LOCAL_SOLAR_TIME) |
    plane_bit(PM_DISTANCE) | plane_bit(PM_RADIAL_VELOCITY) | plane_bit(PM_DOPPLER);
// (PM_MASK: the plane request; k_disc_sph redefines it for its body - a compile-time set in its MASK variants)
#define PM_MASK p.mask
#define PM_WANT(pl) ((PM_MASK >> (pl)) & 1ull)
#define PM_PUT(pl, val)                            \
    do {                                           \
        if (PM_WANT(pl)) p.out[pl][idx] = (val);   \
    } while (0)

// Store through a wave-uniform row pointer (SGPR pair) + 32-bit lane offset: the
// global_store saddr form, one address dword per lane instead of a 64-bit VGPR pointer.
// PM_EXPERIMENT (instrumented builds of tools/, never the shipped library): 1 = waves without an intercept do
// not store their NaN fill, 2 = no plane is stored at all (behind a condition the compiler cannot see through)
#ifndef PM_EXPERIMENT
#define PM_EXPERIMENT 0
#endif
#define PM_PUT_ROW(pl, val)                                               \
    do {                                                                  \
        if (PM_WANT(pl) && (PM_EXPERIMENT != 2 || p.nx < 0))              \
            *reinterpret_cast<double *>(reinterpret_cast<char *>(p.out[pl] + row_base) + lane_off) = (val); \
    } while (0)

// Reference loops fused in the image kernel k_disc_sph below: BodyXY._get_targvec_img body_xy.py:3195, _get_lonlat_img
// :3281, _get_lonlat_centric_img :3346, _get_illumination_gie_img :3658, get_azimuth_angle_img :3742,
// get_local_solar_time_img :3787, _get_state_imgs :3830, get_radial_velocity_img :3895, get_doppler_img :3938,
// _get_ring_plane_coordinate_imgs :4059.
// (Rounds 1-3 kept a second, general image kernel here - k_disc<FLAGS>: J2000 vectors, a 3 x 3 rotation matrix per
//  light-time evaluation, 0.41 ms per headline frame. Round 4 replaced it with the GEN mode of k_disc_sph, the same B0
//  formulation made exact for any body and observer: 0.21 ms, profiles/r04_disc_kernel_times.jsonl.)

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
// asin(n.z) of a unit vector n - the planetographic latitude of a spheroid's surface normal -
// with the machinery of vsep_fast against the pole: asin on |x| <= 0.5 only (half-chord to the
// nearer pole beyond 30 deg of latitude). One reciprocal square root; the atan2(z a / c, rho)
// form costs a square root for rho and a division.
__device__ __forceinline__ double lat_of_normal(V3 n)
{
    const double d = n.z;
    const bool mid = fabs(d) < 0.5;
    // (votes on the lane masks themselves: see wave_any)
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(true), mid_m = __builtin_amdgcn_ballot_w64(mid);
    if (mid_m == lanes) return asin_half(d);  // a wave wholly within 30 deg of the equator (see vsep_fast)
    // (half-chord to the nearer pole: (1 - |n.z|) / 2 while that keeps its accuracy, chosen per lane: see vsep_fast)
    const double h = fma(-0.5, fabs(d), 0.5);
    const unsigned long long close_m = ~mid_m & __builtin_amdgcn_ballot_w64(!(h > 5e-5));
    double s = sqrt_fast(h);
    if (close_m != 0) {
        const double wz = d + ((d > 0.0) ? -1.0 : 1.0);
        const double sd = 0.5 * sqrt_fast(fma(n.x, n.x, fma(n.y, n.y, wz * wz)));
        s = __builtin_amdgcn_inverse_ballot_w64(close_m) ? sd : s;
    }
    // (pi/2 - 2 r >= pi/6 towards the north pole and its negative towards the south: one FMA and the sign of d,
    //  the same bits as the two mirrored FMAs)
    if (mid_m == 0) return copysign(fma_m2_c(asin_half(s), kHalfPi), d);
    const double r = asin_half(mid ? d : s);
    return mid ? r : copysign(fma_m2_c(r, kHalfPi), d);
}

// pixel -> unit vector in the angular frame (BodyXY._xy2obsvec_norm body_xy.py:375: radrec of the view
// angles; the ray is M^T of it in J2000, C of it in B0). The affine map is taken in radians: folding
// arcsec -> rad into its six constants moves an angle of 1e-4 rad by 1 ulp.
// (the view angles of a frame that is not a planetary field of view: out of line, 2 KB of tiers that the
//  frame kernel's instruction stream would otherwise carry around its hot path - measured +1.4 % inlined)
struct SinCos2 {
    double sr, cr, sd, cd;
};
__device__ __attribute__((noinline)) SinCos2 sincos_wide(double ra, double de)  // (by value: in registers, no scratch)
{
    SinCos2 o;
    sincos_auto(ra, o.sr, o.cr);
    sincos_auto(de, o.sd, o.cd);
    return o;
}
__device__ __forceinline__ V3 pixel_va(const Params &p, const int x, const int y)
{
    // (about the disc centre, where the affine map has no constant term: Ar[2] = -(Ar[0] x0 + Ar[1] y0). One
    //  scalar operand per instruction - the constant term would first be copied into vector registers - and
    //  the image kernels hold x - x0, y - y0 for their pre-mask anyway.)
    const double dx = (double)x - p.x0, dy = (double)y - p.y0;
    const double ra = fma(p.Ar[0], dx, p.Ar[1] * dy);
    const double de = fma(p.Ar[3], dx, p.Ar[4] * dy);
    double sr, cr, sd, cd;
    if (p.view_tiny) {  // kernel-argument flag: a scalar branch, no wave vote
        sincos_tiny(ra, sr, cr);
        sincos_tiny(de, sd, cd);
    } else {
        const SinCos2 w = sincos_wide(ra, de);
        sr = w.sr, cr = w.cr, sd = w.sd, cd = w.cd;
    }
    return v3(cr * cd, sr * cd, sd);
}

// Reference loops fused here: BodyXY._get_radec_img body_xy.py:3409, get_x_img :3494,
// get_y_img :3519, _get_km_xy_img :3545, get_angular_x_img :3610,
// _get_limb_coordinate_imgs :3964. One pixel's planes from its angular-frame unit vector `va`; shared by
// k_sky (these planes alone) and by the SKY variants of k_disc_sph (every plane of a frame from one
// launch): the same code on the same `va`, the same bits.
// Constants through the laundered kernel-argument pointer: loaded here, not at kernel entry.
template <bool LIMB>
__device__ __forceinline__ void sky_block(const V3 va, const int x, const int y, const size_t row_base, const unsigned lane_off)
{
    const Params &p = *(const Params *)kernarg_params();
    if (PM_WANT(PM_RA) || PM_WANT(PM_DEC)) {
        double ra, dec;
        recrad_f(mtxv(p.g.M, va), ra, dec);  // the ray in J2000 (finite, non-zero for every pixel)
        PM_PUT_ROW(PM_RA, ra * kDeg);
        PM_PUT_ROW(PM_DEC, dec * kDeg);
    }
    PM_PUT_ROW(PM_PIXEL_X, (double)x);
    PM_PUT_ROW(PM_PIXEL_Y, (double)y);
    if (PM_WANT(PM_KM_X) || PM_WANT(PM_KM_Y) || PM_WANT(PM_ANGULAR_X) || PM_WANT(PM_ANGULAR_Y)) {
        // The reference takes each pixel's RA / Dec (in degrees) back to a ray and to angular coordinates
        // (radec2km, body_xy.py:3545-3552; Body._obsvec2angular body.py:1345): recrad of the vector the
        // view angles were turned into - i.e. the view angles themselves, the affine map of the pixel, as
        // long as they stay inside (-pi, pi) x (-pi/2, pi/2) (host: Params::view_direct; the round trip
        // through two sincos, a 3 x 3 product and two atan2 returns them to 1e-16 rad = 2e-11 arcsec).
        double ax, ay;
        if (p.view_direct) {
            ax = fma(p.A[0], (double)x, fma(p.A[1], (double)y, p.A[2]));
            ay = fma(p.A[3], (double)x, fma(p.A[4], (double)y, p.A[5]));
        } else {
            obsvec2angular_f(p, mtxv(p.g.M, va), ax, ay);
        }
        const double kx = fma(p.K[0], ax, p.K[1] * ay), ky = fma(p.K[2], ax, p.K[3] * ay);
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
        // Body._limb_coordinates_from_obsvec body.py:2081-2110 in B0, like the ring block of k_disc_sph:
        // the point of the ray nearest the body centre, PM's _obsvec2targvec of it (R(t) off =
        // Rz_frame(wdot (t - t0)) (R0 off), lengths are rotation invariant), the surface point under it
        // (surfpt_c from the centre = tv / |tv / radii|) and recpgr_c there.
        const V3 u = mxv(p.C, va);  // the ray in B0 (unit to 1e-16)
        const V3 o0 = v3(p.O0[0], p.O0[1], p.O0[2]);  // -R0 T0
        const double k = -div_fast(dot(o0, u), dot(u, u));
        const V3 nb = {fma(k, u.x, o0.x), fma(k, u.y, o0.y), fma(k, u.z, o0.z)};  // near point - T0, in B0
        const double nd = norm_f(nb);
        const V3 ob = {fma(k, u.x, -p.sub_obs_b[0]), fma(k, u.y, -p.sub_obs_b[1]), fma(k, u.z, -p.sub_obs_b[2])};
        const V3 w = ob - ld3(p.sub_ray_b);
        const double dd = norm_f(w) - p.g.sub_dist;
        const double t = p.g.sub_et - dd * p.inv_c;
        const double ang = p.g.wdot * (t - p.t0);
        double sa, ca;
        sincos_tiered<true>(ang, sa, ca);
        const V3 tv = {fma(ca, ob.x, sa * ob.y) + p.g.sub_sp[0], fma(ca, ob.y, -sa * ob.x) + p.g.sub_sp[1], ob.z + p.g.sub_sp[2]};
        const V3 X = {tv.x * p.ir[0], tv.y * p.ir[1], tv.z * p.ir[2]};
        const double sc = rsqrt_fast(dot(X, X));
        const V3 sfc = sc * tv;
        const double nx = sfc.x * p.limb_n[0], ny = sfc.y * p.limb_n[0], nz = sfc.z * p.limb_n[1];
        const double lat = atan2_fast(nz, sqrt_fast(fma(nx, nx, ny * ny)));
        double l = atan2_fast(sfc.y, sfc.x);
        if (p.g.west_positive) l = -l;
        if (l < 0.0) l += kTwoPi;
        PM_PUT_ROW(PM_LIMB_LON_GRAPHIC, l * kDeg);
        PM_PUT_ROW(PM_LIMB_LAT_GRAPHIC, lat * kDeg);
        PM_PUT_ROW(PM_LIMB_DISTANCE, nd - norm_f(sfc));
    }
}

template <bool LIMB>
__global__ __launch_bounds__(kBlock) void k_sky(const Params p_)
{
    const Params &p = *(const Params *)kernarg_params();
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = p.y_off + (int)blockIdx.y;
    if (x >= p.nx) return;
    sky_block<LIMB>(pixel_va(p, x, y), x, y, (size_t)blockIdx.y * p.nx, (unsigned)x * 8u);  // (row base wave-uniform: saddr stores)
}

// TRI: triaxial ellipsoid (a != b). The shape is no longer invariant under the spin, so each
// light-time evaluation first turns ray and observer by the spin angle of its epoch (a few
// 1e-5 rad: series) into the body-fixed frame and rescales the ray; the intercept is then
// body-fixed, and is turned back to B0 for the illumination geometry.
// SKY: 0 = the planes of the intercept only; 1 / 2 = the planes every pixel has as well (sky_block without /
// with the limb planes): all 26 planes of a frame from ONE launch, what save_observation asks for
// (observation.py:1269-1279). The sky planes go first: a wave issues their stores and computes its intercept
// while they drain.
// MASK: 0 = the planes of the request are read from p.mask: a test and a scalar branch per plane and store, in the
// NaN fill of the store-only waves as well - ten of them for the five planes of the headline set. Non-zero = the
// request is known to be exactly this set (the launcher checks): the tests fold away, -2.7 % on the headline
// frame. Instantiated for the sets the reference's own workloads ask for: BASELINE's headline (lon / lat / phase /
// incidence / emission), config 4 (+ the ring planes) and the whole intercept group of save_observation.
#undef PM_MASK
#define PM_MASK (MASK != 0 ? MASK : p.mask)
// BODY: 0 = spheroid (the closed-form / rotation-free path above), 1 = TRI as described, 2 = GEN: the same B0
// formulation for ANY body and observer - what the general kernel of rounds 1-3 (k_disc: J2000 vectors, a 3 x 3
// rotation matrix per light-time evaluation) did for 0.41 ms per headline frame: the spin angle of an evaluation by
// range-tiered sincos instead of a series (fast rotators, the 1000 s light-time spans of near-field geometry), the
// target's and the Sun's acceleration carried, an observer inside or on the surface (surfpt_c's far
// intersection), the Sun's light time iterated as illumf_c does. Serves near-field observers, fast spinners,
// large accelerations and PM_OPT_GENERAL_KERNEL.
// QUANT: the variant for geometries on which one quantum of the epoch et - lt is visible (Params::cf_iter /
// turn_quantum: every moon, every planet in data after 2015): the closed form steps through the reference's iterates,
// a spheroid turns its point and normal by the quantum. Its own instantiation, so that the geometries that need none
// of it - the 2005 fixtures, the benchmark - carry none of its scalar branches and registers (same-process A/B: the
// headline kernel was 1.7 % slower with them in).
template <int FLAGS, int BODY, int SKY = 0, unsigned long long MASK = 0, bool QUANT = false>
__global__ __launch_bounds__(kSphBlock) void k_disc_sph(const Params p)
{
    constexpr bool TRI = BODY != 0, GEN = BODY == 2;
    // Workgroups are dealt round-robin to the 8 XCDs (linear id % 8); with a row-major grid
    // each XCD would always get the same image columns, and the columns through the disc
    // centre cost far more than the ones at the frame edge. Rotating the column block by the
    // row index gives every XCD the same mix.
    const int x = (int)(p.col_blocks < 2 ? 0u : mod_uniform(blockIdx.x + blockIdx.y, p.col_blocks, p.col_magic)) * kSphBlock + threadIdx.x;
    // Rows are visited in a golden-ratio stride order (a bijection: gcd(row_stride, ny) = 1)
    // so that store-only rows off the disc and FP64-heavy rows through it are resident on the
    // chip at the same time: HBM writes of the former overlap the VALU work of the latter.
    // (rows <= 65535, the grid limit, and row_stride < rows: the product fits 32 bits)
    const int yl = (int)(p.rows < 2 ? 0u : mod_uniform(blockIdx.y * (uint32_t)p.row_stride, (uint32_t)p.rows, p.row_magic));  // row within this launch
    const int y = p.y_off + yl;
    const bool inside = x < p.nx;
    const size_t row_base = (size_t)yl * p.nx;  // wave-uniform
    const unsigned lane_off = (unsigned)x * 8u;  // byte offset in the row (< 4 GiB, checked by the host)
    const double nan = __builtin_nan("");

    const double dx = (double)x - p.x0, dy = (double)y - p.y0;
    const bool beyond = (dx * dx + dy * dy) > p.r2;

    double rr = nan, rl = nan, rd = nan;
    double dist_lt = nan;  // observer -> surface distance (lt * c) of on-disc pixels
    bool stored = false;   // wave-uniform: the disc planes of this wave have been written

    // candidates of the pre-mask as a lane mask (the only form they are used in; combined on the scalar unit)
    const unsigned long long cand_mask =
        __builtin_amdgcn_ballot_w64(inside) & ~(p.optimize_speed ? __builtin_amdgcn_ballot_w64(beyond) : 0ull);
    const bool any_cand = cand_mask != 0;
    V3 va;  // unit vector of the pixel in the angular frame (no default: a store-only wave would set it for nothing)
    if (any_cand || (FLAGS & DF_RING) || SKY != 0) va = pixel_va(p, x, y);
    // the planes every pixel has first: the wave issues their stores and computes its intercept while they drain
    if (SKY != 0 && inside) sky_block<SKY == 2>(va, x, y, row_base, lane_off);

    // the ray in B0 (both the ring and the disc block work there)
    V3 u;
    if (any_cand || (FLAGS & DF_RING)) u = mxv(p.C, va);

    if (FLAGS & DF_RING) {
        // Body._ring_coordinates_from_obsvec(only_visible=False) body.py:2577-2615 for EVERY
        // pixel: inrypl_c, PM's _obsvec2targvec (body.py:972-1006; it mixes J2000 and body-fixed
        // components by design) and recpgr_c of the in-plane point. (The reference rebuilds the ray
        // from RA/Dec in degrees, body_xy.py:3262; that round trip perturbs it by < 1 ulp - below
        // the rounding of the ray itself - and is not replayed.)
        // Everything is taken in B0: R0 (s ray - sub_obsvec) = s u - R0 sub_obsvec, lengths and the
        // plane equation are rotation invariant. The J2000 formulation needs the matrices M and R0
        // and five J2000 vectors as scalar constants on top of the disc block's - hipcc spilled 104
        // scalar registers to VGPR lanes (~270 v_readlane / v_writelane per wave); the B0 constants
        // are three vectors, and 18 FP64 operations of matrix products go as well.
        const double pd = dot(u, ld3(p.ring_nb));
        const double kk = p.g.ring_k;
        const bool ok = (kk == 0.0) ? (pd != 0.0) : (pd > 0.0 && kk < pd * (1.7976931348623157e308 / 3.0));
        if (wave_any(ok)) {
            // lanes without an intersection carry a harmless finite point through the math (at the sub-observer
            // point's distance: its light-time offset, and with it the spin angle below, stays tiny)
            const double s = !ok ? p.g.sub_dist : ((kk == 0.0) ? 0.0 : div_fast(kk, pd));
            const V3 ob = {fma(s, u.x, -p.sub_obs_b[0]), fma(s, u.y, -p.sub_obs_b[1]), fma(s, u.z, -p.sub_obs_b[2])};
            const V3 w = ob - ld3(p.sub_ray_b);
            const double dd = sqrt_fast(dot(w, w)) - p.g.sub_dist;
            const double t = p.g.sub_et - dd * p.inv_c;
            double sa, ca;
            const double ang = p.g.wdot * (t - p.t0);  // spin over the light-time offset: ~1e-4 rad
            sincos_tiered<true>(ang, sa, ca);
            // R(t) off = Rz_frame(ang) (R0 off)
            const V3 tv = {fma(ca, ob.x, sa * ob.y) + p.g.sub_sp[0], fma(ca, ob.y, -sa * ob.x) + p.g.sub_sp[1],
                           ob.z + p.g.sub_sp[2]};
            double le, alt;
            recpgr_alt_lon(p, tv, le, alt, ok);
            double l = p.g.west_positive ? -le : le;
            if (l < 0.0) l += kTwoPi;
            // (|s ray| = s: the ray is a unit vector to 1e-16, no norm needed)
            if (ok) {
                rr = alt + p.radii[0];
                rl = l * kDeg;
                rd = s;
            }
        }
    }

    if (any_cand) {
        // The disc block reads its constants through a laundered copy of the kernel-argument pointer:
        // hipcc otherwise loads them at kernel entry, ahead of the ring block, and - out of scalar
        // registers - parks them in VGPR lanes (v_writelane / v_readlane) until they are needed here.
        // (Params is the kernel's first and only argument: offset 0 of the kernel-argument segment.)
        const __attribute__((address_space(4))) Params *kp =
            (const __attribute__((address_space(4))) Params *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));

        // the intercept in scaled coordinates (unit sphere) and its epoch offset: read only by waves that hold an
        // intercept, which have set them (no defaults: four 64-bit moves per candidate wave)
        V3 Xf;
        double lt = kp->g.lt_c, d;
        // lanes still holding an intercept, as a wave-uniform mask in scalar registers (a
        // per-lane bool carried around the loop costs four VALU operations per evaluation)
        unsigned long long hit_mask = cand_mask;
        bool solved = false;  // wave-uniform: the closed form below has settled every lane of this wave
        // ... or, in a wave the limb runs through, the lanes clear of it (`cf_mask`): their closed-form state is kept
        // while the wave walks the reference's sequence for the others, and put back afterwards - what a pixel
        // gets does not depend on which pixels share its wave
        unsigned long long cf_mask = 0;
        V3 Xf_cf = {0.0, 0.0, 0.0};
        double d_cf = 0.0, lt_cf = 0.0;  // (set where cf_mask is; the defaults only quiet the other instantiations)
        double cz_cf = 1.0, sz_cf = 0.0;  // (TRI: the spin at the closed form's epoch)
        double cz = 1.0, sz = 0.0;  // spin since t0 at the epoch of the current evaluation (TRI)
        if (!GEN && (kp->plain_lt == 0 || (QUANT && kp->cf_iter)) && (!TRI || kp->tri_cf)) {  // (kernel-argument flags: a scalar branch)
            // sincpt_c 'CN' for a target in linear motion, in closed form. The converged light time is the fixed
            // point lt = E((et - lt) - t0) of the iteration further down: the target is taken at the epoch offset
            // d = (et - t0) - s / c when the ray meets it s km from the observer, so in the scaled frame of
            // surfpt_c the observer sits at Y(d) = O0s - VBs d = Y00 + s W (Y00 = O0s - VBs (et - t0), W = VBs / c)
            // and the point of the ray at s is Y00 + s (X + W): ONE intercept, with the ray direction bent by the
            // target's velocity, instead of a sequence of them. The rejection form (k, P, root) is kept - the
            // quadratic's discriminant would cancel eight digits at D / R ~ 1e4.
            const V3 Xp = {fma(u.x, kp->ir[0], kp->Wc[0]), fma(u.y, kp->ir[1], kp->Wc[1]), fma(u.z, kp->ir[2], kp->Wc[2])};
            const double ixp = rcp_fast(dot(Xp, Xp));
            const V3 y00 = {kp->Y00[0], kp->Y00[1], kp->Y00[2]};
            const double yx = dot(y00, Xp);
            const double kq = yx * ixp;
            // (Y00 ~ D / R ~ 1e4 radii is rounded at 1e-12: its low part goes in where the magnitude has dropped to 1)
            const V3 Pq = {fnma_c(kq, Xp.x, y00.x) + kp->Y00lo[0], fnma_c(kq, Xp.y, y00.y) + kp->Y00lo[1],
                           fnma_c(kq, Xp.z, y00.z) + kp->Y00lo[2]};
            const double p2 = dot(Pq, Pq);
            // The reference decides hit or miss at EVERY pass of its own sequence of epochs (CSPICE sincpt: no
            // intercept in any pass -> not found); between the first pass (target at t0) and the fixed point the
            // target moves |VBs| |d| <= |VBs| R / c across the ray, which changes P.P by less than 1 - p2_lo = p2_hi - 1 (host).
            // A lane clear of 1 by that band hits in every pass, or misses in the first; a wave with a lane
            // inside the band (the limb runs through it) walks the reference's sequence as well, for those lanes.
            // (TRI: the band widened by the turn of the shape over a light-time span, host: p2_lo_rot)
            const double band_lo = TRI ? kp->p2_lo_rot : kp->p2_lo, band_hi = TRI ? 2.0 - kp->p2_lo_rot : kp->p2_hi;
            unsigned long long hits =
                hit_mask & __builtin_amdgcn_ballot_w64(p2 < band_lo) & __builtin_amdgcn_ballot_w64(yx < 0.0);
            const unsigned long long misses = ~hit_mask | __builtin_amdgcn_ballot_w64(p2 > band_hi);
            // (without QUANT the votes are final here, and known before the arithmetic as in rounds 1-3: the order the
            //  compiler lays the blocks out in is worth 1-2 % of the headline frame)
            if (!QUANT) {
                solved = (hits | misses) == ~0ull;
                if (solved) hit_mask = hits;
                else cf_mask = hits;
            }
            // (a wave of the pre-mask annulus - candidates, but every ray misses - is done here)
            if (hits != 0) {
                // (clamped away from 0 for the reciprocal square root: lanes that miss carry garbage)
                const double r2 = fmax((1.0 - p2) * ixp, 1e-300);
                double half_inv_root;
                const double root = sqrt_pos_h(r2, half_inv_root);
                const double s = -kq - root;
                const double lts = mul_c(s, kp->inv_c);  // (rounded on its own, as the reference's light time is)
                double dq = 0.0;
                if (!QUANT) {
                    // (no quantum visible: the fixed point itself. Statement order as in round 3 - see QUANT above)
                    d_cf = rsub_c(lts, kp->g.et) - kp->t0;  // two roundings, as the epoch et - lt of the reference has them
                    dq = d_cf - rsub_c(lts, kp->lt_c_eff);
                }
                const V3 F = {fma(-root, Xp.x, Pq.x), fma(-root, Xp.y, Pq.y), fma(-root, Xp.z, Pq.z)};
                const V3 vbs = {kp->VBs[0], kp->VBs[1], kp->VBs[2]};
                // ds / dd of the intercept: s' = F.VBs / F.X, F.X = -root X.X (X' stands in for X: v / c ~ 1e-4 of 4e-12)
                const double sp = dot(F, vbs) * (half_inv_root * ixp) * -2.0;
                const double dstar = QUANT ? rsub_c(lts, kp->lt_c_eff) : 0.0;  // the epoch offset of the fixed point, unrounded
                // The light time whose epoch the reference evaluates its final state at. Its sequence
                // lt_0 = lt_c, lt_(k+1) = E(d_k), d_k = fl(fl(et - lt_k) - t0) stops at the first k >= 1 with
                // |lt_(k+1) - lt_k| <= 1e-17 |et - lt| and returns the state of epoch d_k: a point gamma^k (lt_c - lt*) from
                // the fixed point lt* (gamma = -E' ~ 1e-4: 4e-10 s after two passes over Jupiter's disc). Where one
                // quantum of the epoch (ulp(et): 3e-8 s in 2005, 1.2e-7 s from 2015 on) is NOT visible on the body that
                // difference is none either, and the fixed point itself is taken ...
                double lam = lts;
                if (QUANT && kp->cf_iter) {
                    // ... where it is (Params::cf_iter: the target moves by more than 1e-9 deg of its own radius per
                    // quantum - every moon, Mars, and Jupiter itself in data taken after 2015) the closed form steps
                    // through that sequence from the fixed point, to first order in E': e_k = lt_k - lt* = gamma^k e_0
                    // decides WHERE the reference stops (K), the rounded epoch of its last-but-one iterate gives
                    // lt_K = lt* + E' (d_(K-1) - d*) to 1e-14 s (the second-order term of lt_1, E'' e_0^2 / 2 ~ 1e-9 s,
                    // enters lt_K through gamma^(K-1) and through the quantum d_(K-1) rounds to: 1e-6 of the pixels).
                    // A lane whose deciding step lies within 10 % of the tolerance - the prediction of K could be off by
                    // one there - or that needs more than five passes leaves the closed form and walks the sequence.
                    double e1 = sp * kp->inv_c;
                    if (TRI) e1 = fma(kp->tri_k * F.x * F.y * ixp, (half_inv_root + half_inv_root) * kp->inv_c, e1);
                    const double gam = -e1, g1 = gam - 1.0, tol = kp->lt_tol;
                    // (nearly every wave clear of the limb stops at K = 2 in all its lanes: one vote, no loop)
                    const double e_1 = gam * dstar, st1 = fabs(e_1 * g1), st2 = st1 * fabs(gam);
                    const unsigned long long k2_m = __builtin_amdgcn_ballot_w64(st1 > 1.1 * tol && st2 < 0.9 * tol && fabs(gam) <= 3e-4);
                    bool unsure = false;
                    if ((hits & ~k2_m) == 0) {
                        lam = fma(e1, (rsub_c(lts + e_1, kp->g.et) - kp->t0) - dstar, lts);
                    } else {
                        double e_prev = dstar, e_cur = e_1, e_km1 = dstar;
                        bool done = false, first = false;
#pragma unroll
                        for (int k = 1; k <= 5; k++) {
                            const double step = fabs(e_cur * g1);
                            const bool stop = step <= tol;
                            unsure = unsure || (!done && fabs(step - tol) <= 0.1 * tol);
                            e_km1 = (!done && stop) ? e_prev : e_km1;
                            if (k == 1) first = stop;
                            done = done || stop;
                            e_prev = e_cur;
                            e_cur *= gam;
                        }
                        // (... and so does a lane towards the limb, where E' grows like 1 / cos(emission) and the sequence is
                        //  no longer a geometric one to the precision needed: beyond ~82 deg for a planet seen from afar)
                        unsure = unsure || !done || fabs(gam) > 3e-4;
                        // d_(K-1) as the reference rounds it (d_0 = 0 exactly), then lt_K
                        const double d_km1 = first ? 0.0 : rsub_c(lts + e_km1, kp->g.et) - kp->t0;
                        lam = fma(e1, d_km1 - dstar, lts);
                    }
                    hits &= ~__builtin_amdgcn_ballot_w64(unsure);
                }
                // The reference evaluates its final state at the epoch et - lt ROUNDED to a double (one quantum), the
                // closed form at the unrounded fixed point. The target's linear motion over the difference `dq`
                // moves the intercept by dq (s' X - VBs), s' keeping it on the surface: first order is exact here
                // (dq^2 ~ 1e-14).
                if (QUANT) {
                    d_cf = rsub_c(lam, kp->g.et) - kp->t0;  // two roundings, as the epoch et - lt of the reference has them
                    dq = d_cf - dstar;
                }
                Xf_cf = {fma(dq, fma_cn(sp, Xp.x, vbs.x), F.x), fma(dq, fma_cn(sp, Xp.y, vbs.y), F.y),
                         fma(dq, fma_cn(sp, Xp.z, vbs.z), F.z)};
                lt_cf = fma(dq, sp, s) * kp->inv_c;
                if (TRI) {
                    // A triaxial body (Params::tri_cf: a real moon - the turn of its shape over a light-time span is
                    // second-order small). So far the body was FROZEN in its B0 orientation; at the epoch d it has
                    // turned by wdot d under the ray, which moves the intercept along the ray by kappa d,
                    // kappa = wdot (b/a - a/b) Xf_x Xf_y / (root X.X) (the slope of the Newton step below; 4e-6 km
                    // for Io - first order is exact to 1e-12 km). The point then goes into the body-fixed frame of
                    // its epoch, where longitude, latitude and the normal are taken.
                    const double dt = kp->tri_k * Xf_cf.x * Xf_cf.y * ixp * (half_inv_root + half_inv_root) * d_cf;
                    const V3 p0 = {fma(dt, u.x, Xf_cf.x * kp->radii[0]), fma(dt, u.y, Xf_cf.y * kp->radii[1]),
                                   fma(dt, u.z, Xf_cf.z * kp->radii[2])};
                    const double dl = kp->g.wdot * d_cf, d2 = dl * dl;
                    cz_cf = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
                    sz_cf = dl * fma(d2, -1.0 / 6.0, 1.0);
                    Xf_cf = {fma(cz_cf, p0.x, sz_cf * p0.y) * kp->ir[0], fma(cz_cf, p0.y, -sz_cf * p0.x) * kp->ir[1], p0.z * kp->ir[2]};
                    lt_cf = fma(dt, kp->inv_c, lt_cf);
                }
                if (!QUANT && solved) {
                    Xf = Xf_cf;
                    d = d_cf;
                    lt = lt_cf;
                    if (TRI) {
                        cz = cz_cf;
                        sz = sz_cf;
                    }
                }
            }
            if (QUANT) {
                // (lanes the stepping was unsure about have left `hits`)
                solved = (hits | misses) == ~0ull;
                if (solved) hit_mask = hits;
                else cf_mask = hits;
                if (solved && hits != 0) {
                    Xf = Xf_cf;
                    d = d_cf;
                    lt = lt_cf;
                    if (TRI) {
                        cz = cz_cf;
                        sz = sz_cf;
                    }
                }
            }
        }

        // surfpt_c in scaled coordinates; for a spheroid X and 1/(X.X) are fixed for the pixel
        V3 X = {u.x * kp->ir[0], u.y * kp->ir[1], u.z * kp->ir[2]};
        double ixx = 0.0;
        if (!solved) {
        d = 0.0;
        ixx = rcp_fast(dot(X, X));

        // sincpt_c 'CN': converged light time, CSPICE stopping rule, <= 10 evaluations
        // CSPICE's rule is |dlt| <= 1e-17 |et - lt|; lt varies by 1e-9 relative over a disc
        // (sroot: the signed root - the intercept is P + sroot X, -root for an observer outside the body)
        double k = 0.0, root = 0.0, sroot = 0.0, inv_root = 0.0, p2_first = 0.0;
        bool outside = true;  // (GEN) the observer of the lane's latest evaluation is outside the body
        V3 P = {0.0, 0.0, 0.0};
        // An FMA takes one scalar operand: with VBs there, O0s has to sit in vector registers.
        // Pinned outside the loop (left alone, hipcc re-copies the three pairs every evaluation).
        double o0x = kp->O0s[0], o0y = kp->O0s[1], o0z = kp->O0s[2];
        if (!TRI) asm volatile("" : "+v"(o0x), "+v"(o0y), "+v"(o0z));
        // One evaluation of the intercept with the target taken d seconds after t0; returns the
        // light time it implies. (The target's acceleration moves it by A d^2 / 2 < 1e-8 km over
        // the |d| <= R / c of a disc intercept, 10x below the rounding of the ray itself: not
        // carried here.)
        auto evaluate = [&](const double dd, const bool first) -> double {
            V3 Y;
            if (first) {
                // t0 = et - lt_c on the host, the same subtraction as on the device: d == 0 exactly
                Y = TRI ? v3(kp->O0[0] * kp->ir[0], kp->O0[1] * kp->ir[1], kp->O0[2] * kp->ir[2]) : v3(o0x, o0y, o0z);
            } else if (TRI) {
                V3 obs = {fma(-kp->VB[0], dd, kp->O0[0]), fma(-kp->VB[1], dd, kp->O0[1]), fma(-kp->VB[2], dd, kp->O0[2])};
                const double dl = kp->g.wdot * dd;
                if (GEN) {
                    const double h = -0.5 * dd * dd;  // the target's acceleration: obs = -(T0 + VT d + AT d^2 / 2) in B0
                    obs = {fma(kp->AB[0], h, obs.x), fma(kp->AB[1], h, obs.y), fma(kp->AB[2], h, obs.z)};
                    sincos_tiered<true, false>(dl, sz, cz);  // (the tier is the lane's own; |wdot d| stays far below 1e5 rad)
                } else {
                    const double d2 = dl * dl;  // |dl| < 1e-3 (host check)
                    cz = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
                    sz = dl * fma(d2, -1.0 / 6.0, 1.0);
                }
                const V3 ub = {fma(cz, u.x, sz * u.y), fma(cz, u.y, -sz * u.x), u.z};
                X = {ub.x * kp->ir[0], ub.y * kp->ir[1], ub.z * kp->ir[2]};
                ixx = rcp_fast(dot(X, X));
                Y = {fma(cz, obs.x, sz * obs.y) * kp->ir[0], fma(cz, obs.y, -sz * obs.x) * kp->ir[1], obs.z * kp->ir[2]};
            } else {
                Y = {fma(-kp->VBs[0], dd, o0x), fma(-kp->VBs[1], dd, o0y), fma(-kp->VBs[2], dd, o0z)};
            }
            const double yx = dot(Y, X);
            k = yx * ixx;
            P = {fma(-k, X.x, Y.x), fma(-k, X.y, Y.y), fma(-k, X.z, Y.z)};
            const double p2 = dot(P, P);
            if (first) p2_first = p2;
            double y2 = 2.0;
            if (GEN) {
                // surfpt_c: a ray from outside misses when it passes the centre by more than the (unit) radius or
                // points away; from inside or on the surface it always meets the surface
                y2 = dot(Y, Y);
                outside = y2 > 1.0;
                hit_mask &= ~(__builtin_amdgcn_ballot_w64(outside) & (__builtin_amdgcn_ballot_w64(p2 > 1.0) | __builtin_amdgcn_ballot_w64(yx > 0.0)));
            } else {
                // (an observer inside the body, Y.Y <= 1, never reaches the fast paths: the
                //  launcher requires |O0| scaled > 2 and the target moves km, not radii)
                hit_mask &= ~(__builtin_amdgcn_ballot_w64(p2 > 1.0) | __builtin_amdgcn_ballot_w64(yx > 0.0));
            }
            // (clamped away from 0 once, for the reciprocal square root: a grazing ray gets
            //  root = 1e-150 instead of 0)
            const double r2 = fmax((1.0 - p2) * ixx, 1e-300);
            // the first light time only seeds the next epoch (an error e in it moves the target
            // by VB e): 2^-45 relative is plenty there
            if (first) {
                // (sqrt_seed_pos, with its reciprocal-square-root estimate kept for the slope below)
                inv_root = __builtin_amdgcn_rsq(r2);
                const double g = r2 * inv_root;
                root = fma(g, fma(-0.5 * inv_root, g, 0.5), g);
            } else {
                root = sqrt_pos(r2);
            }
            // (outside: the near intersection; inside: the one ahead; ON the surface surfpt_c returns the observer's own point)
            sroot = !GEN ? -root : (outside ? -root : (y2 == 1.0 ? k : root));
            return (sroot - k) * kp->inv_c;
        };
        // The first evaluation, at t0 itself, is never the last: |lt - lt_c| would have to be
        // below 1e-17 |t0| ~ 1e-8 s for every pixel of the wave, and one more evaluation of a
        // converged light time changes nothing. No test, no select after it.
        lt = evaluate(0.0, true);
        if (!TRI && kp->plain_lt != 1) {  // (kernel-argument flag: a scalar branch)
            // One Newton-like step on that seed. The light time is a smooth function E(d) of the epoch
            // offset d, and the fixed point lt = E((et - lt) - t0) is what the iteration below converges
            // to. With the slope E'(0) = [VBs.X - (P.VBs) / root] / (X.X c) (the target's velocity along
            // the ray and, through the shrinking chord, across it), lt1 + E' d / (1 + E') is within the
            // stopping tolerance of the fixed point wherever the ray does not graze (the neglected
            // E'' d^2 / 2 is ~2e-10 s / cos(emission) for Jupiter against a tolerance of 1.7e-9 s): the
            // next evaluation then already confirms convergence, and most waves do two evaluations
            // instead of three.
            // Lanes near the limb keep the plain seed (|E'| >= 0.02, i.e. cos(emission) below ~0.005 for
            // Jupiter): there E is a square root in d, a linear step overshoots, and - what matters -
            // the reference decides hit or miss at EVERY evaluation of its own sequence of epochs
            // (CSPICE sincpt: no intercept in any pass -> not found). A grazing ray (half chord of a few
            // km) can hit with the target at one epoch of that sequence and miss at another, so those
            // lanes must see exactly the reference's epochs. For the lanes that take the step the epoch
            // error it leaves moves the target by ~E'^3 of the ray's margin to the limb: < 1e-5.
            const V3 vbs = {kp->VBs[0], kp->VBs[1], kp->VBs[2]};
            const double pv = dot(P, vbs);
            const double ep = fma(-pv, inv_root, dot(vbs, X)) * ixx * kp->inv_c;
            const double d0 = (kp->g.et - lt) - kp->t0;
            const double stepped = fma(ep * d0, 1.0 - ep, lt);
            // ... and the step must not jump over a pass of the reference's sequence at which the ray MISSES:
            // its second pass sits at t0 + d0, the farthest of all from the fixed point (the later ones lie
            // between the two), and the squared half chord there is root^2 + 2 (P.VBs) d0 / X.X to first order
            // (the target moves 3 km across the ray in the 0.24 s of a Jupiter radius). A near observer brought
            // it up (fuzz seed 900030: 19 radii away, emission 89.992 deg at the fixed point, a miss at pass 2).
            const double r2_pass2 = fma((pv + pv) * d0, ixx, root * root);
            lt = (fabs(ep) < 0.02 && r2_pass2 > 1e-9 * ixx) ? stepped : lt;
        }
        if (TRI && kp->plain_lt != 1) {
            // The same step for a body whose shape turns under the ray. With the intercept x = obs + t u in the
            // body-fixed frame and the surface G(x) = |x / radii|^2 = 1, dt / dd = -grad G . (obs' + t u') / (grad G . u):
            // grad G . u = 2 Xf . X = 2 sroot X.X; the translation gives -Xf . VBs, and the two rotation terms - the
            // observer and the ray turn TOGETHER, each by wdot D ~ 1e4 km/s - add up to the motion of the surface
            // itself, wdot Xf_x Xf_y (b / a - a / b): zero for a spheroid, 1e-9 for Io. No cancellation is left.
            // Lanes the limb could touch in a later pass of the reference's sequence (the band of the closed form,
            // widened by the turn of the shape over a light-time span - host: p2_lo_rot) keep the plain seed, and so
            // do lanes of an observer inside the body.
            const V3 xf1 = {fma(sroot, X.x, P.x), fma(sroot, X.y, P.y), fma(sroot, X.z, P.z)};
            const double a = fma(kp->tri_k * xf1.x, xf1.y, -dot(xf1, v3(kp->VBs[0], kp->VBs[1], kp->VBs[2])));
            const double ep = a * ixx * inv_root * kp->inv_c;
            const double d0 = (kp->g.et - lt) - kp->t0;
            const double stepped = fma(ep * d0, 1.0 - ep, lt);
            lt = (fabs(ep) < 0.02 && p2_first < kp->p2_lo_rot && outside) ? stepped : lt;
        }
        // (a wave of the pre-mask annulus - candidates, but every ray misses - is done after that one
        //  evaluation: nothing is left to converge)
        // Each lane stops where the reference stops for its pixel: once an evaluation has confirmed its light
        // time, the lane keeps that epoch while the wave goes on for the others (evaluating it again gives the
        // same state) - advancing it would move it to another epoch quantum depending on its company.
        double lt_e = lt;  // the light time the next epoch is formed from
#pragma unroll 1
        for (int it = 1; it < 10 && hit_mask != 0; it++) {
            d = (kp->g.et - lt_e) - kp->t0;  // two roundings, as the epoch et - lt of the reference has them
            lt = evaluate(d, false);  // (lanes without an intercept carry a value nobody reads)
            const bool moving = !(fabs(lt - lt_e) <= kp->lt_tol);
            lt_e = moving ? lt : lt_e;
            // wave-uniform exit once no lane with an intercept is still moving
            if ((hit_mask & __builtin_amdgcn_ballot_w64(moving)) == 0) break;
        }
        Xf = {fma(sroot, X.x, P.x), fma(sroot, X.y, P.y), fma(sroot, X.z, P.z)};
        if (!GEN && cf_mask != 0) {  // (those lanes hit in every pass: they are in hit_mask)
            const bool cf = __builtin_amdgcn_inverse_ballot_w64(cf_mask);
            Xf = {cf ? Xf_cf.x : Xf.x, cf ? Xf_cf.y : Xf.y, cf ? Xf_cf.z : Xf.z};
            d = cf ? d_cf : d;
            lt = cf ? lt_cf : lt;
            if (TRI) {
                cz = cf ? cz_cf : cz;
                sz = cf ? sz_cf : sz;
            }
        }
        }  // !solved

        // From here on EVERY lane of a wave that holds at least one intercept computes: no
        // exec masking and no NaN-initialised result registers (16 v_mov per candidate wave).
        // Lanes without an intercept carry harmless garbage; `miss` (NaN for them, 0.0 for
        // hits) is the addend of each plane's closing radians -> degrees FMA.
        if (hit_mask != 0) {
            static_assert(kSphBlock == 64, "lane id = threadIdx.x: one wave per workgroup");
            const bool hit = __builtin_amdgcn_inverse_ballot_w64(hit_mask);  // (the mask IS the select operand)
            stored = true;
            const double miss = hit ? 0.0 : nan;
            // intercept in B0; body-fixed = Rz_frame(delta) * B0 with delta = wdot d
            if (FLAGS & (DF_RING | DF_STATE)) dist_lt = fma(lt, kp->g.clight, miss);
            // body-fixed at te for TRI, B0 otherwise (body-fixed = Rz_frame(delta) * B0)
            const V3 sp = {Xf.x * kp->radii[0], Xf.y * kp->radii[1], Xf.z * kp->radii[2]};
            // (for a spheroid x and y share their radius: longitude and latitude follow from the
            //  scaled intercept Xf directly, sp / rho are only needed by the triaxial variant)
            const V3 ll = TRI ? sp : Xf;
            // recpgr_c body.py:1030: east longitude in the frame at te = B0 longitude - wdot d,
            // sign by the body's convention (lon_k = {+-1, +-wdot}), then into [0, 2 pi]
            // (a point on the axis, x = y = 0, has longitude 0: ZERO_OK)
            const double theta = atan2_fast<false, true>(ll.y, ll.x);
            double l = TRI ? kp->lon_k[0] * theta : fma(-kp->lon_k[1], d, kp->lon_k[0] * theta);
            l = fma((l < 0.0) ? 1.0 : 0.0, kTwoPi, l);  // (l + 2 pi, rounded once either way: a select of one word)
            const double lon_deg = fma(l, kDeg, miss);
            // surfnm_c: sp / radii^2 = Xf / radii; for a spheroid its z component IS sin(latitude)
            // (a triaxial body's latitude refers to the reference spheroid instead, recpgr_c)
            V3 n = {Xf.x * kp->ir[0], Xf.y * kp->ir[1], Xf.z * kp->ir[2]};
            constexpr bool kLatFromNormal = (FLAGS & DF_ILLUM) && !TRI;  // (the normal is needed anyway)
            if (kLatFromNormal) n = rsqrt_fast(dot(n, n)) * n;
            double lat;
            if (kLatFromNormal) {
                lat = lat_of_normal(n);
            } else {
                // (|Xf| = 1: rho and z never vanish together, atan2 needs no guard here)
                const double rho = sqrt_fast(fma(ll.x, ll.x, ll.y * ll.y));
                lat = atan2_fast<true>(TRI ? sp.z * kp->lat_k : Xf.z * kp->a_over_c, rho);
            }
            const double lat_deg = fma(lat, kDeg, miss);
            if (inside) {
                PM_PUT_ROW(PM_LON_GRAPHIC, lon_deg);
                PM_PUT_ROW(PM_LAT_GRAPHIC, lat_deg);
                if (PM_WANT(PM_LOCAL_SOLAR_TIME)) PM_PUT_ROW(PM_LOCAL_SOLAR_TIME, local_solar_time(p, lon_deg));
            }
            if (PM_WANT(PM_LON_CENTRIC) || PM_WANT(PM_LAT_CENTRIC)) {
                // reclat_c body.py:2905: east-positive, (-pi, pi]
                double lc = TRI ? theta : fma(-kp->g.wdot, d, theta);
                if (lc <= -kPi) lc += kTwoPi;
                if (lc > kPi) lc -= kTwoPi;
                const double rho = sqrt_fast(fma(ll.x, ll.x, ll.y * ll.y));
                const double bc = atan2_fast<true>(TRI ? sp.z : Xf.z / kp->a_over_c, rho);
                if (inside) {
                    PM_PUT_ROW(PM_LON_CENTRIC, fma(lc, kDeg, miss));
                    PM_PUT_ROW(PM_LAT_CENTRIC, fma(bc, kDeg, miss));
                }
            }
            // illumf_c and spkcpt_c solve the surface point's OWN light time: their iteration ends on the fixed point, one
            // contraction further than the last iterate sincpt_c stopped at (1e-10 s: the same double as a rule, the next
            // epoch quantum in a fraction of a percent of the pixels). `lt` - the light time the last evaluation returned
            // for sincpt's epoch - is that fixed point to 1e-14 s. Where a quantum is visible (Params::plain_lt: a fast
            // rotator turns by wdot x quantum, a near target moves by VT x quantum) the illumination and the state are
            // taken at ITS epoch, with the body-fixed point sincpt found, as the reference does.
            double di = d, czi = cz, szi = sz;
            if (TRI && kp->plain_lt == 1) {
                di = (kp->g.et - lt) - kp->t0;
                const double dl = kp->g.wdot * di;
                if (GEN) {
                    sincos_tiered<true, false>(dl, szi, czi);
                } else {
                    const double d2 = dl * dl;
                    czi = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
                    szi = dl * fma(d2, -1.0 / 6.0, 1.0);
                }
            }
            // (A spheroid needs none of that for its shape - but the body-fixed POINT turns with the body between the two
            //  epochs, and its normal with it: by wdot x (di - d), one quantum's worth in the fraction of a percent of the
            //  pixels whose two epochs round apart. Params::turn_quantum: 1.2e-9 deg for Jupiter from 2015 on.)
            // the point in B0 (for the Sun / observer geometry, which lives there)
            const V3 sp0 = TRI ? v3(fma(czi, sp.x, -szi * sp.y), fma(szi, sp.x, czi * sp.y), sp.z) : sp;
            if (FLAGS & DF_ILLUM) {
                // illumf_c body.py:1915: point wrt P_T(t0) in B0; Sun light time: two passes
                V3 q = TRI ? v3(fma(kp->VB[0], di, sp0.x), fma(kp->VB[1], di, sp0.y), fma(kp->VB[2], di, sp0.z))
                           : v3(fma(kp->VBs[0], d, Xf.x) * kp->radii[0], fma(kp->VBs[1], d, Xf.y) * kp->radii[1],
                                fma(kp->VBs[2], d, Xf.z) * kp->radii[2]);
                if (GEN) {
                    const double h = 0.5 * di * di;
                    q = {fma(kp->AB[0], h, q.x), fma(kp->AB[1], h, q.y), fma(kp->AB[2], h, q.z)};
                }
                // Sun light time (spkcpo_c 'CN'): the Sun is taken at te - |S - q| / c. Its epoch
                // offset from ts0 is d + (lts0 - |SB0 - q| / c), and |SB0 - q| = |SB0| - s0.q up to
                // q^2 / (2 |SB0|) ~ 3 km, i.e. 1e-5 s of a Sun that moves 0.013 km/s = 1e-7 km at
                // 8e8 km (2e-16 rad): the square root of the exact form buys nothing, the linear
                // form is one dot product.
                // (operations ordered so that each holds ONE scalar constant: two of them in an FMA cost a copy
                //  of one into a vector register pair first)
                V3 sv;
                if (GEN) {
                    // illumf_c's own iteration (spkcpo_c 'CN'): two passes from the centre value, the Sun's acceleration carried
                    const double dts = (kp->t0 - kp->g.ts0) + di;  // te - ts0
                    double ds = 0.0;
#pragma unroll
                    for (int it = 0; it < 3; it++) {
                        const double h = 0.5 * ds * ds;
                        sv = {fma(kp->ASB[0], h, fma(kp->VSB[0], ds, kp->SB0[0])) - q.x, fma(kp->ASB[1], h, fma(kp->VSB[1], ds, kp->SB0[1])) - q.y,
                              fma(kp->ASB[2], h, fma(kp->VSB[2], ds, kp->SB0[2])) - q.z};
                        if (it < 2) ds = dts - norm_f(sv) * kp->inv_c;
                    }
                } else {
                const double ds = fma(dot(v3(kp->SB0[0], kp->SB0[1], kp->SB0[2]), q), kp->sun_k, TRI ? di : d) + kp->sun_ds0;
                sv = v3(fma(kp->VSB[0], ds, rsub_c(q.x, kp->SB0[0])), fma(kp->VSB[1], ds, rsub_c(q.y, kp->SB0[1])),
                        fma(kp->VSB[2], ds, rsub_c(q.z, kp->SB0[2])));
                }
                const V3 sunb = rsqrt_fast(dot(sv, sv)) * sv;
                const V3 ob = neg(u);  // observer seen from the point: -ray (unit)
                if (QUANT && !TRI && kp->turn_quantum) {  // (kernel-argument flag: one scalar branch; first order: 1e-10 rad)
                    const double turn = kp->g.wdot * (((kp->g.et - lt) - kp->t0) - d);
                    n = {fma(-turn, n.y, n.x), fma(turn, n.x, n.y), n.z};
                }
                if (TRI) {
                    n = {fma(czi, n.x, -szi * n.y), fma(szi, n.x, czi * n.y), n.z};
                    n = rsqrt_fast(dot(n, n)) * n;
                }
                // phase angle: a four-term series in cos g about the body centre's value where the host
                // found it good to 1e-15 rad (every planet seen from afar), vsep_c otherwise
                double phr;
                if (kp->phase_series) {  // kernel-argument flag: a scalar branch
                    const double t = dot(sunb, ob) - kp->ph[0];
                    phr = fma_c(t, fma_c(t, fma_c(t, horner_head(t, kp->ph[5], kp->ph[4]), kp->ph[3]), kp->ph[2]), kp->ph[1]);
                } else {
                    phr = vsep_fast(sunb, ob);
                }
                const double ph = fma(phr, kDeg, miss);
                const double in = fma(vsep_fast(n, sunb), kDeg, miss);
                const double em = fma(vsep_fast(n, ob), kDeg, miss);
                if (inside) {
                    PM_PUT_ROW(PM_PHASE, ph);
                    PM_PUT_ROW(PM_INCIDENCE, in);
                    PM_PUT_ROW(PM_EMISSION, em);
                    if (PM_WANT(PM_AZIMUTH)) PM_PUT_ROW(PM_AZIMUTH, azimuth_from_cosines(dot(sunb, ob), dot(n, sunb), dot(n, ob)) + miss);
                }
            }
            if (FLAGS & DF_STATE) {
                // spkcpt_c body.py:2830: distance = lt c; velocity with the light-time rate
                const double dv = TRI ? di : d;
                V3 sv0 = sp0;  // the point the state belongs to
                if (QUANT && !TRI && kp->turn_quantum) {
                    const double turn = kp->g.wdot * (((kp->g.et - lt) - kp->t0) - d);
                    sv0 = {fma(-turn, sp0.y, sp0.x), fma(turn, sp0.x, sp0.y), sp0.z};
                }
                // (the point's own motion: the spin about B0's z, and the drift of the pole - pm_geometry.WP, 1e-9 km/s)
                const V3 wp = cross(v3(kp->WPB[0], kp->WPB[1], kp->WPB[2]), sv0);
                const V3 vp = {fma(kp->ASB_state[0], dv, kp->VSB_state[0]) - kp->g.wdot * sv0.y + wp.x,
                               fma(kp->ASB_state[1], dv, kp->VSB_state[1]) + kp->g.wdot * sv0.x + wp.y, fma(kp->ASB_state[2], dv, kp->VSB_state[2]) + wp.z};
                const V3 vo = v3(kp->VOB[0], kp->VOB[1], kp->VOB[2]);
                const double dlt = (dot(u, vp - vo) * kp->inv_c) / (1.0 + dot(u, vp) * kp->inv_c);
                const double rv = dot((1.0 - dlt) * vp - vo, u) + miss;
                const double beta = rv * kp->inv_c;  // SpiceBase.calculate_doppler_factor base.py:550
                if (inside) {
                    PM_PUT_ROW(PM_DISTANCE, dist_lt);
                    PM_PUT_ROW(PM_RADIAL_VELOCITY, rv);
                    PM_PUT_ROW(PM_DOPPLER, sqrt_fast(div_fast(1.0 + beta, 1.0 - beta)));
                }
            }
        }
    }

    if (!stored && inside && PM_EXPERIMENT != 1) {
        // no intercept anywhere in this wave: every disc plane is NaN
        PM_PUT_ROW(PM_LON_GRAPHIC, nan);
        PM_PUT_ROW(PM_LAT_GRAPHIC, nan);
        PM_PUT_ROW(PM_LON_CENTRIC, nan);
        PM_PUT_ROW(PM_LAT_CENTRIC, nan);
        PM_PUT_ROW(PM_LOCAL_SOLAR_TIME, nan);
        if (FLAGS & DF_ILLUM) {
            PM_PUT_ROW(PM_PHASE, nan);
            PM_PUT_ROW(PM_INCIDENCE, nan);
            PM_PUT_ROW(PM_EMISSION, nan);
            PM_PUT_ROW(PM_AZIMUTH, nan);
        }
        if (FLAGS & DF_STATE) {
            PM_PUT_ROW(PM_DISTANCE, nan);
            PM_PUT_ROW(PM_RADIAL_VELOCITY, nan);
            PM_PUT_ROW(PM_DOPPLER, nan);
        }
    }

    if ((FLAGS & DF_RING) && inside) {
        // hidden behind the disc (NaN compares false): body_xy.py:4077-4080
        if (rd > dist_lt) rr = rl = rd = nan;
        PM_PUT_ROW(PM_RING_RADIUS, rr);
        PM_PUT_ROW(PM_RING_LON_GRAPHIC, rl);
        PM_PUT_ROW(PM_RING_DISTANCE, rd);
    }
}
#undef PM_MASK
#define PM_MASK p.mask

// Map-space chain: BodyXY._get_targvec_map body_xy.py:3227, _get_illumf_map :3667,
// _get_obsvec_map :3273, _get_radec_map :3419, _get_xy_map :3478 and the get_*_map
// planes. One lane per map location; lon/lat grids are read coalesced.
// SUN: phase / incidence / azimuth or the `lit` gate of the limb / ring planes are wanted;
// STATE: distance / radial velocity / doppler. The x/y-map request of a reprojection needs
// neither: only the emission angle (visibility) is evaluated then.
template <bool SUN, bool STATE>
__global__ __launch_bounds__(kBlock) void k_map(const Params p_, const double *__restrict__ lon_in,
                                                const double *__restrict__ lat_in)
{
    // constants through the laundered kernel-argument pointer: loaded where they are used (see
    // kernarg_params; after inlining the address space is inferred back, the loads stay scalar)
    const Params &p = *(const Params *)kernarg_params();
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

// Ring-plane and limb coordinates of an observer-frame direction given in B0 (`u`: unit vector), for the map kernel below:
// the arithmetic of the ring block of k_disc_sph and of the limb block of sky_block (derivations there), one lane at a time.
// Body._ring_coordinates_from_obsvec(only_visible=False) body.py:2577-2615
__device__ __forceinline__ void ring_coords_b0(const Params &p, const V3 u, double &radius, double &lon_deg, double &dist)
{
    radius = lon_deg = dist = __builtin_nan("");
    const double pd = dot(u, ld3(p.ring_nb));
    const double kk = p.g.ring_k;
    const bool ok = (kk == 0.0) ? (pd != 0.0) : (pd > 0.0 && kk < pd * (1.7976931348623157e308 / 3.0));
    if (!wave_any(ok)) return;
    // (lanes without an intersection carry a harmless finite point through the math)
    const double s = !ok ? p.g.sub_dist : ((kk == 0.0) ? 0.0 : div_fast(kk, pd));
    const V3 ob = {fma(s, u.x, -p.sub_obs_b[0]), fma(s, u.y, -p.sub_obs_b[1]), fma(s, u.z, -p.sub_obs_b[2])};
    const V3 w = ob - ld3(p.sub_ray_b);
    const double dd = sqrt_fast(dot(w, w)) - p.g.sub_dist;
    const double t = p.g.sub_et - dd * p.inv_c;
    double sa, ca;
    sincos_tiered<true>(p.g.wdot * (t - p.t0), sa, ca);
    const V3 tv = {fma(ca, ob.x, sa * ob.y) + p.g.sub_sp[0], fma(ca, ob.y, -sa * ob.x) + p.g.sub_sp[1], ob.z + p.g.sub_sp[2]};
    double le, alt;
    recpgr_alt_lon(p, tv, le, alt, ok);
    double l = p.g.west_positive ? -le : le;
    if (l < 0.0) l += kTwoPi;
    if (ok) {
        radius = alt + p.radii[0];
        lon_deg = l * kDeg;
        dist = s;
    }
}
// Body._limb_coordinates_from_obsvec body.py:2081-2110
__device__ __forceinline__ void limb_coords_b0(const Params &p, const V3 u, double &lon_deg, double &lat_deg, double &dist)
{
    const V3 o0 = v3(p.O0[0], p.O0[1], p.O0[2]);  // -R0 T0
    const double k = -div_fast(dot(o0, u), dot(u, u));
    const V3 nb = {fma(k, u.x, o0.x), fma(k, u.y, o0.y), fma(k, u.z, o0.z)};  // near point - T0, in B0
    const double nd = norm_f(nb);
    const V3 ob = {fma(k, u.x, -p.sub_obs_b[0]), fma(k, u.y, -p.sub_obs_b[1]), fma(k, u.z, -p.sub_obs_b[2])};
    const V3 w = ob - ld3(p.sub_ray_b);
    const double dd = norm_f(w) - p.g.sub_dist;
    const double t = p.g.sub_et - dd * p.inv_c;
    double sa, ca;
    sincos_tiered<true>(p.g.wdot * (t - p.t0), sa, ca);
    const V3 tv = {fma(ca, ob.x, sa * ob.y) + p.g.sub_sp[0], fma(ca, ob.y, -sa * ob.x) + p.g.sub_sp[1], ob.z + p.g.sub_sp[2]};
    const V3 X = {tv.x * p.ir[0], tv.y * p.ir[1], tv.z * p.ir[2]};
    const V3 sfc = rsqrt_fast(dot(X, X)) * tv;
    const double nx = sfc.x * p.limb_n[0], ny = sfc.y * p.limb_n[0], nz = sfc.z * p.limb_n[1];
    const double lat = atan2_fast(nz, sqrt_fast(fma(nx, nx, ny * ny)));
    double l = atan2_fast(sfc.y, sfc.x);
    if (p.g.west_positive) l = -l;
    if (l < 0.0) l += kTwoPi;
    lon_deg = l * kDeg;
    lat_deg = lat * kDeg;
    dist = nd - norm_f(sfc);
}

// ------------------------------------------------------------------ map-space planes in B0
// The same chain as k_map for the planes a map is usually asked for, evaluated like k_map_xy and the image kernels: in
// B0, the body-fixed frame frozen at t0 (every rotation a turn about z by the spin angle of the epoch), with the fast
// elementary functions, and only the groups of the chain that the requested planes need (the plane mask is a kernel
// argument: every `if (want...)` below is a scalar branch). k_map costs ~1400 vector instructions per cell whatever is
// asked - fine for the 64 800 cells of a 1 deg grid, which are latency, but a 0.05 deg map has 26 M cells: 0.68 ms for
// five planes against 0.13 ms for the five planes of a 4096^2 IMAGE, which solves an intercept on top.
//   * surface point: pgrrec_c at altitude 0 (body.py:903), as map_cell_xy;
//   * its light time: spkcpt_c / illumf_c 'CN' (body.py:1915, 2830) - three passes from the centre value and the state at
//     the epoch of the third, the sequence the oracle's point_lt<3> walks, epochs rounded as the reference rounds them;
//   * illumination: point, normal, Sun (two passes + the final evaluation, acceleration carried) and observer in B0;
//   * state: velocity with the light-time rate, as the STATE block of k_disc_sph;
//   * RA / Dec, pixel, km and angular coordinates: PM's own transform about the sub-observer point, as map_cell_xy;
//   * limb / ring planes: the B0 blocks of sky_block / k_disc_sph on the observer vector (ring_coords_b0, limb_coords_b0).
// PM_OPT_GENERAL_KERNEL keeps k_map, the independent J2000 evaluation, as the image planes keep theirs.
template <bool SUN, bool STATE>
__global__ __launch_bounds__(kBlock) void k_map_b0(const Params p_, const double *__restrict__ lon_in,
                                                   const double *__restrict__ lat_in)
{
    const Params &p = *(const Params *)kernarg_params();
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t n = (size_t)p.n0 * p.n1;
    if (idx >= n) return;
    const double nan = __builtin_nan("");
    double lon_deg = lon_in[idx], lat_deg = lat_in[idx];
    const bool have = isfinite(lon_deg) && isfinite(lat_deg);
    if (!have) lon_deg = lat_deg = nan;
    const double miss = have ? 0.0 : nan;  // added to a cell's values: NaN for a cell without coordinates
    PM_PUT(PM_LON_GRAPHIC, lon_deg);
    PM_PUT(PM_LAT_GRAPHIC, lat_deg);
    if (PM_WANT(PM_LOCAL_SOLAR_TIME)) PM_PUT(PM_LOCAL_SOLAR_TIME, local_solar_time(p, lon_deg));
    constexpr unsigned long long kCellBits = plane_bit(PM_LON_GRAPHIC) | plane_bit(PM_LAT_GRAPHIC) | plane_bit(PM_LOCAL_SOLAR_TIME);
    constexpr unsigned long long kCentricBits = plane_bit(PM_LON_CENTRIC) | plane_bit(PM_LAT_CENTRIC);
    constexpr unsigned long long kRadecBits = plane_bit(PM_RA) | plane_bit(PM_DEC);
    constexpr unsigned long long kXyBits = plane_bit(PM_PIXEL_X) | plane_bit(PM_PIXEL_Y) | plane_bit(PM_KM_X) | plane_bit(PM_KM_Y) |
                                           plane_bit(PM_ANGULAR_X) | plane_bit(PM_ANGULAR_Y);
    constexpr unsigned long long kLimbBits = plane_bit(PM_LIMB_LON_GRAPHIC) | plane_bit(PM_LIMB_LAT_GRAPHIC) | plane_bit(PM_LIMB_DISTANCE);
    constexpr unsigned long long kRingBits = plane_bit(PM_RING_RADIUS) | plane_bit(PM_RING_LON_GRAPHIC) | plane_bit(PM_RING_DISTANCE);
    if ((p.mask & ~kCellBits) == 0) return;

    // pgrrec_c, altitude 0: (a^2 cos(lat) cos(l), a^2 cos(lat) sin(l), c^2 sin(lat)) / sqrt(a^2 cos^2 + c^2 sin^2)
    const double lon = have ? lon_deg * kRad : 0.0, lat = have ? lat_deg * kRad : 0.0;
    const double a = p.radii[0], c = p.radii[2];
    double sl, cl, so, co;
    sincos_auto(lat, sl, cl);
    sincos_auto(p.g.west_positive ? -lon : lon, so, co);
    const double acl = a * cl, csl = c * sl;
    const double den = rsqrt_fast(fma(acl, acl, csl * csl));
    const double ha = a * acl * den;
    const V3 tv = {ha * co, ha * so, c * csl * den};
    if (p.mask & kCentricBits) {
        // reclat_c body.py:2905 (longitude 0 on the axis)
        const double bc = atan2_fast(tv.z, sqrt_fast(fma(tv.x, tv.x, tv.y * tv.y)));
        const double lc = atan2_fast<false, true>(tv.y, tv.x);
        PM_PUT(PM_LON_CENTRIC, fma(lc, kDeg, miss));
        PM_PUT(PM_LAT_CENTRIC, fma(bc, kDeg, miss));
    }
    if ((p.mask & ~(kCellBits | kCentricBits)) == 0) return;

    // light time of the point: pos(te) = T(te) + R(te)^T tv, in B0  w(d) = VB d + AB d^2 / 2 - O0 + Rz(wdot d)^T tv
    const double wdot = p.g.wdot;
    // (the first pass starts from the centre's light time: its epoch is t0 itself - d = (et - lt_c) - t0 = 0 exactly, host and
    //  device form t0 by the same subtraction - so the body has not turned and the target has not moved: the pass is the
    //  distance of the point as it stands, the same bits as the general expression below without its sincos and nine FMAs)
    double d = 0.0, sa = 0.0, ca = 1.0;
    V3 q = tv, w = {tv.x - p.O0[0], tv.y - p.O0[1], tv.z - p.O0[2]};
    double lt = sqrt_fast(dot(w, w)) * p.inv_c;
#pragma unroll
    for (int it = 1; it < 4; it++) {
        d = (p.g.et - lt) - p.t0;
        const double h = 0.5 * d * d;
        sincos_tiered<true>(wdot * d, sa, ca);
        q = {fma(ca, tv.x, -sa * tv.y), fma(sa, tv.x, ca * tv.y), tv.z};
        w = {fma(p.AB[0], h, fma(p.VB[0], d, q.x - p.O0[0])), fma(p.AB[1], h, fma(p.VB[1], d, q.y - p.O0[1])),
             fma(p.AB[2], h, fma(p.VB[2], d, q.z - p.O0[2]))};
        if (it < 3) lt = sqrt_fast(dot(w, w)) * p.inv_c;
    }
    const V3 u = rsqrt_fast(dot(w, w)) * w;  // observer -> point
    const V3 ob = neg(u);
    // surfnm_c in the body-fixed frame of the epoch, turned into B0 like the point
    const V3 nb = {tv.x * (p.ir[0] * p.ir[0]), tv.y * (p.ir[1] * p.ir[1]), tv.z * (p.ir[2] * p.ir[2])};
    V3 nrm = {fma(ca, nb.x, -sa * nb.y), fma(sa, nb.x, ca * nb.y), nb.z};
    nrm = rsqrt_fast(dot(nrm, nrm)) * nrm;
    const double em = vsep_fast(nrm, ob);
    const bool vis = have && em < kHalfPi;
    bool lit = false;
    PM_PUT(PM_EMISSION, fma(em, kDeg, miss));
    const double surf_dist = lt * p.g.clight + miss;
    PM_PUT(PM_DISTANCE, surf_dist);
    if (SUN) {
        // illumf_c: the point wrt P_T(t0); the Sun at te - |S - q| / c, two passes from the centre value + the final one
        const double h = 0.5 * d * d;
        const V3 qi = {fma(p.AB[0], h, fma(p.VB[0], d, q.x)), fma(p.AB[1], h, fma(p.VB[1], d, q.y)), fma(p.AB[2], h, fma(p.VB[2], d, q.z))};
        const double dts = (p.t0 - p.g.ts0) + d;  // te - ts0
        double ds = 0.0;
        V3 sv = {0.0, 0.0, 0.0};
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const double hs = 0.5 * ds * ds;
            sv = {fma(p.ASB[0], hs, fma(p.VSB[0], ds, p.SB0[0])) - qi.x, fma(p.ASB[1], hs, fma(p.VSB[1], ds, p.SB0[1])) - qi.y,
                  fma(p.ASB[2], hs, fma(p.VSB[2], ds, p.SB0[2])) - qi.z};
            if (it < 2) ds = dts - norm_f(sv) * p.inv_c;
        }
        const V3 sunb = rsqrt_fast(dot(sv, sv)) * sv;
        const double ph = vsep_fast(sunb, ob), in = vsep_fast(nrm, sunb);
        lit = have && in < kHalfPi;
        PM_PUT(PM_PHASE, fma(ph, kDeg, miss));
        PM_PUT(PM_INCIDENCE, fma(in, kDeg, miss));
        if (PM_WANT(PM_AZIMUTH)) PM_PUT(PM_AZIMUTH, azimuth_from_cosines(dot(sunb, ob), dot(nrm, sunb), dot(nrm, ob)) + miss);
    }
    if (STATE) {
        // spkcpt_c's velocity with the light-time rate (body.py:2830-2850), as the STATE block of k_disc_sph
        const V3 wp = cross(v3(p.WPB[0], p.WPB[1], p.WPB[2]), q);  // (the drift of the pole: pm_geometry.WP)
        const V3 vp = {fma(p.ASB_state[0], d, p.VSB_state[0]) - wdot * q.y + wp.x, fma(p.ASB_state[1], d, p.VSB_state[1]) + wdot * q.x + wp.y,
                       fma(p.ASB_state[2], d, p.VSB_state[2]) + wp.z};
        const V3 vo = {p.VOB[0], p.VOB[1], p.VOB[2]};
        const double dlt = div_fast(dot(u, vp - vo) * p.inv_c, fma(dot(u, vp), p.inv_c, 1.0));
        const double rv = dot((1.0 - dlt) * vp - vo, u) + miss;
        const double beta = rv * p.inv_c;  // SpiceBase.calculate_doppler_factor base.py:550
        PM_PUT(PM_RADIAL_VELOCITY, rv);
        PM_PUT(PM_DOPPLER, sqrt_fast(div_fast(1.0 + beta, 1.0 - beta)));
    }
    if ((p.mask & (kRadecBits | kXyBits | kLimbBits | kRingBits)) == 0) return;

    // Body._targvec2obsvec (body.py:917-948): the offset from the sub-observer point, turned at ITS light-time epoch
    const V3 off = {tv.x - p.g.sub_sp[0], tv.y - p.g.sub_sp[1], tv.z - p.g.sub_sp[2]};
    const V3 sr = {p.g.sub_ray[0] + off.x, p.g.sub_ray[1] + off.y, p.g.sub_ray[2] + off.z};
    const double dist = sqrt_fast(dot(sr, sr)) - p.g.sub_dist;
    const double t = p.g.sub_et - dist * p.inv_c;
    double s2, c2;
    sincos_tiered<true>(wdot * (t - p.t0), s2, c2);
    // R0 ov = R0 sub_obsvec + Rz(ang2)^T off
    const V3 b = {p.sub_obs_b[0] + fma(c2, off.x, -s2 * off.y), p.sub_obs_b[1] + fma(s2, off.x, c2 * off.y), p.sub_obs_b[2] + off.z};
    const double hide = vis ? 0.0 : nan;  // RA / Dec and what follows from them: visible cells only (body_xy.py:3430)
    if (p.mask & kRadecBits) {
        double ra, dec;
        recrad_f(mtxv(p.g.R0, b), ra, dec);
        PM_PUT(PM_RA, fma(ra, kDeg, hide));
        PM_PUT(PM_DEC, fma(dec, kDeg, hide));
    }
    if (p.mask & kXyBits) {
        // Body._obsvec2angular: M ov = C^T (R0 ov); recrad_c is scale free
        const V3 m = {fma(p.C[0], b.x, fma(p.C[3], b.y, p.C[6] * b.z)), fma(p.C[1], b.x, fma(p.C[4], b.y, p.C[7] * b.z)),
                      fma(p.C[2], b.x, fma(p.C[5], b.y, p.C[8] * b.z))};
        double ra, dec;
        recrad_f(m, ra, dec);
        double xx = -(ra * kDeg);
        if (xx < 0.0) xx += 360.0;
        if (xx > 180.0) xx -= 360.0;
        const double ax = xx * 3600.0, ay = (dec * kDeg) * 3600.0;
        const double px = fma(p.Ai[0], ax, fma(p.Ai[1], ay, p.Ai[2]));
        const double py = fma(p.Ai[3], ax, fma(p.Ai[4], ay, p.Ai[5]));
        // BodyXY._xy_in_image_frame body_xy.py:1868
        const bool in_frame = vis && -0.5 < px && px < p.nx - 0.5 && -0.5 < py && py < p.ny - 0.5;
        PM_PUT(PM_PIXEL_X, in_frame ? px : nan);
        PM_PUT(PM_PIXEL_Y, in_frame ? py : nan);
        const double kx = fma(p.K[0], ax, p.K[1] * ay) + hide, ky = fma(p.K[2], ax, p.K[3] * ay) + hide;
        PM_PUT(PM_KM_X, kx);
        PM_PUT(PM_KM_Y, ky);
        if (PM_WANT(PM_ANGULAR_X) || PM_WANT(PM_ANGULAR_Y)) {
            const double ik = rcp_fast(p.g.km_per_arcsec);
            const double qx = kx * ik, qy = ky * ik;
            PM_PUT(PM_ANGULAR_X, fma(fma(-p.g.km_per_arcsec, qx, kx), ik, qx));
            PM_PUT(PM_ANGULAR_Y, fma(fma(-p.g.km_per_arcsec, qy, ky), ik, qy));
        }
    }
    if (SUN && (p.mask & (kLimbBits | kRingBits))) {
        // gated on illumf column 4 (lit) like the reference (body_xy.py:3981, 4097)
        double ll = nan, lb = nan, ld = nan, rr = nan, rl = nan, rd = nan;
        if (lit) {
            const V3 ub = rsqrt_fast(dot(b, b)) * b;
            if (p.mask & kLimbBits) limb_coords_b0(p, ub, ll, lb, ld);
            if (p.mask & kRingBits) ring_coords_b0(p, ub, rr, rl, rd);
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

// ------------------------------------------------------------------ x/y map alone
// What a reprojection needs of the map chain: pixel coordinates of the visible grid cells,
// nothing else (BodyXY._get_xy_map body_xy.py:3478 and its inputs :3227-3300, 3419-3491, 3667).
// k_map above evaluates that chain with the general helpers - 3 x 3 rotations rebuilt at every
// light-time pass, four passes, libm sincos / asin, IEEE divisions, ~1400 vector instructions per
// cell, and one wave per SIMD for 9.5 us in every step of the frame benchmark. This kernel
// evaluates the SAME quantities in B0, the body-fixed frame frozen at t0 (R(t) = Rz(wdot (t - t0))
// R0, so every rotation is a turn about z by a small angle), with the fast elementary functions:
//   * surface point: pgrrec_c at altitude 0 (body.py:903);
//   * visibility (illumf_c's `visibl`, emission < 90 deg, body.py:1925): the sign of
//     n . (observer - point) - the emission angle itself is not an output here. The observer
//     direction needs the light time of the point only to 1e-13 rad: two passes (the general code's
//     four agree with them far below the rounding of the test);
//   * PM's own body-fixed -> observer transform (Body._targvec2obsvec body.py:917-948), which
//     ignores the target's translation by design;
//   * RA/Dec -> angular -> pixel (body.py:1345, body_xy.py:379, 1868). The reference passes
//     RA/Dec through degrees and back to a unit vector (body_xy.py:3430-3488); that round trip moves
//     the direction by < 1 ulp and is not replayed (same choice as k_sky / the ring block).
// Constants are read through a laundered kernel-argument pointer where they are used (no scalar
// registers parked in VGPR lanes: k_map spills 108-160 of them), 64-thread workgroups spread the
// 64 800 cells of a 1 deg map over all 1024 SIMDs.
__global__ __launch_bounds__(kSphBlock) void k_map_xy(const Params p, const double *__restrict__ lon_in,
                                                      const double *__restrict__ lat_in)
{
    const KParams kp = kernarg_params();
    const size_t idx = (size_t)blockIdx.x * kSphBlock + threadIdx.x;
    const size_t n = (size_t)kp->n0 * kp->n1;
    if (idx >= n) return;
    double px, py;
    map_cell_xy(kp, lon_in[idx], lat_in[idx], px, py);
    if ((kp->mask >> PM_PIXEL_X) & 1ull) kp->out[PM_PIXEL_X][idx] = px;
    if ((kp->mask >> PM_PIXEL_Y) & 1ull) kp->out[PM_PIXEL_Y][idx] = py;
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

__global__ __launch_bounds__(kBlock) void k_transform(const Params p_, const TransformArgs t)
{
    const Params &p = *(const Params *)kernarg_params();  // constants loaded where they are used (k_map)
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
        for (int k = 0; k < 3; k++) {
            p0.radii[k] = t.radii0[k];
            p0.ir[k] = 1.0 / t.radii0[k];  // (the helpers scale by the reciprocals)
        }
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

// Sky points -> everything the per-pixel loops derive from a ray, for caller-supplied RA/Dec:
// Body.radec2lonlat body.py:1083, Body.ring_plane_coordinates body.py:2617-2658 (with its
// only_visible rule :2598-2611), Body.limb_coordinates_from_radec body.py:2040-2110.
// out: 8 arrays of n doubles = lon, lat, ring radius, ring lon, ring distance, limb lon,
// limb lat, limb distance.
__global__ __launch_bounds__(kBlock) void k_radec_query(const Params p_, const double *__restrict__ ra_deg,
                                                        const double *__restrict__ dec_deg, unsigned long long n,
                                                        int ring_only_visible, double *__restrict__ out)
{
    const Params &p = *(const Params *)kernarg_params();  // constants loaded where they are used (k_map)
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double nan = __builtin_nan("");
    double o[8] = {nan, nan, nan, nan, nan, nan, nan, nan};
    const double ra = ra_deg[i] * kRad, dec = dec_deg[i] * kRad;
    if (isfinite(ra) && isfinite(dec)) {  // body.py:964-967
        const V3 ray = radrec(ra, dec);
        V3 sp = {nan, nan, nan};
        double lt = 0.0;
        const bool hit = sincpt(p, ray, sp, lt);
        if (hit) {
            double lon, lat;
            recpgr_surface(p, sp, lon, lat);
            o[0] = lon * kDeg;
            o[1] = lat * kDeg;
        }
        double rr, rl, rd;
        ring_coords(p, ray, rr, rl, rd);
        if (ring_only_visible && !isnan(rr)) {
            if (rr - p.radii[0] < 0.0) {
                rr = rl = rd = nan;  // inside the planet
            } else if (hit) {
                V3 pos;
                M3 R;
                point_lt<1>(p, sp, lt, pos, R);
                if (lt * p.g.clight < rd) rr = rl = rd = nan;  // behind the disc
            }
        }
        o[2] = rr;
        o[3] = rl;
        o[4] = rd;
        limb_coords(p, ray, o[5], o[6], o[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) out[(size_t)k * n + i] = o[k];
}

// The same for a spheroid seen from outside (every planet; the predicate of the image kernels' fast path, pm_capi.hip), in
// the B0 formulation of k_disc_sph / sky_block / k_map_b0: the ray turned into B0 once (u = R0 ray), sincpt_c 'CN' as the
// reference's own sequence of light-time epochs in the scaled frame of surfpt_c - the body's spin leaves a spheroid where
// it is, the target's motion is Y(d) = O0s - VBs d, no rotation per pass - longitude and latitude from the scaled
// intercept, the ring and limb blocks on the same u (ring_coords_b0 / limb_coords_b0). k_radec_query above, the J2000
// evaluation with a 3 x 3 rotation per light-time pass and recpgr_c by iteration, serves every other body and
// PM_OPT_GENERAL_KERNEL (16.7 M points: 2.78 ms).
__global__ __launch_bounds__(kBlock) void k_radec_query_b0(const Params p_, const double *__restrict__ ra_deg,
                                                           const double *__restrict__ dec_deg, unsigned long long n,
                                                           int ring_only_visible, double *__restrict__ out)
{
    const Params &p = *(const Params *)kernarg_params();  // constants loaded where they are used (k_map)
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double nan = __builtin_nan("");
    double o[8] = {nan, nan, nan, nan, nan, nan, nan, nan};
    const double ra = ra_deg[i] * kRad, dec = dec_deg[i] * kRad;
    const bool given = isfinite(ra) && isfinite(dec);  // body.py:964-967
    // (a lane without a point carries a harmless ray through the wave-level helpers)
    const V3 ray = given ? radrec(ra, dec) : (1.0 / (p.g.lt_c * p.g.clight)) * v3(p.g.T0[0], p.g.T0[1], p.g.T0[2]);
    const V3 u = mxv(p.g.R0, ray);
    // sincpt_c 'CN' (body.py:1008-1020): the intercept repeated at te = et - lt until the light time moves by
    // <= 1e-17 |te|, at most 10 passes; no intercept in any pass -> not found
    const V3 X = {u.x * p.ir[0], u.y * p.ir[1], u.z * p.ir[2]};
    const double ixx = rcp_fast(dot(X, X));
    double lt = p.g.lt_c, d = 0.0, k = 0.0, sroot = 0.0;
    V3 P = {0.0, 0.0, 0.0}, Y = {0.0, 0.0, 0.0};
    bool hit = given;
#pragma unroll 1
    for (int it = 0; it < 10 && hit; it++) {
        d = (p.g.et - lt) - p.t0;  // two roundings, as the epoch et - lt of the reference has them (first pass: 0 exactly)
        Y = {fma(-p.VBs[0], d, p.O0s[0]), fma(-p.VBs[1], d, p.O0s[1]), fma(-p.VBs[2], d, p.O0s[2])};
        const double yx = dot(Y, X);
        k = yx * ixx;
        P = {fma(-k, X.x, Y.x), fma(-k, X.y, Y.y), fma(-k, X.z, Y.z)};
        const double p2 = dot(P, P);
        if (p2 > 1.0 || yx > 0.0) {  // (the observer is outside: |Y| > 2, host)
            hit = false;
            break;
        }
        sroot = -sqrt_fast(fmax(0.0, 1.0 - p2) * ixx);
        const double nlt = (sroot - k) * p.inv_c;  // |u| = 1
        const double err = fabs(nlt - lt);
        lt = nlt;
        if (err <= 1e-17 * fabs(p.g.et - lt)) break;
    }
    if (hit) {
        const V3 Xf = {fma(sroot, X.x, P.x), fma(sroot, X.y, P.y), fma(sroot, X.z, P.z)};
        // recpgr_c: east longitude in the frame at te = B0 longitude - wdot d, sign by the body's convention
        const double theta = (Xf.x == 0.0 && Xf.y == 0.0) ? 0.0 : atan2_fast(Xf.y, Xf.x);
        double l = fma(-p.lon_k[1], d, p.lon_k[0] * theta);
        if (l < 0.0) l += kTwoPi;
        const double rho = sqrt_fast(fma(Xf.x, Xf.x, Xf.y * Xf.y));
        o[0] = l * kDeg;
        o[1] = atan2_fast<true>(Xf.z * p.a_over_c, rho) * kDeg;
    }
    double rr, rl, rd;
    ring_coords_b0(p, u, rr, rl, rd);
    if (ring_only_visible && !isnan(rr)) {
        if (rr - p.radii[0] < 0.0) {
            rr = rl = rd = nan;  // inside the planet
        } else if (hit) {
            // the intercept's distance with one more pass of its light time (point_lt<1>: the point as it stands - the
            // turn of the body over the 1e-9 s the light time still moves by is 1e-8 km)
            const double d1 = (p.g.et - lt) - p.t0;
            const V3 w = {(fma(sroot, X.x, P.x) - fma(-p.VBs[0], d1, p.O0s[0])) * p.radii[0],
                          (fma(sroot, X.y, P.y) - fma(-p.VBs[1], d1, p.O0s[1])) * p.radii[1],
                          (fma(sroot, X.z, P.z) - fma(-p.VBs[2], d1, p.O0s[2])) * p.radii[2]};
            if (norm_f(w) < rd) rr = rl = rd = nan;  // behind the disc
        }
    }
    o[2] = rr;
    o[3] = rl;
    o[4] = rd;
    limb_coords_b0(p, u, o[5], o[6], o[7]);
    if (!given) {
#pragma unroll
        for (int q = 0; q < 8; q++) o[q] = nan;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) out[(size_t)q * n + i] = o[q];
}

}  // namespace pm

// ------------------------------------------------------------------ launchers (called from pm_capi.hip)
extern "C++" {

// the general image kernel: k_disc_sph<FLAGS, 2> (near-field observers, fast spinners, large accelerations,
// PM_OPT_GENERAL_KERNEL)
void pm_launch_disc(const pm::Params &p, int flags, hipStream_t s)
{
    dim3 grid((p.nx + pm::kSphBlock - 1) / pm::kSphBlock, p.rows);
    dim3 block(pm::kSphBlock);
    switch (flags & 7) {
    case 0: hipLaunchKernelGGL((pm::k_disc_sph<0, 2>), grid, block, 0, s, p); break;
    case 1: hipLaunchKernelGGL((pm::k_disc_sph<1, 2>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((pm::k_disc_sph<2, 2>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((pm::k_disc_sph<3, 2>), grid, block, 0, s, p); break;
    case 4: hipLaunchKernelGGL((pm::k_disc_sph<4, 2>), grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL((pm::k_disc_sph<5, 2>), grid, block, 0, s, p); break;
    case 6: hipLaunchKernelGGL((pm::k_disc_sph<6, 2>), grid, block, 0, s, p); break;
    case 7: hipLaunchKernelGGL((pm::k_disc_sph<7, 2>), grid, block, 0, s, p); break;
    }
}

// flags: DiscFlags in bits 0..2; bits 3..4: 0 = intercept planes only, 1 / 2 = + the sky planes (/ + limb planes)
void pm_launch_disc_spheroid(const pm::Params &p, int flags, hipStream_t s)
{
    dim3 grid((p.nx + pm::kSphBlock - 1) / pm::kSphBlock, p.rows);
    dim3 block(pm::kSphBlock);
    const bool tri = p.radii[0] != p.radii[1];
    const int sky = (flags >> 3) & 3;
    // (the QUANT variants exist without the fused sky planes: the caller does not fuse for such geometries)
    const bool quant = sky == 0 && (p.cf_iter != 0 || p.turn_quantum != 0);
#define PM_SPH_CASE(F)                                                                                          \
    case F:                                                                                                     \
        if (quant && tri) hipLaunchKernelGGL((pm::k_disc_sph<F, 1, 0, 0, true>), grid, block, 0, s, p);         \
        else if (quant && F == 1 && p.mask == pm::kMaskHeadline)                                                 \
            hipLaunchKernelGGL((pm::k_disc_sph<1, 0, 0, pm::kMaskHeadline, true>), grid, block, 0, s, p);       \
        else if (quant && F == 5 && p.mask == pm::kMaskRings)                                                    \
            hipLaunchKernelGGL((pm::k_disc_sph<5, 0, 0, pm::kMaskRings, true>), grid, block, 0, s, p);          \
        else if (quant && F == 7 && p.mask == pm::kMaskDisc)                                                     \
            hipLaunchKernelGGL((pm::k_disc_sph<7, 0, 0, pm::kMaskDisc, true>), grid, block, 0, s, p);           \
        else if (quant) hipLaunchKernelGGL((pm::k_disc_sph<F, 0, 0, 0, true>), grid, block, 0, s, p);           \
        else if (tri && sky == 0) hipLaunchKernelGGL((pm::k_disc_sph<F, 1, 0>), grid, block, 0, s, p);          \
        else if (tri && sky == 1) hipLaunchKernelGGL((pm::k_disc_sph<F, 1, 1>), grid, block, 0, s, p);          \
        else if (tri) hipLaunchKernelGGL((pm::k_disc_sph<F, 1, 2>), grid, block, 0, s, p);                      \
        else if (sky == 0 && F == 1 && p.mask == pm::kMaskHeadline)                                              \
            hipLaunchKernelGGL((pm::k_disc_sph<1, 0, 0, pm::kMaskHeadline>), grid, block, 0, s, p);             \
        else if (sky == 0 && F == 5 && p.mask == pm::kMaskRings)                                                 \
            hipLaunchKernelGGL((pm::k_disc_sph<5, 0, 0, pm::kMaskRings>), grid, block, 0, s, p);                \
        else if (sky == 0 && F == 7 && p.mask == pm::kMaskDisc)                                                  \
            hipLaunchKernelGGL((pm::k_disc_sph<7, 0, 0, pm::kMaskDisc>), grid, block, 0, s, p);                 \
        else if (sky == 0) hipLaunchKernelGGL((pm::k_disc_sph<F, 0, 0>), grid, block, 0, s, p);                 \
        else if (sky == 1) hipLaunchKernelGGL((pm::k_disc_sph<F, 0, 1>), grid, block, 0, s, p);                 \
        else hipLaunchKernelGGL((pm::k_disc_sph<F, 0, 2>), grid, block, 0, s, p);                               \
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

void pm_launch_radec_query(const pm::Params &p, const double *ra, const double *dec, unsigned long long n,
                           int ring_only_visible, double *out, bool b0, hipStream_t s)
{
    unsigned long long blocks = (n + pm::kBlock - 1) / pm::kBlock;
    if (b0)
        hipLaunchKernelGGL(pm::k_radec_query_b0, dim3((unsigned)blocks), dim3(pm::kBlock), 0, s, p, ra, dec, n, ring_only_visible, out);
    else
        hipLaunchKernelGGL(pm::k_radec_query, dim3((unsigned)blocks), dim3(pm::kBlock), 0, s, p, ra, dec, n, ring_only_visible, out);
}

void pm_launch_map_xy(const pm::Params &p, const double *lon, const double *lat, hipStream_t s)
{
    const size_t n = (size_t)p.n0 * p.n1;
    hipLaunchKernelGGL(pm::k_map_xy, dim3((unsigned)((n + pm::kSphBlock - 1) / pm::kSphBlock)), dim3(pm::kSphBlock), 0, s, p, lon,
                       lat);
}

// general: k_map, the J2000 evaluation with the general helpers (PM_OPT_GENERAL_KERNEL); otherwise k_map_b0
void pm_launch_map(const pm::Params &p, const double *lon, const double *lat, bool general, hipStream_t s)
{
    size_t n = (size_t)p.n0 * p.n1;
    dim3 grid((unsigned)((n + pm::kBlock - 1) / pm::kBlock));
    const uint64_t sun_bits = PM_PLANE_BIT(PM_PHASE) | PM_PLANE_BIT(PM_INCIDENCE) | PM_PLANE_BIT(PM_AZIMUTH) |
                              PM_PLANE_BIT(PM_LIMB_LON_GRAPHIC) | PM_PLANE_BIT(PM_LIMB_LAT_GRAPHIC) |
                              PM_PLANE_BIT(PM_LIMB_DISTANCE) | PM_PLANE_BIT(PM_RING_RADIUS) |
                              PM_PLANE_BIT(PM_RING_LON_GRAPHIC) | PM_PLANE_BIT(PM_RING_DISTANCE);
    const uint64_t state_bits = PM_PLANE_BIT(PM_RADIAL_VELOCITY) | PM_PLANE_BIT(PM_DOPPLER);
    const bool sun = (p.mask & sun_bits) != 0, state = (p.mask & state_bits) != 0;
    if (!general) {
        if (sun && state) hipLaunchKernelGGL((pm::k_map_b0<true, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
        else if (sun) hipLaunchKernelGGL((pm::k_map_b0<true, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
        else if (state) hipLaunchKernelGGL((pm::k_map_b0<false, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
        else hipLaunchKernelGGL((pm::k_map_b0<false, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
        return;
    }
    if (sun && state) hipLaunchKernelGGL((pm::k_map<true, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else if (sun) hipLaunchKernelGGL((pm::k_map<true, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else if (state) hipLaunchKernelGGL((pm::k_map<false, true>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
    else hipLaunchKernelGGL((pm::k_map<false, false>), grid, dim3(pm::kBlock), 0, s, p, lon, lat);
}
}
