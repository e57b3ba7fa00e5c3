// pm_fastmath.hip.h -- branch-free binary64 elementary functions for the pixel kernels.
//
// The hot kernel is FP64-VALU bound (MI355X: 16 FP64 FMA lanes / clk / SIMD), so the
// general-purpose libm entry points (range reduction for huge arguments, special
// cases, double-double tails) are replaced where the argument range is known:
//
//   asin_half(x)    |x| <= 0.5            12-term odd polynomial, max abs err 1.1e-16
//   atan_small(t)   |t| <= tan(pi/8)      10-term odd polynomial, max abs err 1.1e-16
//   atan2_fast      any finite (y, x) != (0, 0): ONE division + atan_small
//   sincos_small    |x| <= 0.25           4-term tails, max abs err 1.1e-16
//   sincos_medium   |x| <= 1e5            3-part pi/2 reduction + 6-term tails, 2.1e-16
//
// Coefficients: tools/gen_poly.py (Chebyshev interpolation at 60 digits with mpmath,
// rounded to binary64 and re-verified in float64 Horner form).
#pragma once

#include <hip/hip_runtime.h>

namespace pm {

// Wave votes on the lane mask itself. HIP's __any / __all take an int: the condition is first written to a
// vector register as 0 / 1 and compared again (two VALU instructions per vote); a ballot leaves the mask in
// a scalar register pair, and the test is one scalar compare.
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
__device__ __forceinline__ bool wave_all(bool c) { return __builtin_amdgcn_ballot_w64(!c) == 0; }

constexpr double kPiF = 3.14159265358979323846;
constexpr double kTanPi8 = 0.41421356237309503;

// ---- reciprocal / division / square roots without the libm range scaffolding ----------
// v_rcp_f64 / v_rsq_f64 deliver ~26 good bits; Newton / Goldschmidt steps as in the
// compiler's own expansion, minus the exponent rescaling and class checks that only
// matter for subnormal / huge operands (never the case here: km, seconds, unit vectors).

// 1 / b, |b| in the normal range
__device__ __forceinline__ double rcp_fast(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
// a / b to ~1 ulp. One Newton step on the reciprocal is enough here (2^-46 relative): the
// residual correction of the quotient squares that error again.
__device__ __forceinline__ double div_fast(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// sqrt(x), x >= 0 in the normal range. x == 0 needs no select: the seed is taken from
// max(x, 1e-300), so g = 0 * finite = 0 and every correction term stays 0.
__device__ __forceinline__ double sqrt_fast(double x)
{
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);  // 1.5 e0^2 = 2^-45 relative from the 2^-23 seed
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);  // exact residual: error squared again, <= 1 ulp
}
// the same for x >= 1e-300 (no guard of the seed)
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}
// sqrt_pos that also hands out h = 1 / (2 sqrt(x)) to 2^-45 relative (it is computed anyway)
__device__ __forceinline__ double sqrt_pos_h(double x, double &h)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}
// sqrt(x), x >= 1e-300, to 2^-45 relative (sqrt_pos without its closing residual step)
__device__ __forceinline__ double sqrt_seed_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    return fma(g, fma(-0.5 * y, g, 0.5), g);
}
// 1 / sqrt(x), x > 0 in the normal range
__device__ __forceinline__ double rsqrt_fast(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);  // third order: e0^3 = 2^-69, ~1 ulp after rounding
}

// q * z + C with the constant C as a scalar (SGPR) operand of ONE v_fma_f64. Left to
// itself hipcc emits the two-address v_fmac_f64, which needs C copied into a VGPR pair
// first (two v_mov_b32 per Horner step, tripling the VALU cost of a polynomial); the
// s_mov pairs this form needs instead issue on the scalar unit, beside the VALU.
__device__ __forceinline__ double fma_c(double q, double z, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(z), "s"(c));
    return r;
}

// -q * z + C, q * z - C, z * C and z + C with C as the scalar operand of one instruction (see fma_c)
__device__ __forceinline__ double fnma_c(double q, double z, double c)
{
    double r;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(q), "v"(z), "s"(c));
    return r;
}
__device__ __forceinline__ double fma_cn(double q, double z, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(r) : "v"(q), "v"(z), "s"(c));
    return r;
}
__device__ __forceinline__ double mul_c(double z, double c)
{
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(z), "s"(c));
    return r;
}
__device__ __forceinline__ double add_c(double z, double c)
{
    double r;
    asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(z), "s"(c));
    return r;
}
// C - z (an explicit instruction also keeps the compiler from contracting it with a product behind z into an
// FMA that would hold two scalars)
__device__ __forceinline__ double rsub_c(double z, double c)
{
    double r;
    asm("v_add_f64 %0, %2, -%1" : "=v"(r) : "v"(z), "s"(c));
    return r;
}
// C - 2 r in one instruction (-2.0 is an inline constant, C the scalar operand)
__device__ __forceinline__ double fma_m2_c(double r, double c)
{
    double o;
    asm("v_fma_f64 %0, %1, -2.0, %2" : "=v"(o) : "v"(r), "s"(c));
    return o;
}
// The first Horner step c0 z + c1 of a polynomial with constant coefficients holds two scalars, and one
// instruction reads one: as an FMA the leading coefficient is first copied into a vector register pair (two
// v_mov_b32), as a product and a sum it is two FP64 operations and no copy. (The highest coefficient carries
// 1e-10 of the result: the extra rounding is far below the last place.)
__device__ __forceinline__ double horner_head(double z, double c0, double c1) { return add_c(mul_c(z, c0), c1); }

// asin(x) for |x| <= 0.5
__device__ __forceinline__ double asin_half(double x)
{
    const double z = x * x;
    double q = horner_head(z, 0.028169218060881414, -0.010749050339697808);
    q = fma_c(q, z, 0.01603551434914882);
    q = fma_c(q, z, 0.0078029494773533175);
    q = fma_c(q, z, 0.011875494382636922);
    q = fma_c(q, z, 0.013929652902326633);
    q = fma_c(q, z, 0.017355259955786323);
    q = fma_c(q, z, 0.02237204763174451);
    q = fma_c(q, z, 0.03038194736709848);
    q = fma_c(q, z, 0.044642857103423646);
    q = fma_c(q, z, 0.07500000000020764);
    q = fma_c(q, z, 0.1666666666666665);
    return fma(x * z, q, x);
}

// atan(t) for |t| <= tan(pi/8)
__device__ __forceinline__ double atan_small(double t)
{
    const double z = t * t;
    double q = horner_head(z, 0.02275052699336167, -0.04483334622272886);
    q = fma_c(q, z, 0.05736332165907643);
    q = fma_c(q, z, -0.06649613695291669);
    q = fma_c(q, z, 0.0769105515839315);
    q = fma_c(q, z, -0.09090852557176049);
    q = fma_c(q, z, 0.11111109636534361);
    q = fma_c(q, z, -0.1428571426609662);
    q = fma_c(q, z, 0.19999999999898407);
    q = fma_c(q, z, -0.3333333333333325);
    return fma(t * z, q, t);
}

// atan2(y, x) in (-pi, pi]; (y, x) finite and not both zero. XPOS: x >= 0 is known.
// atan(mn/mx) with mn <= mx is reduced by atan(q) = pi/4 + atan((q - 1)/(q + 1)) for
// q > tan(pi/8); numerator and denominator are chosen BEFORE dividing, so the whole
// function costs one division. max / min of the magnitudes take the |.| operand modifiers
// directly (through fmax(fabs()) hipcc first canonicalises each operand with an extra
// v_max_f64), and the reduction is blended in with a 0/1 factor: two FMAs instead of two
// differences and four selects.
// ZERO_OK: (0, 0) gives 0 (recpgr_c's longitude of a point on the axis) instead of NaN: the denominator is kept
// off zero, 0 / 1e-300 = 0 goes through the polynomial and the selects as any other first-octant angle.
template <bool XPOS = false, bool ZERO_OK = false>
__device__ __forceinline__ double atan2_fast(double y, double x)
{
    double mx, mn;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
    const bool big = mn > kTanPi8 * mx;
    const double bf = big ? 1.0 : 0.0;
    const double num = fma(-bf, mx, mn);
    double den = fma(bf, mn, mx);
    if (ZERO_OK) den = fmax(den, 1e-300);
    double a = atan_small(div_fast(num, den));
    a = big ? a + 0.25 * kPiF : a;
    a = (fabs(y) > fabs(x)) ? 0.5 * kPiF - a : a;
    if (!XPOS) a = (x < 0.0) ? kPiF - a : a;
    return copysign(a, y);
}

// sin and cos for |x| <= 0.25
__device__ __forceinline__ void sincos_small(double x, double &s, double &c)
{
    const double z = x * x;
    double qs = 2.7526021333363912e-06;
    qs = fma_c(qs, z, -0.00019841257617613103);
    qs = fma_c(qs, z, 0.008333333331805479);
    qs = fma_c(qs, z, -0.1666666666666637);
    s = fma(x * z, qs, x);
    double qc = -2.753123559604959e-07;
    qc = fma_c(qc, z, 2.4801577114157226e-05);
    qc = fma_c(qc, z, -0.0013888888887615533);
    qc = fma_c(qc, z, 0.041666666666666415);
    c = fma(z * z, qc, fma(-0.5, z, 1.0));
}

// sin and cos for |x| <= 1e-3 (the view angles of a planetary frame, a host-side decision):
// the next terms, x^5 / 120 and x^6 / 720, are below 1e-17 absolute
__device__ __forceinline__ void sincos_tiny(double x, double &s, double &c)
{
    const double z = x * x;
    s = x * fma(z, -1.0 / 6.0, 1.0);
    c = fma(z, fma(z, 1.0 / 24.0, -0.5), 1.0);
}

// sin and cos for |x| <= 1e5: Cody-Waite reduction by pi/2 in three parts (k = rint(2x/pi)
// is exact below 2^20 and k * PIO2_HI, k * PIO2_MID are exact products), then degree-5 tails
// on |r| <= pi/4 (max abs err 1.1e-16 each) and the quadrant swap. Used for RA/Dec-sized
// angles; anything larger goes to libm.
__device__ __forceinline__ void sincos_medium(double x, double &s, double &c)
{
    const double kf = rint(x * 0.63661977236758134308);  // 2 / pi
    // pi/2 = HI + MID + LO with HI, MID carrying 33 bits each (fdlibm's pio2_1, pio2_2, pio2_3)
    double r = fma(-kf, 1.57079632673412561417e+00, x);
    r = fma(-kf, 6.07710050630396597660e-11, r);
    r = fma(-kf, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double qs = 1.5918129294866608e-10;
    qs = fma_c(qs, z, -2.5051131845003624e-08);
    qs = fma_c(qs, z, 2.755731610255244e-06);
    qs = fma_c(qs, z, -0.00019841269836758574);
    qs = fma_c(qs, z, 0.008333333333330948);
    qs = fma_c(qs, z, -0.16666666666666666);
    const double sr = fma(r * z, qs, r);
    double qc = -1.1382632425521717e-11;
    qc = fma_c(qc, z, 2.08761462684032e-09);
    qc = fma_c(qc, z, -2.7557317271729793e-07);
    qc = fma_c(qc, z, 2.480158729876569e-05);
    qc = fma_c(qc, z, -0.0013888888888887398);
    qc = fma_c(qc, z, 0.041666666666666664);
    const double cr = fma(z * z, qc, fma(-0.5, z, 1.0));
    const int q = (int)kf & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// sincos by tiers: the short polynomials where they hold, libm beyond. The tier is each LANE's own (|x| <= 1e-3 with
// TINY, <= 0.25, <= 1e5, the rest), so what a lane gets does not depend on the lanes that share its wave; the wave
// only decides which tiers are EVALUATED - one of them when every lane sits in it (always the case for planetary
// fields of view), the others as well in a wave that straddles a boundary, with a select per lane.
// (LIBM = false: no libm tier - angles beyond 1e5 rad, NaN and infinities give NaN; for arguments that cannot get
//  there, the spin angle over a light-time span: the libm path costs the whole kernel its registers)
template <bool TINY = false, bool LIBM = true>
__device__ __forceinline__ void sincos_tiered(double x, double &s, double &c)
{
    const double ax = fabs(x);
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(true);
    const unsigned long long tiny_m = TINY ? __builtin_amdgcn_ballot_w64(ax <= 1e-3) : 0ull;
    if (TINY && tiny_m == lanes) {
        sincos_tiny(x, s, c);
        return;
    }
    const unsigned long long small_m = __builtin_amdgcn_ballot_w64(ax <= 0.25) & ~tiny_m;
    if (small_m == lanes) {
        sincos_small(x, s, c);
        return;
    }
    const unsigned long long med_m = __builtin_amdgcn_ballot_w64(ax <= 1e5) & ~(small_m | tiny_m);
    if (med_m == lanes) {
        sincos_medium(x, s, c);
        return;
    }
    // a wave across tiers (lanes outside a tier's range carry its garbage, which the selects drop)
    double sr = __builtin_nan(""), cr = __builtin_nan("");
    if (LIBM && (lanes & ~(tiny_m | small_m | med_m)) != 0) sincos(x, &sr, &cr);  // (NaN and infinities end here too)
    if (med_m != 0) {
        double t, u;
        sincos_medium(x, t, u);
        const bool m = __builtin_amdgcn_inverse_ballot_w64(med_m);
        sr = m ? t : sr;
        cr = m ? u : cr;
    }
    if (small_m != 0) {
        double t, u;
        sincos_small(x, t, u);
        const bool m = __builtin_amdgcn_inverse_ballot_w64(small_m);
        sr = m ? t : sr;
        cr = m ? u : cr;
    }
    if (TINY && tiny_m != 0) {
        double t, u;
        sincos_tiny(x, t, u);
        const bool m = __builtin_amdgcn_inverse_ballot_w64(tiny_m);
        sr = m ? t : sr;
        cr = m ? u : cr;
    }
    s = sr;
    c = cr;
}
__device__ __forceinline__ void sincos_auto(double x, double &s, double &c) { sincos_tiered<false>(x, s, c); }

}  // namespace pm
