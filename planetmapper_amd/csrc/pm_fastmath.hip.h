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
//
// Coefficients: tools/gen_poly.py (Chebyshev interpolation at 60 digits with mpmath,
// rounded to binary64 and re-verified in float64 Horner form).
#pragma once

#include <hip/hip_runtime.h>

namespace pm {

constexpr double kPiF = 3.14159265358979323846;
constexpr double kTanPi8 = 0.41421356237309503;

// asin(x) for |x| <= 0.5
__device__ __forceinline__ double asin_half(double x)
{
    const double z = x * x;
    double q = 0.028169218060881414;
    q = fma(q, z, -0.010749050339697808);
    q = fma(q, z, 0.01603551434914882);
    q = fma(q, z, 0.0078029494773533175);
    q = fma(q, z, 0.011875494382636922);
    q = fma(q, z, 0.013929652902326633);
    q = fma(q, z, 0.017355259955786323);
    q = fma(q, z, 0.02237204763174451);
    q = fma(q, z, 0.03038194736709848);
    q = fma(q, z, 0.044642857103423646);
    q = fma(q, z, 0.07500000000020764);
    q = fma(q, z, 0.1666666666666665);
    return fma(x * z, q, x);
}

// atan(t) for |t| <= tan(pi/8)
__device__ __forceinline__ double atan_small(double t)
{
    const double z = t * t;
    double q = 0.02275052699336167;
    q = fma(q, z, -0.04483334622272886);
    q = fma(q, z, 0.05736332165907643);
    q = fma(q, z, -0.06649613695291669);
    q = fma(q, z, 0.0769105515839315);
    q = fma(q, z, -0.09090852557176049);
    q = fma(q, z, 0.11111109636534361);
    q = fma(q, z, -0.1428571426609662);
    q = fma(q, z, 0.19999999999898407);
    q = fma(q, z, -0.3333333333333325);
    return fma(t * z, q, t);
}

// atan2(y, x) in (-pi, pi]; (y, x) finite and not both zero.
// atan(mn/mx) with mn <= mx is reduced by atan(q) = pi/4 + atan((q - 1)/(q + 1)) for
// q > tan(pi/8); numerator and denominator are chosen BEFORE dividing, so the whole
// function costs one division.
__device__ __forceinline__ double atan2_fast(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool big = mn > kTanPi8 * mx;
    const double num = big ? mn - mx : mn;
    const double den = big ? mn + mx : mx;
    double a = atan_small(num / den);
    a = big ? a + 0.25 * kPiF : a;
    a = (ay > ax) ? 0.5 * kPiF - a : a;
    a = (x < 0.0) ? kPiF - a : a;
    return copysign(a, y);
}

// sin and cos for |x| <= 0.25
__device__ __forceinline__ void sincos_small(double x, double &s, double &c)
{
    const double z = x * x;
    double qs = 2.7526021333363912e-06;
    qs = fma(qs, z, -0.00019841257617613103);
    qs = fma(qs, z, 0.008333333331805479);
    qs = fma(qs, z, -0.1666666666666637);
    s = fma(x * z, qs, x);
    double qc = -2.753123559604959e-07;
    qc = fma(qc, z, 2.4801577114157226e-05);
    qc = fma(qc, z, -0.0013888888887615533);
    qc = fma(qc, z, 0.041666666666666415);
    c = fma(z * z, qc, fma(-0.5, z, 1.0));
}

// sincos with a wave-uniform choice: the short polynomials when every lane of the wave is
// inside their range (always true for planetary fields of view), libm otherwise.
__device__ __forceinline__ void sincos_auto(double x, double &s, double &c)
{
    if (__all(fabs(x) <= 0.25)) {
        sincos_small(x, s, c);
    } else {
        sincos(x, &s, &c);
    }
}

}  // namespace pm
