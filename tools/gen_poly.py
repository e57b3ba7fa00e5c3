#!/usr/bin/env python3
"""
Generate the polynomial coefficients used by planetmapper_amd/csrc/pm_fastmath.hip.h.

Near-minimax (Chebyshev-interpolation) fits computed with mpmath at 60 digits, rounded
to binary64, then checked by evaluating the rounded polynomial in float64 Horner form
against mpmath on a dense grid. Prints C initialisers + the measured max abs error.

    python tools/gen_poly.py
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def fit(name, f, zmax, deg, full, xs):
    # f(z): the "tail" function of z = x^2; full(x, tail) reconstructs the function
    coeffs = mp.chebyfit(f, [0, zmax], deg + 1)  # highest degree first
    c64 = [float(c) for c in coeffs]
    # float64 Horner evaluation like the device code
    z = xs * xs
    acc = np.full_like(z, c64[0])
    for c in c64[1:]:
        acc = acc * z + c
    approx = full(xs, acc)
    exact = np.array([float(full_exact[name](mp.mpf(float(x)))) for x in xs])
    err = np.max(np.abs(approx - exact))
    print(f'// {name}: degree {deg} in z = x*x on z <= {float(zmax):.8g}, max abs err {err:.2e}')
    print('{' + ', '.join(f'{c!r}' for c in c64[::-1]) + '}  // c0..cN')
    return c64


full_exact = {
    'asin': mp.asin,
    'atan': mp.atan,
    'sin': mp.sin,
    'cos': mp.cos,
}

if __name__ == '__main__':
    def tail_asin(z):
        if z == 0:
            return mp.mpf(1) / 6
        s = mp.sqrt(z)
        return (mp.asin(s) - s) / (z * s)

    def tail_atan(z):
        if z == 0:
            return -mp.mpf(1) / 3
        s = mp.sqrt(z)
        return (mp.atan(s) - s) / (z * s)

    def tail_sin(z):
        if z == 0:
            return -mp.mpf(1) / 6
        s = mp.sqrt(z)
        return (mp.sin(s) - s) / (z * s)

    def tail_cos(z):
        if z == 0:
            return mp.mpf(1) / 24
        s = mp.sqrt(z)
        return (mp.cos(s) - 1 + z / 2) / (z * z)

    xs = np.linspace(0, 0.5, 20001)
    for deg in (10, 11, 12):
        fit('asin', tail_asin, mp.mpf('0.25'), deg, lambda x, t: x + x * (x * x) * t, xs)
    t8 = float(mp.tan(mp.pi / 8))
    xs = np.linspace(0, t8, 20001)
    for deg in (8, 9, 10):
        fit('atan', tail_atan, mp.tan(mp.pi / 8) ** 2, deg, lambda x, t: x + x * (x * x) * t, xs)
    xs = np.linspace(0, 0.25, 20001)
    for deg in (3, 4):
        fit('sin', tail_sin, mp.mpf('0.0625'), deg, lambda x, t: x + x * (x * x) * t, xs)
        fit('cos', tail_cos, mp.mpf('0.0625'), deg, lambda x, t: 1 - 0.5 * x * x + (x * x) * (x * x) * t, xs)
    # sincos_medium: |r| <= pi/4 after the Cody-Waite reduction
    xs = np.linspace(0, float(mp.pi / 4), 20001)
    for deg in (5,):
        fit('sin', tail_sin, (mp.pi / 4) ** 2, deg, lambda x, t: x + x * (x * x) * t, xs)
        fit('cos', tail_cos, (mp.pi / 4) ** 2, deg, lambda x, t: 1 - 0.5 * x * x + (x * x) * (x * x) * t, xs)
