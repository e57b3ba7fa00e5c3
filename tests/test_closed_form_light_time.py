"""
The closed-form light time of the spheroid frame kernel (k_disc_sph, profiles/EXPERIMENTS.md, round 3), restated with mpmath
from the geometry block and checked without a GPU:
  (1) it IS the fixed point of the reference's iteration lt = E((et - lt) - t0) for linear target motion;
  (2) the first-order step from the fixed point onto an epoch one quantum away lands on the intercept evaluated
      at that epoch (what the kernel does to follow the reference's rounding of `et - lt`);
  (3) the mask argument: a ray whose squared impact parameter at the fixed point is clear of 1 by the host's band
      hits in EVERY pass of the reference's sequence, or misses in the first.
That the kernel computes this is the `-m gpu` test test_closed_form_light_time_against_the_reference_sequence.
"""

import numpy as np
import pytest
from mpmath import mp, mpf, sqrt

from planetmapper_amd.scenarios import load_scenario

mp.dps = 50


def _consts(g):
    r0 = np.array(g.R0[:]).reshape(3, 3)
    radii = np.array(g.radii[:])
    o0 = -(r0 @ np.array(g.T0[:]))
    vb = r0 @ np.array(g.VT[:])
    return r0, radii, o0, vb


def _v(a):
    return [mpf(float(x)) for x in a]


def _dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def _evaluate(o0s, vbs, x, d, c):
    """one pass of the reference's sequence (surfpt_c on the unit sphere): light time it implies, P.P, hit"""
    y = [o - v * d for o, v in zip(o0s, vbs)]
    xx = _dot(x, x)
    k = _dot(y, x) / xx
    p = [a - k * b for a, b in zip(y, x)]
    p2 = _dot(p, p)
    if p2 > 1 or _dot(y, x) > 0:
        return None, p2, None
    root = sqrt((1 - p2) / xx)
    f = [a - root * b for a, b in zip(p, x)]
    return (-k - root) / c, p2, f


@pytest.mark.parametrize('which', ['jupiter_hst_2005', 'saturn_earth_2005'])
def test_closed_form_is_the_fixed_point_and_the_step_follows_the_quantum(which):
    g = load_scenario(which)
    _, radii, o0, vb = _consts(g)
    c = mpf(g.clight)
    o0s, vbs = _v(o0 / radii), _v(vb / radii)
    t0 = mpf(g.et - g.lt_c)
    span = mpf(g.et) - t0  # et - t0, exact in binary64 (Sterbenz)
    y00 = [o - v * span for o, v in zip(o0s, vbs)]
    w = [v / c for v in vbs]
    rng = np.random.default_rng(5)
    worst_fp, worst_step = 0, 0
    for _ in range(60):
        tgt = rng.normal(size=3)
        tgt = tgt / np.linalg.norm(tgt) * radii * rng.uniform(0, 0.995)
        u = tgt - o0
        u /= np.linalg.norm(u)
        x = _v(u / radii)
        xp = [a + b for a, b in zip(x, w)]
        ixp = 1 / _dot(xp, xp)
        kq = _dot(y00, xp) * ixp
        pq = [a - kq * b for a, b in zip(y00, xp)]
        p2 = _dot(pq, pq)
        root = sqrt((1 - p2) * ixp)
        s = -kq - root
        lt_star = s / c
        d_star = span - lt_star
        # (1) one more pass of the reference's iteration at the fixed point returns the same light time
        lt_again, _, f_star = _evaluate(o0s, vbs, x, d_star, c)
        worst_fp = max(worst_fp, abs(lt_again - lt_star))
        # (2) an epoch one quantum of et away
        dq = mpf(float(np.spacing(g.et))) * mpf(float(rng.uniform(-1, 1)))
        f = [a - root * b for a, b in zip(pq, xp)]
        sp = -2 * _dot(f, vbs) * (mpf(0.5) / root) * ixp
        f_q = [a + dq * (sp * b - v) for a, b, v in zip(f, xp, vbs)]
        _, _, f_exact = _evaluate(o0s, vbs, x, d_star + dq, c)
        worst_step = max(worst_step, max(abs(a - b) for a, b in zip(f_q, f_exact)))
    assert worst_fp < mpf('1e-40')
    # the step's error: X' stands in for X in the slide along the ray, i.e. (v / c) = 4e-5 of a displacement of
    # |VBs| dq = 6e-12 radii, over cos(emission) >= 0.1 for these rays: a few 1e-15 radii = 1e-13 deg on the body
    assert worst_step < mpf('5e-15')


@pytest.mark.parametrize('which', ['jupiter_hst_2005', 'saturn_earth_2005'])
def test_a_ray_clear_of_the_band_is_decided_alike_in_every_pass(which):
    g = load_scenario(which)
    _, radii, o0, vb = _consts(g)
    c = mpf(g.clight)
    o0s, vbs = _v(o0 / radii), _v(vb / radii)
    span = mpf(g.et) - mpf(g.et - g.lt_c)
    y00 = [o - v * span for o, v in zip(o0s, vbs)]
    w = [v / c for v in vbs]
    # the host's band (pm_capi.hip fill_params)
    dv = float(np.linalg.norm(vb / radii)) * 1.05 * float(radii.max()) / g.clight
    band = 1.5 * (2.0 * dv + dv * dv) + 1e-10
    rng = np.random.default_rng(11)
    n_hit = n_miss = 0
    centre = -o0 / np.linalg.norm(o0)
    e1 = np.cross(centre, [0.0, 0.0, 1.0])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(centre, e1)
    lim = float(radii.max() / np.linalg.norm(o0))
    for _ in range(400):
        # rays in a thin annulus around the limb: grazing hits and near misses
        ang = rng.uniform(0, 2 * np.pi)
        rho = lim * (1 + rng.uniform(-0.08, 0.02) * rng.uniform(0, 1) ** 3)
        u = centre + rho * (np.cos(ang) * e1 + np.sin(ang) * e2)
        u /= np.linalg.norm(u)
        x = _v(u / radii)
        xp = [a + b for a, b in zip(x, w)]
        ixp = 1 / _dot(xp, xp)
        kq = _dot(y00, xp) * ixp
        pq = [a - kq * b for a, b in zip(y00, xp)]
        p2 = float(_dot(pq, pq))
        if abs(p2 - 1.0) <= band:
            continue  # inside the band: the kernel walks the reference's sequence for this lane
        # the reference's sequence (CSPICE sincpt CN): passes until the light time settles, a miss in any pass is a miss
        lt = mpf(g.lt_c)
        hit = True
        for _it in range(10):
            d = span - lt  # (the epoch's rounding moves the target by 1e-10 of the band: not modelled)
            nlt, _, _ = _evaluate(o0s, vbs, x, d, c)
            if nlt is None:
                hit = False
                break
            if abs(nlt - lt) <= mpf('1e-17') * abs(mpf(g.et) - nlt):
                break
            lt = nlt
        assert hit == (p2 < 1.0), (which, p2, band)
        n_hit += hit
        n_miss += not hit
    assert n_hit > 20 and n_miss > 20
