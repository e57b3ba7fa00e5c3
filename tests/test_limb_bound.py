"""
The limb bound of the frame kernels (pm_capi.hip fill_params, profiles/EXPERIMENTS.md, round 3), restated in numpy and held
against the CPU oracle: no pixel that the oracle finds ON the body - with the pre-mask off, i.e. every pixel
ray-tested as the reference does with optimize_speed=False - may lie outside the bound's circle. Runs without
a GPU: it pins the FORMULA (reach of the limb, offset of the angular origin, spherical excess, half a pixel);
that the library applies it is the `-m gpu` test test_limb_bound_is_never_tighter_than_the_limb.
"""

import numpy as np
import pytest

from oracle import oracle
from planetmapper_amd.scenarios import load_scenario


def limb_bound_radius_px(g, r0, alt=0.0):
    """pixel radius beyond which no line of sight can meet the body (None: bound not applicable)"""
    radii = np.array(g.radii[:]) + alt
    rmax = float(radii.max())
    t0 = np.array(g.T0[:])
    dist = float(np.linalg.norm(t0))
    vt = float(np.linalg.norm(np.array(g.VT[:])))
    reach = rmax * (1.0 + 1e-9) + 4.0 * vt * rmax / g.clight + 1e-6
    m0 = np.array(g.M[:3])
    off = float(np.linalg.norm(np.cross(m0, t0))) / dist
    s_rad = (g.diameter_arcsec / (2.0 * r0)) * (np.pi / 648000.0)  # BodyXY.get_plate_scale_arcsec, in rad / pixel
    if not (reach / dist < 0.09 and off < 0.009):
        return None
    theta = np.arcsin(reach / dist) + 1.001 * off
    return theta / s_rad * 1.002 + 0.5


def _near_field(distance_km):
    from planetmapper_amd.ephem import Ephemeris, RotationModel
    from planetmapper_amd.geometry import CLIGHT, GeometryBuilder
    from planetmapper_amd.scenarios import _load_json

    d = _load_json('jupiter_hst_2005')
    gb = GeometryBuilder(Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck']), d['target_id'])
    h = d['header']
    return gb.build(d['et'] + 3 * 3600.0, observer_velocity=[-10.0, 28.0, 3.0],
                    target_ra_dec_dist_lt=(h['PLANMAP TARGET RA'] + 40.0, h['PLANMAP TARGET DEC'] + 20.0, distance_km,
                                           distance_km / CLIGHT))  # fmt: skip


@pytest.mark.parametrize('which', ['jupiter_hst_2005', 'saturn_earth_2005', 'near_2.5e6', 'near_9e5', 'near_2e5'])
def test_no_pixel_on_the_body_lies_outside_the_limb_bound(which):
    g = _near_field(float(which[5:])) if which.startswith('near_') else load_scenario(which)
    rng = np.random.default_rng(7)
    checked = 0
    for k in range(6):
        nx, ny = int(rng.integers(120, 260)), int(rng.integers(120, 260))
        r0 = float(rng.uniform(20.0, 110.0))
        x0, y0 = float(rng.uniform(0.2, 0.8) * nx), float(rng.uniform(0.2, 0.8) * ny)
        rot = float(rng.uniform(0, 360))
        alt = float(rng.choice([0.0, 2500.0, -400.0]))
        d = oracle.make_disc(x0, y0, r0, rot, nx, ny, optimize_speed=False)
        d.rotation_rad = float(np.deg2rad(rot))
        emi = oracle.backplanes_img(g, d, ['EMISSION'], alt=alt)['EMISSION']
        rt = limb_bound_radius_px(g, r0, alt)
        if rt is None:
            assert which == 'near_2e5'  # the disc subtends more than 0.09 rad: the bound stands down
            continue
        yy, xx = np.mgrid[0:ny, 0:nx]
        rr = np.hypot(xx - x0, yy - y0)
        on = np.isfinite(emi)
        assert on.any()
        assert rr[on].max() <= rt, (which, k, float(rr[on].max()), rt)
        # ... and it is a useful bound: within 1.5 % + 1.5 pixels of the outermost pixel on the body's long axis
        assert rt <= 1.015 * max(rr[on].max(), r0 * (max(g.radii[:]) + alt) / g.radii[0] * 0.99) + 1.5, (which, k, rt)
        checked += 1
    assert checked > 0 or which == 'near_2e5'
