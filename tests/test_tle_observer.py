"""
The reference's canonical observer - HST (`tests/test_body.py:29-31`, every golden FITS) - from its own ephemeris: an SPK
type 10 segment of two-line elements, `spke10` restated in planetmapper_amd.ephem (`TleSegment`: SGP4, the cosine blend of
the two bracketing element sets, TEME -> J2000 with the segment's nutation angles). CPU only.

What pins it is the reference's golden header: TARGET RA / DEC / DISTANCE of Jupiter seen from HST are printed with 16-17
digits, which fixes the observer to ~1e-6 km; and the golden RADIAL-VELOCITY plane, which fixes its velocity along the line of
sight. Rounds 1-5 used those two the other way round (position FROM the header, velocity FITTED to the plane); here nothing is
fitted, and both are checks.
"""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle
from planetmapper_amd import ephem
from planetmapper_amd.geometry import GeometryBuilder, radrec
from planetmapper_amd.scenarios import load_scenario, scenario_info

REF_KERNELS = '/root/reference/tests/data/kernels'


@pytest.fixture(scope='module')
def parts():
    d = scenario_info('jupiter_hst_2005')
    eph = ephem.Ephemeris.from_json(d['ephemeris'])
    rot = ephem.RotationModel.from_json(d['pck'])
    return d, eph, rot, GeometryBuilder(eph, rot, 599)


def test_scenario_holds_hst_as_two_line_elements_and_nothing_fitted(parts):
    d, eph, _, _ = parts
    assert d['observer_id'] == -48 and 'observer_velocity_fit' not in d
    tle = [s for s in eph.segments if isinstance(s, ephem.TleSegment)]
    assert len(tle) == 1 and tle[0].target == -48 and tle[0].center == 399 and tle[0].frame == 1
    assert 2 <= len(tle[0].epochs) <= 12 and tle[0].packets.shape[1] == 14
    # the segment's own geophysical constants (not WGS-72's 6378.135 only by coincidence: they are read, not assumed)
    assert np.allclose(tle[0].geophs[:3], (1.082616e-3, -2.53881e-6, -1.65597e-6), rtol=1e-14, atol=0) and tle[0].geophs[6] == 6378.135
    period_min = 2 * math.pi / tle[0].packets[0][8]
    assert 94.0 < period_min < 98.0  # HST: near-earth model


def test_observer_position_is_where_the_golden_header_puts_it(parts):
    """TARGET RA / DEC / DISTANCE / LIGHT-TIME of the header -> where the observer must be; SGP4 puts HST there to 1e-6 km"""
    d, eph, _, gb = parts
    h, et = d['header'], d['et']
    t0 = et - h['PLANMAP LIGHT-TIME']
    implied = gb._ptarget(t0) - radrec(h['PLANMAP DISTANCE'], math.radians(h['PLANMAP TARGET RA']), math.radians(h['PLANMAP TARGET DEC']))
    pos, vel, _ = eph.ssb_state(-48, et)
    assert np.linalg.norm(pos - implied) <= 1.0e-6  # km (the header's own digits are worth 4e-7 km)
    geo = pos - eph.ssb_state(399, et)[0]
    assert 6930.0 < np.linalg.norm(geo) < 6960.0 and 7.5 < np.linalg.norm(vel - eph.ssb_state(399, et)[1]) < 7.7
    g = load_scenario('jupiter_hst_2005')
    desc = GeometryBuilder.describe(g)
    assert abs(desc['target_ra'] - h['PLANMAP TARGET RA']) <= 1e-12 and abs(desc['target_dec'] - h['PLANMAP TARGET DEC']) <= 1e-12
    assert abs(desc['target_distance'] - h['PLANMAP DISTANCE']) <= 1e-6 and abs(desc['target_light_time'] - h['PLANMAP LIGHT-TIME']) <= 1e-11


def test_unfitted_velocity_reproduces_the_golden_radial_velocity_plane(parts):
    """... and the drift of Jupiter's pole (pm_geometry.WP) is what the fit of rounds 1-5 had absorbed into the observer"""
    d, eph, rot, gb = parts
    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav.npz'))
    ok = np.isfinite(gold['RADIAL-VELOCITY'])
    disc = oracle.make_disc(2.5, 3.1, 3.9, 123.456, 7, 10)
    g = load_scenario('jupiter_hst_2005')
    out = oracle.backplanes_img(g, disc, ['RADIAL-VELOCITY', 'DOPPLER'])
    assert np.max(np.abs(out['RADIAL-VELOCITY'][ok] - gold['RADIAL-VELOCITY'][ok])) <= 1e-10  # km/s (measured 4.4e-11)
    assert np.max(np.abs(out['DOPPLER'][ok] - gold['DOPPLER'][ok])) <= 5e-16
    # without the pole's drift: a gradient of 2e-14 (km/s) / km across the disc, 1.3e-9 km/s at its edge
    drift = np.array(g.WP[:])
    assert 5e-14 < np.linalg.norm(drift) < 1e-13 and abs(drift @ np.array(g.R0[6:9])) < 1e-18
    g0 = g.copy()
    g0.WP[:] = [0.0, 0.0, 0.0]
    rv0 = oracle.backplanes_img(g0, disc, ['RADIAL-VELOCITY'])['RADIAL-VELOCITY']
    assert 5e-10 < np.max(np.abs(rv0[ok] - gold['RADIAL-VELOCITY'][ok])) < 3e-9
    # the old route - position from the header, three velocity components fitted to the plane - as a cross-check: the
    # fit agrees with the ephemeris along the line of sight to 1e-8 km/s (across it the plane says little: 2e-5)
    h, et = d['header'], d['et']
    tgt = (h['PLANMAP TARGET RA'], h['PLANMAP TARGET DEC'], h['PLANMAP DISTANCE'], h['PLANMAP LIGHT-TIME'])
    vo = np.array(eph.ssb_state(399, et)[1])

    def rv_of(v):
        gg = gb.build(et, observer_velocity=v, target_ra_dec_dist_lt=tgt)
        gg.WP[:] = [0.0, 0.0, 0.0]
        return oracle.backplanes_img(gg, disc, ['RADIAL-VELOCITY'])['RADIAL-VELOCITY'][ok]

    for _ in range(3):
        jac = np.stack([(rv_of(vo + e) - rv_of(vo - e)) / 2e-3 for e in 1e-3 * np.eye(3)], axis=1)
        vo = vo - np.linalg.lstsq(jac, rv_of(vo) - gold['RADIAL-VELOCITY'][ok], rcond=None)[0]
    los = radrec(1.0, math.radians(h['PLANMAP TARGET RA']), math.radians(h['PLANMAP TARGET DEC']))
    diff = vo - np.array(g.VO[:])
    assert abs(diff @ los) < 1e-7 and np.linalg.norm(diff) < 1e-4


def test_blend_and_frame_of_spke10(parts):
    """the pieces, on the fixture's own element sets"""
    d, eph, _, _ = parts
    tle = [s for s in eph.segments if isinstance(s, ephem.TleSegment)][0]
    et = d['et']
    i1, i2 = tle._bracket(et)
    assert i2 == i1 + 1 and tle.epochs[i1] <= et < tle.epochs[i2]
    # at an element set's own epoch the blend is that set alone (weight 1 / 0), from either side
    t1 = float(tle.epochs[i1])
    own = ephem._Sgp4(tle.geophs, tle.packets[i1]).state(t1)[0]
    m = tle._j2000_to_teme_at(t1, i1, i1)
    for eps in (0.0, 1e-3):
        p = tle.state(t1 + eps)[0]
        assert np.linalg.norm(m @ p - ephem._Sgp4(tle.geophs, tle.packets[i1]).state(t1 + eps)[0]) < 1e-6, eps
    assert abs(np.linalg.norm(own) - np.linalg.norm(tle.state(t1)[0])) < 1e-9
    # the rotation is a rotation, 5 years of precession from the identity, and continuous across an element set
    mt = tle._j2000_to_teme_at(et, i1, i2)
    assert np.allclose(mt @ mt.T, np.eye(3), atol=1e-15) and 1.0e-3 < np.linalg.norm(mt - np.eye(3)) < 2.5e-3
    before, after = tle.state(float(tle.epochs[i2]) - 1e-3), tle.state(float(tle.epochs[i2]) + 1e-3)
    assert np.linalg.norm(after[0] - before[0] - 2e-3 * after[1]) < 1e-6
    # the velocity is SGP4's own (not the derivative of the position: the two differ by 1e-5 km/s in the model itself),
    # plus the weight's derivative and the frame's rate; the acceleration is the two-body estimate
    p, v, a = tle.state(et)
    num = (tle.state(et + 0.5)[0] - tle.state(et - 0.5)[0])
    assert 1e-6 < np.linalg.norm(num - v) < 5e-5
    assert abs(np.linalg.norm(a) - 398600.8 / np.linalg.norm(p) ** 2) < 1e-5
    # outside its span / a deep-space element set is refused, not extrapolated silently
    assert not tle.covers(tle.et_end + 1.0)
    deep = tle.packets[i1].copy()
    deep[8] = 2 * math.pi / 720.0  # a 12 h orbit
    with pytest.raises(ValueError, match='deep-space'):
        ephem._Sgp4(tle.geophs, deep)


@pytest.mark.skipif(not os.path.isdir(REF_KERNELS), reason='the reference checkout (kernel files) is not on this machine')
def test_the_kernel_file_itself_and_bodyxy_from_kernels(parts):
    """hst.bsp as the reference ships it: the generic-segment reader, and `BodyXY(..., observer='HST', kernels=...)`"""
    d, eph, _, _ = parts
    segs = ephem.read_spk_segments(os.path.join(REF_KERNELS, 'testing', 'nested', 'directory', 'hst.bsp'))
    assert len(segs) == 1 and isinstance(segs[0], ephem.TleSegment) and len(segs[0].epochs) == 15518
    et = d['et']
    full, mini = segs[0].state(et), eph._find(-48, et).state(et)
    assert np.array_equal(full[0], mini[0]) and np.array_equal(full[1], mini[1])
    from planetmapper_amd.kernels import geometry_from_kernels

    g = geometry_from_kernels('jupiter', '2005-01-01T00:00:00', 'HST', REF_KERNELS)
    ref = load_scenario('jupiter_hst_2005')
    assert g.et == ref.et == 157809664.1839331
    for name in ('T0', 'VO', 'VT', 'WP', 'R0', 'sub_sp'):
        assert np.allclose(np.array(getattr(g, name)[:]), np.array(getattr(ref, name)[:]), rtol=0, atol=1e-9 * max(1.0, np.abs(getattr(ref, name)[:]).max())), name


@pytest.mark.skipif(not os.path.isdir(REF_KERNELS), reason='the reference checkout (kernel files) is not on this machine')
def test_the_whole_span_of_hst_bsp_is_an_orbit():
    """15 518 element sets, 1990-2023: every epoch sampled gives HST's orbit (no decayed set, no deep-space set, no jump)"""
    seg = ephem.read_spk_segments(os.path.join(REF_KERNELS, 'testing', 'nested', 'directory', 'hst.bsp'))[0]
    rng = np.random.default_rng(10)
    ets = rng.uniform(seg.et_begin + 10.0, seg.et_end - 10.0, 150)
    for et in ets:
        p, v, _ = seg.state(float(et))
        r, speed = float(np.linalg.norm(p)), float(np.linalg.norm(v))
        assert 6850.0 < r < 7010.0 and 7.50 < speed < 7.65, (et, r, speed)  # 470-630 km above 6378 km, over 33 years of decay and reboosts
        incl = np.degrees(np.arccos(np.cross(p, v)[2] / np.linalg.norm(np.cross(p, v))))
        assert 28.3 < incl < 28.7, (et, incl)  # HST's inclination, in J2000 (precession moves the equator by < 0.2 deg over the span)
    # across element-set epochs the blend is continuous in position and velocity (weights 1 / 0 there)
    for k in rng.integers(1, len(seg.epochs) - 1, 40):
        t = float(seg.epochs[k])
        a, b = seg.state(t - 1e-4), seg.state(t + 1e-4)
        assert np.linalg.norm(b[0] - a[0] - 2e-4 * a[1]) < 1e-6 and np.linalg.norm(b[1] - a[1]) < 1e-5, k


def test_sgp4_against_the_published_verification_vectors():
    """
    Independent of the reference and of its kernels: the near-earth verification case of Vallado, Crawford, Hujsak & Kelso
    2006 ("Revisiting Spacetrack Report #3", AIAA 2006-6753: satellite 00005, 58002B, epoch 2000-06-27; WGS-72) - TEME
    position / velocity at 0, 360 and 720 minutes from the epoch, as published with the paper's code (8-9 decimals).
    The segment-style constants carry KE to 9 digits (7.43669161e-2, as in hst.bsp) where the paper's code derives it from
    mu: 1e-10 relative, 2e-6 km here.
    """
    geophs = [1.082616e-3, -2.53881e-6, -1.65597e-6, 7.43669161e-2, 120.0, 78.0, 6378.135, 1.0]
    d2r = math.pi / 180.0
    elements = [0.0, 0.0, 0.28098e-4, 34.2682 * d2r, 348.7242 * d2r, 0.1859667, 331.7664 * d2r, 19.3264 * d2r,
                10.82419157 * 2.0 * math.pi / 1440.0, 0.0]  # NDT20 NDD60 BSTAR INCL NODE0 ECC OMEGA M0 N0 (rad / min) EPOCH
    sat = ephem._Sgp4(geophs, elements)
    published = {
        0.0: (7022.46529266, -1400.08296755, 0.03995155, 1.893841015, 6.405893759, 4.534807250),
        360.0: (-7154.03120202, -3783.17682504, -3536.19412294, 4.741887409, -4.151817765, -2.093935425),
        720.0: (-7134.59340119, 6531.68641334, 3260.27186483, -4.113793027, -2.911922039, -2.557327851),
    }
    for minutes, ref in published.items():
        r, v = sat.state(60.0 * minutes)
        assert np.max(np.abs(r - np.array(ref[:3]))) < 5e-6, (minutes, r)
        assert np.max(np.abs(v - np.array(ref[3:]))) < 5e-9, (minutes, v)
