"""
The reference-side binding (planetmapper_amd/reference_binding.py) EXECUTED: `geometry_from_body` asks a spiceypy-shaped
module (tests/spice_standin.py, on this repo's own kernel readers) and a duck-typed Body (the attributes a reference
`planetmapper.Body` carries after `__init__`, body.py:501-606 / base.py:795-839) and must reproduce the block
`GeometryBuilder` computes from the same kernel data analytically - to the bars of tests/test_motion_model.py:
displacement of target and Sun < 1e-9 km over +-4 R/c, rotation increment < 1e-13 rad over the disc's light-time
span; and (`-m gpu`) a 256^2 frame through the engine from either block inside tests/parity.py.
Cases: Jupiter / Earth 2009 (SPK type 3 target: state velocity != derivative of the position series), Saturn / Earth
2016, Mars / Earth 2012 (type 2 chain only), an Io-like moon (periodic terms in W, RA, Dec: the rate about the body's
own z axis is not the PM[1] coefficient) and the Moon from Earth (a moon with its own segment, 60 radii away).
"""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = {
    'jupiter_earth_2009': ('JUPITER', 'IAU_JUPITER'),
    'saturn_earth_2016': ('SATURN', 'IAU_SATURN'),
    'mars_earth_2012': ('MARS', 'IAU_MARS'),
    'io_like_earth_2009': ('IO', 'IAU_IO'),
    'moon_earth_2012': ('MOON', 'IAU_MOON'),
}


def build(name):
    """(fixture, GeometryBuilder's block, the binding's block through the stand-in)"""
    from planetmapper_amd.ephem import Ephemeris, RotationModel
    from planetmapper_amd.geometry import GeometryBuilder
    from planetmapper_amd.reference_binding import geometry_from_body
    from spice_standin import DuckBody, SpiceStandIn

    d = json.load(open(os.path.join(GOLDEN, f'motion_{name}.json')))
    eph, rot = Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck'])
    g = GeometryBuilder(eph, rot, d['target_id']).build(d['et'], observer_id=d['observer_id'])
    target, frame = CASES[name]
    spice = SpiceStandIn(eph, {frame: rot}, {target: d['target_id'], 'SUN': 10, 'EARTH': 399})
    body = DuckBody(g, target=target, target_id=d['target_id'], observer='EARTH', frame=frame)
    return d, g, geometry_from_body(body, spice), spice


def vec(g, name):
    return np.array(getattr(g, name)[:])


@pytest.mark.parametrize('name', list(CASES))
def test_binding_reproduces_the_geometry_builder_block(name):
    d, g, b, spice = build(name)
    r_c = max(g.radii[:]) / g.clight
    span = 4.0 * r_c
    # what Body.__init__ handed over comes back unchanged
    for f in ('et', 'lt_c', 'clight', 'sub_et', 'sub_dist', 'ring_k', 'diameter_arcsec', 'km_per_arcsec', 'np_angle_rad', 'west_positive'):
        assert getattr(b, f) == getattr(g, f), f
    for f in ('radii', 'T0', 'sub_sp', 'sub_ray', 'sub_obsvec', 'ring_n', 'M', 'R0'):
        assert np.array_equal(vec(b, f), vec(g, f)), f
    # the motion model: what the kernels add to T0 / S0 over the spans they use
    worst_t = np.max(np.abs((vec(b, 'VT') - vec(g, 'VT')) * span) + np.abs(vec(b, 'AT') - vec(g, 'AT')) * span * span / 2)
    worst_s = np.max(np.abs((vec(b, 'VS') - vec(g, 'VS')) * span) + np.abs(vec(b, 'AS') - vec(g, 'AS')) * span * span / 2)
    assert worst_t < 1e-9, (name, worst_t)  # km: target displacement over +-4 R/c
    assert worst_s < 1e-9, (name, worst_s)  # km: the Sun's over the same span
    assert abs(b.ts0 - g.ts0) < 2e-7 and np.max(np.abs(vec(b, 'S0') - vec(g, 'S0'))) < 4e-6, name  # (the Sun moves 0.013 km/s)
    # the state (radial velocity: 1e-9 km/s bar) and the observer
    assert np.max(np.abs(vec(b, 'DVT') - vec(g, 'DVT'))) < 1e-10, name
    assert np.max(np.abs(vec(b, 'DAT') - vec(g, 'DAT'))) * span < 1e-10, name
    assert np.array_equal(vec(b, 'VO'), vec(g, 'VO')), name
    # rotation increment Rz(wdot d) over a disc's light-time span (1e-13 rad) and over 4 R/c
    assert abs(b.wdot - g.wdot) * r_c < 1e-13, (name, b.wdot, g.wdot)
    assert abs(b.wdot - g.wdot) * span < 4e-13, name
    assert abs(b.lst_sun_lon - g.lst_sun_lon) < 1e-9, name
    # the drift of the pole (STATE planes: what sxform's derivative block holds beyond the spin) to 1e-4 of itself - 1e-13 km/s
    assert np.max(np.abs(vec(b, 'WP') - vec(g, 'WP'))) < 1e-4 * max(np.linalg.norm(vec(g, 'WP')), 1e-16), (name, vec(b, 'WP'), vec(g, 'WP'))
    assert abs(vec(b, 'WP') @ vec(g, 'R0')[6:9]) < 1e-17
    assert 'spksfs' in spice.calls  # the chain was walked segment by segment


def test_binding_with_an_observer_on_two_line_elements_meets_the_golden_planes():
    """
    The reference's own golden geometry - Jupiter from HST, an observer whose SPK is type 10 - through the binding: the
    observer's state is whatever `spkssb` says (here the stand-in on `ephem.TleSegment`; in a deployment CSPICE's spke10), the
    block equals GeometryBuilder's, and the oracle on it reproduces the golden RADIAL-VELOCITY plane with nothing fitted.
    """
    from oracle import oracle
    from planetmapper_amd.ephem import Ephemeris, RotationModel
    from planetmapper_amd.geometry import GeometryBuilder
    from planetmapper_amd.reference_binding import geometry_from_body
    from planetmapper_amd.scenarios import scenario_info
    from spice_standin import DuckBody, SpiceStandIn

    d = scenario_info('jupiter_hst_2005')
    eph, rot = Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck'])
    g = GeometryBuilder(eph, rot, 599).build(d['et'], observer_id=-48)
    spice = SpiceStandIn(eph, {'IAU_JUPITER': rot}, {'JUPITER': 599, 'SUN': 10, 'EARTH': 399, 'HST': -48})
    b = geometry_from_body(DuckBody(g, target='JUPITER', target_id=599, observer='HST', frame='IAU_JUPITER'), spice)
    assert np.array_equal(vec(b, 'VO'), vec(g, 'VO')) and np.array_equal(vec(b, 'T0'), vec(g, 'T0'))
    assert np.max(np.abs(vec(b, 'WP') - vec(g, 'WP'))) < 1e-4 * np.linalg.norm(vec(g, 'WP'))
    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav.npz'))
    ok = np.isfinite(gold['RADIAL-VELOCITY'])
    disc = oracle.make_disc(2.5, 3.1, 3.9, 123.456, 7, 10)
    out = oracle.backplanes_img(b, disc, ['RADIAL-VELOCITY', 'LON-GRAPHIC', 'EMISSION'])
    assert np.max(np.abs(out['RADIAL-VELOCITY'][ok] - gold['RADIAL-VELOCITY'][ok])) <= 1e-10
    assert np.max(np.abs(out['LON-GRAPHIC'][ok] - gold['LON-GRAPHIC'][ok])) <= 3e-9 and np.max(np.abs(out['EMISSION'][ok] - gold['EMISSION'][ok])) <= 2e-9


@pytest.mark.parametrize('name', ['io_like_earth_2009', 'moon_earth_2012', 'saturn_earth_2016'])
def test_the_rate_about_the_body_z_axis_is_not_the_pm_coefficient(name):
    """
    Missing section 2 of the round-4 verdict: a `wdot` taken from PM[1] alone drops the periodic terms of W (every moon
    of pck00010.tpc has them) and the pole's precession along the pole: Io 9e-14 rad/s, Saturn 2e-13, the Moon 1e-10 -
    against a rotation budget of 1e-13 rad over the spans the kernels use. The derivative block of sxform has both.
    """
    d, g, b, _ = build(name)
    pm1 = np.deg2rad(d['pck']['pm'][1]) / 86400.0
    print(f'\n[{name}] PM[1] alone is {abs(pm1 - g.wdot):.2e} rad/s from the rate about the body z axis; the binding {abs(b.wdot - g.wdot):.2e}')
    assert abs(pm1 - g.wdot) > 5e-14
    assert abs(b.wdot - g.wdot) < 1e-3 * abs(pm1 - g.wdot)


def test_binding_refuses_what_the_engine_does_not_evaluate():
    from planetmapper_amd.reference_binding import geometry_from_body

    d, g, b, spice = build('mars_earth_2012')
    from spice_standin import DuckBody

    body = DuckBody(g, target='MARS', target_id=499, observer='EARTH', frame='IAU_MARS')
    body.aberration_correction = 'LT+S'
    with pytest.raises(NotImplementedError):
        geometry_from_body(body, spice)
    body.aberration_correction = 'CN'
    body.observer_frame = 'ECLIPJ2000'
    with pytest.raises(NotImplementedError):
        geometry_from_body(body, spice)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['jupiter_earth_2009', 'io_like_earth_2009', 'moon_earth_2012'])
def test_a_frame_from_the_binding_block_equals_the_frame_from_the_builder_block(name):
    from parity import compare_planes
    from planetmapper_amd.engine import Engine

    d, g, b, _ = build(name)
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH', 'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER', 'LOCAL-SOLAR-TIME',
             'RA', 'DEC', 'RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']  # fmt: skip
    eng = Engine(0)
    try:
        out = {}
        for key, block in (('builder', g), ('binding', b)):
            eng.set_geometry(block)
            eng.set_disc(127.5, 127.5, 100.0, 0.4, 256, 256, True)
            out[key] = eng.backplanes_img(names)
        lon, lat = np.meshgrid(np.arange(2.5, 360, 5.0), np.arange(-87.5, 90, 5.0))
        maps = {}
        for key, block in (('builder', g), ('binding', b)):
            eng.set_geometry(block)
            eng.set_disc(127.5, 127.5, 100.0, 0.4, 256, 256, True)
            maps[key] = eng.xy_map(lon, lat)
    finally:
        eng.close()
    compare_planes(out['binding'], out['builder'], names, g, plate_scale_arcsec=g.diameter_arcsec / 200.0)
    for a, c in zip(maps['binding'], maps['builder']):
        assert np.array_equal(np.isnan(a), np.isnan(c)) and np.nanmax(np.abs(a - c)) < 1e-9


@pytest.mark.gpu
def test_the_mixin_over_a_duck_typed_bodyxy():
    """
    `HipBackplanes` in front of a class with the reference BodyXY's disc accessors, caches and map-grid helper
    (body_xy.py:818-871, 3290-3300; base.py:58-112): every producer it overrides returns what the engine returns for the same
    block and disc, in the reference's shapes, read-only, cached where the reference caches (image space: until the Body's
    `_cache` is cleared, per altitude adjustment; map space: `_stable_cache`, per set of map keywords).
    """
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.reference_binding import _IMG_FAMILIES, _MAP_FAMILIES, HipBackplanes
    from spice_standin import DuckBody

    d, g, b, spice = build('jupiter_earth_2009')

    class DuckBodyXY(DuckBody):
        _nx, _ny, _optimize_speed, _alt_adjustment = 200, 160, True, 0.0

        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self._cache, self._stable_cache = {}, {}
            self.disc = [99.5, 79.5, 60.0, 0.3]

        def get_x0(self): return self.disc[0]
        def get_y0(self): return self.disc[1]
        def get_r0(self): return self.disc[2]
        def _get_rotation_radians(self): return self.disc[3]

        def set_x0(self, x0):  # body_xy.py:696-700: a new disc empties the clearable cache
            self.disc[0] = x0
            self._cache.clear()

        def _get_lonlat_map(self, degree_interval=10.0, **kw):
            lon, lat = np.meshgrid(np.arange(degree_interval / 2, 360, degree_interval)[::-1], np.arange(-90 + degree_interval / 2, 90, degree_interval))
            return np.stack([lon, lat], axis=-1)

        def get_x_map(self, **kw): return self._get_xy_map(**kw)[..., 0]
        def get_y_map(self, **kw): return self._get_xy_map(**kw)[..., 1]

        def illumination_angles_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):  # body.py:2295: the scalar SPICE path
            return ('the reference\'s own', lon, lat, alt, planetocentric)

    class Bound(HipBackplanes, DuckBodyXY):
        _hip_spice = spice

    body = Bound(g, target='JUPITER', target_id=599, observer='EARTH', frame='IAU_JUPITER')
    eng = Engine(0)
    try:
        eng.set_geometry(b)
        eng.set_disc(99.5, 79.5, 60.0, 0.3, 200, 160, True)
        # ---- image space: every overridden producer against the engine's planes of the same family
        for method, names, how in _IMG_FAMILIES:
            ref = eng.backplanes_img(list(names))
            got = getattr(body, method)()
            assert got is getattr(body, method)(), method  # cached
            if how == 'stack':
                assert got.shape == (160, 200, len(names)) and not got.flags.writeable, method
                parts = [got[..., i] for i in range(len(names))]
            elif how == 'tuple':
                assert isinstance(got, tuple) and len(got) == len(names), method
                parts = list(got)
            else:
                parts, names = [got], (how,)
            for part, n in zip(parts, names):
                assert part.shape == (160, 200) and not part.flags.writeable and np.array_equal(part, ref[n], equal_nan=True), (method, n)
        assert np.isfinite(body.get_local_solar_time_img()).sum() > 1000 and np.isfinite(body._get_ring_plane_coordinate_imgs()[0]).sum() > 1000
        # ---- the clearable cache: a new disc gives new planes, an altitude adjustment its own entry
        lon_before = body._get_lonlat_img()
        body.set_x0(104.5)
        eng.set_disc(104.5, 79.5, 60.0, 0.3, 200, 160, True)
        lon_after = body._get_lonlat_img()
        assert lon_after is not lon_before and np.array_equal(lon_after[..., 0], eng.backplanes_img(['LON-GRAPHIC', 'LAT-GRAPHIC'])['LON-GRAPHIC'], equal_nan=True)
        body._alt_adjustment = 2500.0
        hi = body._get_illumination_gie_img()
        assert np.array_equal(hi[..., 2], eng.backplanes_img(['PHASE', 'INCIDENCE', 'EMISSION'], alt=2500.0)['EMISSION'], equal_nan=True)
        body._alt_adjustment = 0.0
        assert body._get_illumination_gie_img() is not hi
        # ---- map space, on the grid the (duck) reference builds for the keywords
        for kw in ({}, {'degree_interval': 15.0}, {'degree_interval': 15.0, 'alt': 1000.0}):
            grid = body._get_lonlat_map(**kw)
            for method, names, how in _MAP_FAMILIES:
                ref = eng.backplanes_map(list(names), grid[..., 0], grid[..., 1], alt=kw.get('alt', 0.0))
                got = getattr(body, method)(**kw)
                parts = [got[..., i] for i in range(len(names))] if how == 'stack' else (list(got) if how == 'tuple' else [got])
                for part, n in zip(parts, names if how in ('stack', 'tuple') else (how,)):
                    assert part.shape == grid.shape[:2] and not part.flags.writeable and np.array_equal(part, ref[n], equal_nan=True), (method, n, kw)
        assert len([k for k in body._stable_cache if k[0] == '_hip_map']) >= 3 * 5  # kept per set of keywords ...
        assert any(k[0] == '_hip_map' and 'PIXEL-X' in k[1] for k in body._cache)  # ... but the x / y map goes with the disc
        # ---- map_img through the overridden x / y map
        grid = body._get_lonlat_map()
        x, y = eng.xy_map(grid[..., 0], grid[..., 1])
        img = np.random.default_rng(0).standard_normal((160, 200))
        for interp in ('linear', 'cubic'):
            m = body.map_img(img, interpolation=interp)
            assert m.shape == grid.shape[:2] and np.array_equal(m, eng.map_cube(img, x, y, interp, True)[0], equal_nan=True), interp
        saved = body._get_backplane_imgs_for_saving(['RA', 'DEC', 'EMISSION'])
        assert np.array_equal(saved['EMISSION'], eng.backplanes_img(['RA', 'DEC', 'EMISSION'])['EMISSION'], equal_nan=True)
        # ---- the point forms: floats for a scalar point like the reference's (body.py:2295, 2617), arrays for arrays; a point
        # above the surface or planetocentric input is the reference's own scalar path (`alt` there is not the backplanes' adjustment)
        ref = eng.backplanes_map(['PHASE', 'INCIDENCE', 'EMISSION'], np.array([[150.0, 12.5]]), np.array([[-3.0, 40.0]]))
        got = body.illumination_angles_from_lonlat(150.0, -3.0)
        assert all(type(v) is float for v in got) and got == tuple(float(ref[n][0, 0]) for n in ('PHASE', 'INCIDENCE', 'EMISSION'))
        assert f'{got[0]:.1f}'  # (what the (1, 1) arrays of round 5 broke)
        arr = body.illumination_angles_from_lonlat(np.array([150.0, 12.5]), np.array([-3.0, 40.0]))
        assert all(a.shape == (2,) for a in arr) and np.array_equal(arr[2], ref['EMISSION'][0], equal_nan=True)
        assert body.illumination_angles_from_lonlat(np.zeros((0, 3)), np.zeros((0, 3)))[0].shape == (0, 3)
        assert body.illumination_angles_from_lonlat(150.0, -3.0, alt=12.0) == ('the reference\'s own', 150.0, -3.0, 12.0, False)
        assert body.illumination_angles_from_lonlat(150.0, -3.0, planetocentric=True) == ('the reference\'s own', 150.0, -3.0, 0.0, True)
        ra0, dec0 = float(np.nanmean(saved['RA'])), float(np.nanmean(saved['DEC']))
        for vis in (True, False):
            q = eng.radec_query(np.array([ra0 + 0.004, ra0]), np.array([dec0, dec0 + 0.003]), ring_only_visible=vis)
            one = body.ring_plane_coordinates(ra0 + 0.004, dec0, only_visible=vis)
            assert all(type(v) is float for v in one) and np.array_equal(one, q[2:5, 0], equal_nan=True)
            many = body.ring_plane_coordinates(np.array([ra0 + 0.004, ra0]), np.array([dec0, dec0 + 0.003]), only_visible=vis)
            assert all(np.array_equal(m, q[2 + i], equal_nan=True) for i, m in enumerate(many))
        # ---- one context per device for every bound body (the native BodyXY's), a body keeps its geometry block only
        from planetmapper_amd.body_xy import _shared_engine

        assert body._hip() is _shared_engine(0) and '_hip_engine' not in body.__dict__ and '_hip_geometry' in body.__dict__
        other = Bound(g, target='JUPITER', target_id=599, observer='EARTH', frame='IAU_JUPITER')
        assert other._hip() is body._hip()
    finally:
        eng.close()
