"""
The value tables of the reference's own point-transform tests (tests/test_body.py, geometry
Body('Jupiter', observer='HST', utc='2005-01-01T00:00:00')), replayed through the BodyXY methods
of the same names: lonlat <-> radec / angular / km incl. altitudes, `not_visible_nan`, planetocentric
variants; limb coordinates of sky points; visibility / illumination tests; illumination angles,
azimuth, local solar time, ring-plane coordinates, radial velocity, distance; graphic <-> centric.
Tables: tests/golden/kat_body_transforms.json (lifted by tests/golden/make_body_kat_fixtures.py; rows
with custom angular origins / rotations, which this engine does not take, are left out).

The illumination / azimuth / radial-velocity / distance tables of tests/test_body.py are only
consistent with the reference's golden FITS files at the reference's own tolerance (7e-6 deg in
incidence / emission, 9e-8 deg in phase - an older state of kernels or toolkit behind those literals);
the point functions themselves agree with the golden MAP planes to 2e-10 deg
(`test_point_api_equals_the_golden_map_planes`), so those tables are held to the reference's rule only.

Run twice: on the CPU with the oracle-backed engine double (pins the ORACLE to the reference's
numbers) and, `-m gpu`, on the HIP engine (pins the product). Comparison: the reference's own rule
(np.allclose defaults: rtol 1e-5, atol 1e-8) and a 1000x tighter one the restatement actually meets.
"""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

with open(os.path.join(GOLDEN, 'kat_body_transforms.json'), encoding='utf-8') as _f:
    KAT = json.load(_f)


def _v(x):
    """JSON value -> float / tuple (strings 'nan' / 'inf' stand for the non-finite literals)"""
    if isinstance(x, str):
        return float(x) if x in ('nan', 'inf', '-inf') else x
    if isinstance(x, list):
        return tuple(_v(e) for e in x)
    return x


def _close(got, want, rtol=1e-5, atol=1e-8, tight=True):
    got = np.asarray(got, dtype=float)
    want = np.asarray(want, dtype=float)
    assert np.allclose(got, want, rtol=rtol, atol=atol, equal_nan=True), (got, want)
    if tight and atol < 1e-4:
        # forward direction of a table (expected values printed to 16-17 digits): 1000x tighter than
        # the reference's rule. (Rows checked with a loose atol are round trips of printed values.)
        assert np.allclose(got, want, rtol=1e-8, atol=max(1e-9, atol), equal_nan=True), (got, want)


def _make_body(jupiter, engine):
    from planetmapper_amd import BodyXY

    return BodyXY('Jupiter', '2005-01-01T00:00:00', observer='HST', geometry=jupiter, engine=engine)


@pytest.fixture()
def oracle_body(jupiter):
    from oracle_engine import OracleEngine

    return _make_body(jupiter, OracleEngine())


def _c(body, lonlat, planetocentric):
    return body.graphic2centric_lonlat(*lonlat) if planetocentric else lonlat


def replay(body):
    n = 0
    t = KAT['test_lonlat2radec']
    for lonlat, radec in map(_v, t['pairs']):
        for pc in (False, True):
            _close(body.lonlat2radec(*_c(body, lonlat, pc), not_visible_nan=False, planetocentric=pc), radec)
            n += 1
    for (lon, lat, alt), expected in map(_v, t['pairs_with_alts']):
        _close(body.lonlat2radec(lon, lat, alt=alt, not_visible_nan=False), expected)
    for lonlat, nvn_true, nvn_false in map(_v, t['not_visible_nan_coordinates']):
        _close(body.lonlat2radec(*lonlat, not_visible_nan=True), nvn_true)
        _close(body.lonlat2radec(*lonlat, not_visible_nan=False), nvn_false)
        n += 2
    t = KAT['test_radec2lonlat']
    for radec, lonlat in map(_v, t['pairs']):
        for pc in (False, True):
            _close(body.radec2lonlat(*radec, planetocentric=pc), _c(body, lonlat, pc))
        if all(np.isfinite(radec)):
            _close(body.lonlat2radec(*lonlat), radec)
        n += 3
    for (ra, dec, alt), expected in map(_v, t['pairs_with_alts']):
        _close(body.radec2lonlat(ra, dec, alt=alt), expected)
    for (x, y), kw, radec in map(_v, KAT['test_angular_radec']['pairs']):
        if kw:
            continue  # custom origin / rotation of the angular frame: not part of this engine's API
        _close(body.angular2radec(x, y), radec)
        _close(body.radec2angular(*radec), (x, y), atol=1e-4)
        n += 2
    t = KAT['test_angular_lonlat']
    for (x, y), kw, lonlat in map(_v, t['pairs']):
        if kw:
            continue
        for pc in (False, True):
            want = _c(body, lonlat, pc)
            _close(body.angular2lonlat(x, y, planetocentric=pc), want, atol=1e-3)
            if np.isfinite(lonlat[0]):
                _close(body.lonlat2angular(*want, planetocentric=pc), (x, y), atol=1e-4)
            n += 2
    for a in map(_v, t['inputs']):
        assert not np.isfinite(body.angular2lonlat(*a)).any() and not np.isfinite(body.lonlat2angular(*a)).any()
    for (lon, lat, alt), expected in map(_v, t['pairs_with_alts']):
        _close(body.lonlat2angular(lon, lat, alt=alt, not_visible_nan=False), expected)
    for (x, y, alt), expected in map(_v, t['pairs_with_alts_2']):
        _close(body.angular2lonlat(x, y, alt=alt), expected)
    for lonlat, nvn_true, nvn_false in map(_v, t['not_visible_nan_coordinates']):
        _close(body.lonlat2angular(*lonlat, not_visible_nan=True), nvn_true)
        _close(body.lonlat2angular(*lonlat, not_visible_nan=False), nvn_false)
        n += 2
    t = KAT['test_km_radec']
    for km, radec in map(_v, t['pairs']):
        _close(body.km2radec(*km), radec)
        _close(body.radec2km(*radec), km, atol=1e-3)
    for a in map(_v, t['inputs']):
        assert not np.isfinite(body.km2radec(*a)).any() and not np.isfinite(body.radec2km(*a)).any()
    t = KAT['test_km_lonlat']
    for km, lonlat in map(_v, t['pairs']):
        for pc in (False, True):
            want = _c(body, lonlat, pc)
            _close(body.km2lonlat(*km, planetocentric=pc), want)
            _close(body.lonlat2km(*want, planetocentric=pc), km, atol=1e-3)
            n += 2
    assert np.isnan(body.km2lonlat(100000000, 0)).all()
    for a in map(_v, t['inputs']):
        assert not np.isfinite(body.km2lonlat(*a)).any() and not np.isfinite(body.lonlat2km(*a)).any()
    for (lon, lat, alt), expected in map(_v, t['pairs_with_alts']):
        _close(body.lonlat2km(lon, lat, alt=alt, not_visible_nan=False), expected, atol=1e-5)
    for (x, y, alt), expected in map(_v, t['pairs_with_alts_2']):
        _close(body.km2lonlat(x, y, alt=alt), expected)
    for lonlat, nvn_true, nvn_false in map(_v, t['not_visible_nan_coordinates']):
        _close(body.lonlat2km(*lonlat, not_visible_nan=True), nvn_true, atol=1e-5)
        _close(body.lonlat2km(*lonlat, not_visible_nan=False), nvn_false, atol=1e-5)
    for (x, y), kw, km in map(_v, KAT['test_km_angular']['pairs']):
        if kw:
            continue
        _close(body.angular2km(x, y), km, atol=1e-3)
        _close(body.km2angular(*km), (x, y), atol=1e-3)
    for (ra, dec), (lon_e, lat_e, dist_e) in map(_v, KAT['test_limb_coordinates_from_radec']['args']):
        for pc in (False, True):
            lon_w, lat_w = body.graphic2centric_lonlat(lon_e, lat_e) if pc else (lon_e, lat_e)
            lon, lat, dist = body.limb_coordinates_from_radec(ra, dec, planetocentric=pc)
            for got, want in ((lon, lon_w), (lat, lat_w), (dist, dist_e)):
                assert np.allclose(got, want, rtol=1e-5, equal_nan=True), (ra, dec, got, want)
                assert np.allclose(got, want, rtol=1e-8, atol=1e-6, equal_nan=True), (ra, dec, got, want)
            n += 1
    t = KAT['test_if_lonlat_visible']
    for lonlat, visible in map(_v, t['pairs']):
        for pc in (False, True):
            assert bool(body.test_if_lonlat_visible(*_c(body, lonlat, pc), planetocentric=pc)) == visible, lonlat
            n += 1
    for (lon, lat, alt), visible in map(_v, t['pairs_with_alts']):
        assert bool(body.test_if_lonlat_visible(lon, lat, alt=alt)) == visible, (lon, lat, alt)
        use = body.graphic2centric_lonlat(lon, lat, alt=alt)
        assert bool(body.test_if_lonlat_visible(*use, alt=alt, planetocentric=True)) == visible, (lon, lat, alt)
        n += 2
    for lonlat, angles in map(_v, KAT['test_illimination_angles_from_lonlat']['args']):
        for pc in (False, True):
            _close(body.illumination_angles_from_lonlat(*_c(body, lonlat, pc), planetocentric=pc), angles, tight=False)
            n += 1
    for lonlat, angle in map(_v, KAT['test_azimuth_angle_from_lonlat']['args']):
        for pc in (False, True):
            got = body.azimuth_angle_from_lonlat(*_c(body, lonlat, pc), planetocentric=pc)
            assert np.allclose(got, angle, equal_nan=True)
            n += 1
    for lon, lst_expected, s_expected in map(_v, KAT['test_local_solar_time']['args']):
        assert np.isclose(body.local_solar_time_from_lon(lon), lst_expected, equal_nan=True)
        assert body.local_solar_time_string_from_lon(lon) == s_expected
        n += 1
    for (lon, lat), illuminated in map(_v, KAT['test_if_lonlat_illuminated']['pairs']):
        for pc in (False, True):
            use = body.graphic2centric_lonlat(lon, lat) if pc else (lon, lat)
            assert bool(body.test_if_lonlat_illuminated(*use, planetocentric=pc)) == illuminated, (lon, lat)
            n += 1
    for (ra, dec, only_visible), coords in map(_v, KAT['test_ring_plane_coordinates']['args']):
        got = body.ring_plane_coordinates(ra, dec, only_visible=only_visible)
        assert np.allclose(got, coords, equal_nan=True), (ra, dec, only_visible, got, coords)
        # (tight rule per column: km to 1e-3 km; the longitude of a ring point r km from the centre moves by
        #  deg(1e-3 / r) for that)
        if np.isfinite(coords[0]):
            assert abs(got[0] - coords[0]) <= 1e-3 + 1e-9 * abs(coords[0]) and abs(got[2] - coords[2]) <= 1e-3 + 1e-9 * abs(coords[2]), (ra, dec, got, coords)
            assert abs(got[1] - coords[1]) <= np.rad2deg(2e-3 / abs(coords[0])) + 1e-8, (ra, dec, got, coords)
        n += 1
    _close(body.ring_plane_coordinates(196.3, -5.5), (9305877.091704229, 145.3644753085151, 810435703.2382222), atol=1e-5)
    for lonlat, x in map(_v, KAT['test_radial_velocity_from_lonlat']['args']):
        for pc in (False, True):
            _close(body.radial_velocity_from_lonlat(*_c(body, lonlat, pc), planetocentric=pc), x)
            n += 1
    for lonlat, x in map(_v, KAT['test_distance_from_lonalt']['args']):
        for pc in (False, True):
            got = body.distance_from_lonlat(*_c(body, lonlat, pc), planetocentric=pc)
            assert np.allclose(got, x, equal_nan=True) and np.allclose(got, x, rtol=1e-12, atol=1e-4, equal_nan=True)
            n += 1
    t = KAT['test_graphic_centric_lonlat']
    for graphic, centric in map(_v, t['pairs']):
        _close(body.graphic2centric_lonlat(*graphic), centric)
        _close(body.centric2graphic_lonlat(*centric), graphic)
        n += 2
    for a, b in map(_v, t['pairs_2']):
        _close(body.graphic2centric_lonlat(*a), b)
        _close(body.centric2graphic_lonlat(*a), b)
    return n


def test_point_api_equals_the_golden_map_planes(oracle_body):
    """illumination_angles_from_lonlat etc. at the cells of the golden 30 deg map == the golden planes"""
    gold = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_linear.npz'))
    lon, lat = gold['LON-GRAPHIC'], gold['LAT-GRAPHIC']
    ph, inc, em = oracle_body.illumination_angles_from_lonlat(lon, lat)
    assert np.nanmax(np.abs(ph - gold['PHASE'])) <= 1e-12
    assert np.nanmax(np.abs(inc - gold['INCIDENCE'])) <= 1e-9 and np.nanmax(np.abs(em - gold['EMISSION'])) <= 1e-9
    assert np.nanmax(np.abs(oracle_body.distance_from_lonlat(lon, lat) - gold['DISTANCE'])) <= 1e-5
    assert np.nanmax(np.abs(oracle_body.radial_velocity_from_lonlat(lon, lat) - gold['RADIAL-VELOCITY'])) <= 1e-9


def test_reference_value_tables_on_the_oracle(oracle_body):
    assert replay(oracle_body) > 150


@pytest.mark.gpu
def test_reference_value_tables_on_the_hip_engine(jupiter):
    from planetmapper_amd.engine import Engine

    eng = Engine(0)
    try:
        assert replay(_make_body(jupiter, eng)) > 150
    finally:
        eng.close()
