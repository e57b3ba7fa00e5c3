"""
Host-side behaviour of the reference-compatible API (no GPU: the engine is replaced by
tests/oracle_engine.OracleEngine). Mirrors the reference's own tests:
tests/test_body_xy.py (disc params :140-266, map_img :1085-1327, backplanes :1990-2170,
cache invalidation :2495-2590) and tests/test_observation.py (mapped data).
"""

import numpy as np
import pytest

from oracle_engine import OracleEngine
from planetmapper_amd import BackplaneNotFoundError, BodyXY, Observation
from planetmapper_amd._lib import UnsupportedError

nan = np.nan


@pytest.fixture()
def body(jupiter):
    return BodyXY('Jupiter', '2005-01-01T00:00:00', observer='HST', geometry=jupiter, engine=OracleEngine())


def test_constructor_and_disc_params(jupiter):
    """tests/test_body_xy.py:78-138, body_xy.py:186-199, 772-803"""
    with pytest.raises(ValueError):
        BodyXY('jupiter', geometry=jupiter, nx=3, sz=5, engine=OracleEngine())
    with pytest.raises(ValueError):
        BodyXY('jupiter', engine=OracleEngine())  # neither geometry nor scenario
    b = BodyXY('jupiter', geometry=jupiter, engine=OracleEngine())
    assert b.get_img_size() == (0, 0)
    assert b.get_disc_params() == (0.0, 0.0, 10.0, 0.0)
    assert b.get_disc_method() == 'zero'
    b = BodyXY('jupiter', geometry=jupiter, nx=21, ny=31, engine=OracleEngine())
    assert b.get_disc_params() == (10.0, 15.0, 9.0, 0.0)  # centre_disc docstring example
    assert b.get_disc_method() == 'centre_disc'
    b.set_disc_params(x0=1.5, rotation=123.456)
    assert b.get_x0() == 1.5
    assert b.get_rotation() == 123.45600000000002  # stored modulo 2 pi (golden header DISC ROT)
    b.set_rotation(-90)
    assert b.get_rotation() == pytest.approx(270)
    b.adjust_disc_params(dx=1, dy=-1, dr=2, drotation=10)
    assert b.get_disc_params() == pytest.approx((2.5, 14.0, 11.0, 280.0))
    for bad in (nan, np.inf):
        for setter in (b.set_x0, b.set_y0, b.set_r0, b.set_rotation):
            with pytest.raises(ValueError):
                setter(bad)
    for bad in (0, -1.5):
        with pytest.raises(ValueError):
            b.set_r0(bad)
    with pytest.raises(ValueError):
        b.set_img_size(-1, 5)
    b.set_plate_scale_arcsec(0.1)
    assert b.get_plate_scale_arcsec() == pytest.approx(0.1)
    assert b.get_r0() == pytest.approx(b.target_diameter_arcsec / 0.2)
    b.set_plate_scale_km(1000)
    assert b.get_plate_scale_km() == pytest.approx(1000)


def test_empty_image_raises(body):
    """body_xy.py:3166-3168"""
    with pytest.raises(ValueError):
        body.get_lon_img()
    with pytest.raises(ValueError):
        body.get_backplane_img('EMISSION')


def test_backplane_registry(body):
    """tests/test_body_xy.py:1990-2118; body_xy.py:2492-2584"""
    assert len(body.backplanes) == 26
    assert list(body.backplanes)[:3] == ['LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC']
    assert body.backplanes['LON-GRAPHIC'].description == 'Planetographic longitude, positive W [deg]'
    assert body.standardise_backplane_name('  emission ') == 'EMISSION'
    assert body.get_backplane(' dec ').name == 'DEC'
    with pytest.raises(BackplaneNotFoundError):
        body.get_backplane('<<< test >>>')
    with pytest.raises(ValueError):
        body.register_backplane('emission', 'dup', lambda: None, lambda **kw: None)
    body.set_img_size(4, 3)
    body.register_backplane(
        'custom', 'ones', lambda: np.ones((3, 4)), lambda **kw: np.ones((2, 2)) * kw.get('degree_interval', 1)
    )
    assert np.array_equal(body.get_backplane_img('Custom'), np.ones((3, 4)))
    assert np.array_equal(body.get_backplane_map('CUSTOM', degree_interval=90), np.full((2, 2), 90.0))
    assert 'CUSTOM: ones' in body.backplane_summary_string()


def test_emission_img_and_map_kat(body):
    """tests/test_body_xy.py:2120-2154: disc (2, 1, 1.5, 45.678), 4x3 image"""
    body.set_img_size(4, 3)
    body.set_disc_params(2, 1, 1.5, 45.678)
    img = body.get_backplane_img('EMISSION')
    assert img.shape == (3, 4)
    assert np.isfinite(img).sum() == np.isfinite(body.get_lon_img()).sum() > 0
    m = body.get_backplane_map('EMISSION', degree_interval=90)
    assert m.shape == (2, 4) and np.isfinite(m).all()
    assert np.array_equal(m, body.get_emission_angle_map(degree_interval=90))


def test_readonly_views_and_copies(body):
    """tests/test_body_xy.py:2156-2170 (test_backplane_readonly); base.py:115-138"""
    body.set_img_size(7, 10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    for name, bp in body.backplanes.items():
        img = bp.get_img()
        assert not img.flags.writeable, name
        with pytest.raises(ValueError):
            img[0, 0] = 1
        mp = bp.get_map(degree_interval=30)
        assert not mp.flags.writeable, name
        cp = body.get_backplane_img(name)
        assert cp.flags.writeable and cp is not img
        cp[:] = 0  # modifying the copy must not touch the cache
        assert np.array_equal(bp.get_img(), img, equal_nan=True)
        cm = body.get_backplane_map(name, degree_interval=30)
        assert cm.flags.writeable
    lons, lats, xx, yy, tr, info = body.generate_map_coordinates(degree_interval=30)
    assert not lons.flags.writeable and info == {
        'projection': 'rectangular', 'degree_interval': 30, 'xlim': None, 'ylim': None,
    }  # fmt: skip


def test_golden_planes_through_the_api(body, jupiter_info):
    """The reference-compatible getters return the golden FITS planes (atol 1e-6)."""
    import os

    from conftest import GOLDEN

    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav.npz'))
    body.set_img_size(7, 10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    for name in body.backplanes:
        assert np.allclose(body.get_backplane_img(name), gold[name], rtol=1e-5, atol=1e-6, equal_nan=True), name
    gold_alt = np.load(os.path.join(GOLDEN, 'golden_test_nav_alt.npz'))
    for name in ('LON-GRAPHIC', 'EMISSION', 'RING-RADIUS', 'DISTANCE'):
        a = body.get_backplane_img(name, alt=34567.8912)
        assert np.allclose(a, gold_alt[name], rtol=1e-5, atol=1e-6, equal_nan=True), name
    assert body._alt_adjustment == 0.0
    gmap = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_nearest_alt.npz'))
    for name in ('PIXEL-X', 'EMISSION', 'LIMB-DISTANCE'):
        a = body.get_backplane_map(name, degree_interval=30, alt=34567.8912)
        assert np.allclose(a, gmap[name], rtol=1e-5, atol=1e-6, equal_nan=True), name


def test_caching_and_invalidation(body):
    """
    tests/test_body_xy.py:2495-2590: image planes are cached until a disc parameter or
    the image size changes; disc-independent maps survive; x/y maps do not.
    """
    eng = body._engine
    body.set_img_size(7, 10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    a = body.get_lon_img()
    n = len(eng.calls)
    assert body.get_lat_img() is not None and len(eng.calls) == n  # same family: one launch
    assert body.get_lon_img() is a
    em = body.get_emission_angle_map(degree_interval=30)
    xm = body.get_x_map(degree_interval=30)
    n = len(eng.calls)
    body.get_emission_angle_map(degree_interval=30)
    body.get_x_map(degree_interval=30)
    assert len(eng.calls) == n
    mutations = [
        lambda: body.set_x0(3.0), lambda: body.set_y0(2.0), lambda: body.set_r0(4.4),
        lambda: body.set_rotation(10.0), lambda: body.set_img_size(8, 9),
    ]  # fmt: skip
    for mutate in mutations:
        before = body.get_lon_img()
        mutate()
        n = len(eng.calls)
        after = body.get_lon_img()
        assert len(eng.calls) == n + 1 and after is not before
        # stable (disc independent) map cache survives, x/y map is recomputed
        n = len(eng.calls)
        assert body.get_emission_angle_map(degree_interval=30) is em
        assert len(eng.calls) == n
        body.get_x_map(degree_interval=30)
        assert len(eng.calls) == n + 1
    # altitude is part of the cache key (body.py:255-272)
    a0 = body.get_backplane_img('EMISSION')
    a1 = body.get_backplane_img('EMISSION', alt=1000.0)
    assert not np.array_equal(a0, a1, equal_nan=True)
    n = len(eng.calls)
    body.get_backplane_img('EMISSION', alt=1000.0)
    body.get_backplane_img('EMISSION')
    assert len(eng.calls) == n
    assert xm is not None
    with pytest.raises(ValueError):
        body.get_backplane_img('EMISSION', alt=nan)


def test_prefetch_uses_one_launch(body):
    body.set_img_size(7, 10)
    eng = body._engine
    n = len(eng.calls)
    body.prefetch_backplane_imgs()
    assert len(eng.calls) == n + 1 and len(eng.calls[-1][1]) == 26
    for name in body.backplanes:
        body.get_backplane_img(name)
    assert len(eng.calls) == n + 1


def test_generate_map_coordinates(body):
    """tests/test_body_xy.py:1551-1700; body_xy.py:2899-2929, 2982-3012"""
    lons, lats, xx, yy, tr, info = body.generate_map_coordinates(degree_interval=90)
    assert np.array_equal(lons, [[315, 225, 135, 45], [315, 225, 135, 45]])  # W-positive: reversed
    assert np.array_equal(lats, [[-45] * 4, [45] * 4])
    lons, lats, *_ = body.generate_map_coordinates(degree_interval=90, xlim=(100, 250), ylim=(0, 90))
    assert np.array_equal(lons, [[225, 135]]) and np.array_equal(lats, [[45, 45]])
    lons, lats, *_ = body.generate_map_coordinates('manual', lon_coords=[0, 10, 20], lat_coords=[-5, 5])
    assert lons.shape == (2, 3) and np.array_equal(lats[:, 0], [-5, 5])
    lons, lats, *_ = body.generate_map_coordinates(
        'manual', lon_coords=np.array([[0, np.inf]]), lat_coords=np.array([[1.0, 2.0]])
    )
    assert np.isnan(lons[0, 1])
    with pytest.raises(ValueError):
        body.generate_map_coordinates('manual')
    with pytest.raises(ValueError):
        body.generate_map_coordinates('manual', lon_coords=[1, 2], lat_coords=[[1, 2]])
    with pytest.raises(ValueError):
        body.generate_map_coordinates('manual', lon_coords=np.zeros((2, 2)), lat_coords=np.zeros((2, 3)))
    with pytest.raises(ValueError):
        body.generate_map_coordinates('manual', lon_coords=np.zeros((1, 2, 2)), lat_coords=np.zeros((1, 2, 2)))
    with pytest.raises(UnsupportedError):
        body.generate_map_coordinates('+proj=ortho +axis=wnu +type=crs', projection_x_coords=[0, 1])
    assert body.get_lon_map(degree_interval=90).shape == (2, 4)
    assert body.get_lon_map(projection='manual', lon_coords=[-10, 370], lat_coords=[0])[0].tolist() == [350, 10]


def test_lonlat_grids_are_the_interleaved_map_cut_once(body):
    """
    body_xy.py:3290-3300: `_get_lonlat_map` is (n0, n1, 2), longitudes modulo 360, non-finite -> NaN, read-only, cached for
    good. The engine takes the two contiguous grids (`_get_lonlat_grids`): the same values, cut once per grid, and what the
    map planes are computed from.
    """
    for kw in (dict(degree_interval=30), dict(degree_interval=45, xlim=(100, 250), ylim=(0, 90)), dict(projection='orthographic', size=7),
               dict(projection='manual', lon_coords=np.array([[-10.0, 370.0, np.inf]]), lat_coords=np.array([[0.0, 1.0, 2.0]]))):
        lon, lat = body._get_lonlat_grids(**kw)
        ll = body._get_lonlat_map(**kw)
        assert ll.shape == lon.shape + (2,) and lon.flags.c_contiguous and lat.flags.c_contiguous
        assert np.array_equal(ll[..., 0], lon, equal_nan=True) and np.array_equal(ll[..., 1], lat, equal_nan=True)
        assert not lon.flags.writeable and not lat.flags.writeable and not ll.flags.writeable
        assert body._get_lonlat_grids(**kw)[0] is lon and body._get_lonlat_map(**kw) is ll  # cached
        fin = np.isfinite(lon)
        assert ((lon[fin] >= 0) & (lon[fin] < 360)).all() and not np.isinf(ll).any()
        assert np.array_equal(body.get_lon_map(**kw), lon, equal_nan=True) or kw.get('projection') == 'orthographic'
    lon, _ = body._get_lonlat_grids(projection='manual', lon_coords=np.array([[-10.0, 370.0, np.inf]]), lat_coords=np.array([[0.0, 1.0, 2.0]]))
    assert lon[0, :2].tolist() == [350.0, 10.0] and np.isnan(lon[0, 2])
    # the grids are their own arrays: the caller's coordinates are not frozen or changed by the cache
    mine = np.array([[0.0, 90.0]])
    body._get_lonlat_grids(projection='manual', lon_coords=mine, lat_coords=mine)
    mine[0, 0] = 5.0
    assert mine.flags.writeable


IMAGE = np.array(
    [
        [0.0, 100.0, -1.0, 2.2, 3.3, 4.4],
        [0.0, 75.0, 999.0, 50.0, 1.0, 123.456789],
        [0.0, 25.0, 0.0, 123.45, nan, 3],
        [0.0, 0.123, 0.0, 3.0, 0.1, nan],
        [100.0, -100.0, 100.0, -100.0, 100.0, nan],
    ]
)


def test_map_img_kats(body):
    """tests/test_body_xy.py:1086-1200: 6x5 image, disc (2.75, 1.3, 2.3, 45.678), 45 deg map"""
    body.set_img_size(6, 5)
    body.set_disc_params(2.75, 1.3, 2.3, 45.678)
    # fmt: off
    expected = {
        'nearest': [[nan, nan, 100.0, 100.0, -1.0, nan, nan, nan], [nan, nan, nan, 75.0, 999.0, 3.3, 3.3, nan], [nan, nan, nan, 0.0, 123.45, nan, 123.456789, nan], [nan, nan, nan, 3.0, 3.0, 0.1, nan, nan]],
        'linear': [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 61.591824124152424, 488.0893412811879, 4.181692402514696, nan, nan], [nan, nan, nan, 3.678385742930187, 94.03788871233297, nan, nan, nan], [nan, nan, nan, -25.28910210942658, -1.6502703714050462, nan, nan, nan]],
    }
    no_propagation = [[nan, nan, 83.42502054006614, 61.410255547165704, 1.0972142916279704, nan, nan, nan], [nan, nan, nan, 61.591824124152424, 488.0893412811879, 4.181692402514696, 3.8032713799190443, nan], [nan, nan, nan, 3.678385742930187, 94.03788871233297, 35.721226497463014, 94.00305287602345, nan], [nan, nan, nan, -25.28910210942658, -1.6502703714050462, 4.265385156596395, nan, nan]]
    # fmt: on
    for interp, exp in expected.items():
        got = body.map_img(IMAGE, degree_interval=45, interpolation=interp)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), interp
    assert np.array_equal(
        body.map_img(IMAGE, degree_interval=45),
        body.map_img(IMAGE, degree_interval=45, interpolation='linear', spline_smoothing=0.0, propagate_nan=True),
        equal_nan=True,
    )
    for alias in (1, (1, 1)):
        assert np.array_equal(
            body.map_img(IMAGE, degree_interval=45, interpolation=alias),
            body.map_img(IMAGE, degree_interval=45), equal_nan=True,
        )  # fmt: skip
    assert np.isnan(body.map_img(IMAGE * nan, degree_interval=45)).all()
    got = body.map_img(IMAGE, degree_interval=45, propagate_nan=False)
    assert np.allclose(got, no_propagation, rtol=1e-5, atol=1e-8, equal_nan=True)
    # cube input -> stack of maps (body_xy.py:1571-1585)
    cube = np.stack([IMAGE, IMAGE * 2])
    out = body.map_img(cube, degree_interval=45)
    assert out.shape == (2, 4, 8)
    assert np.allclose(out[1], np.array(expected['linear']) * 2, equal_nan=True)
    with pytest.raises(ValueError):
        body.map_img(IMAGE[:, :-1], degree_interval=45)
    with pytest.raises(ValueError):
        body.map_img(IMAGE, degree_interval=45, interpolation='<<<test>>>')
    # fmt: off
    splines = {
        'quadratic': [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 47.43961193970507, 780.1933190874719, -11.958641161828965, nan, nan], [nan, nan, nan, -40.33639788223132, 106.33548747800452, nan, nan, nan], [nan, nan, nan, -35.84554405305129, -19.35757229218872, nan, nan, nan]],
        'cubic': [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 38.17050096080083, 837.0682797065551, -40.810161294299334, nan, nan], [nan, nan, nan, -77.21287210436617, 103.88323214798433, nan, nan, nan], [nan, nan, nan, -29.994884067130222, -35.81550582449343, nan, nan, nan]],
        (1, 2): [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 48.82728713390978, 584.7164003757379, -0.9895987798646678, nan, nan], [nan, nan, nan, -0.625402661173368, 99.24054961575526, nan, nan, nan], [nan, nan, nan, -33.19407454333914, -8.380623602166663, nan, nan, nan]],
    }
    # fmt: on
    for interp, exp in splines.items():
        got = body.map_img(IMAGE, degree_interval=45, interpolation=interp)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), interp
    for alias, name in ((2, 'quadratic'), (3, 'cubic'), ((2, 2), 'quadratic'), ((3, 3), 'cubic')):
        assert np.array_equal(
            body.map_img(IMAGE, degree_interval=45, interpolation=alias),
            body.map_img(IMAGE, degree_interval=45, interpolation=name), equal_nan=True,
        )  # fmt: skip
    assert np.isnan(body.map_img(IMAGE * nan, degree_interval=45, interpolation='cubic')).all()
    # spline_smoothing > 0: FITPACK smoothing splines (tests/test_body_xy.py:1194-1231)
    # fmt: off
    smoothings = {
        0: [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 61.591824124152424, 488.0893412811879, 4.181692402514696, nan, nan], [nan, nan, nan, 3.678385742930187, 94.03788871233297, nan, nan, nan], [nan, nan, nan, -25.28910210942658, -1.6502703714050462, nan, nan, nan]],
        1: [[nan, nan, nan, nan, nan, nan, nan, nan], [nan, nan, nan, 61.78601266274162, 487.8146612006081, 4.182606695966389, nan, nan], [nan, nan, nan, 3.7096818751834397, 94.00272821072134, nan, nan, nan], [nan, nan, nan, -25.261262121302103, -1.6266437752937738, nan, nan, nan]],
    }
    # fmt: on
    for sm, exp in smoothings.items():
        got = body.map_img(IMAGE, interpolation='linear', degree_interval=45, spline_smoothing=sm)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), sm
    for sm, exp in _smoothing_kats():
        got = body.map_img(IMAGE, interpolation='linear', degree_interval=45, spline_smoothing=sm)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), sm
    factors = [1, 2.345, -10, 3456.789, np.nan]
    kwargs = dict(interpolation='cubic', degree_interval=45, spline_smoothing=1)
    mapped = body.map_img([IMAGE * f for f in factors], **kwargs)
    for f, m in zip(factors, mapped):
        assert np.array_equal(m, body.map_img(IMAGE * f, **kwargs), equal_nan=True), f
    with pytest.raises(ValueError):
        body.map_img(IMAGE, degree_interval=45, interpolation='cubic', spline_smoothing=-1.0)


def _smoothing_kats():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_map_img_smoothing.json'), encoding='utf-8') as f:
        return [(k['smoothing'], np.array(k['expected'], dtype=float)) for k in json.load(f)['cases']]


def _smooth_kats(name):
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_map_img_smooth.json'), encoding='utf-8') as f:
        kats = json.load(f)[name]
    return [(k['kwargs'], np.array(k['expected'], dtype=float)) for k in kats]  # null -> NaN


def smooth_test_image():
    """the 90 x 120 input of tests/test_body_xy.py:1329-1348 (recipe: sin x cos pattern, every
    other row scaled, two spikes, a NaN pixel, a NaN column and a constant row)"""
    xs = np.linspace(0, 1, 90)
    ys = np.linspace(0, 1, 120)
    image = np.sin(xs[None, :] * 10 * np.pi) * np.cos(ys[:, None] * 5 * np.pi)
    image[::2, :] *= 1.5
    image[50, 30] = 3
    image[60, 40] = -2
    image[45, 35] = np.nan
    image[:, 22] = np.nan
    image[40, :] = 1
    return image


def test_map_img_smooth_kats(body):
    """'smooth' interpolation against the reference's own expected values
    (tests/test_body_xy.py:1290-1327 and 1329-1372)"""
    body.set_img_size(6, 5)
    body.set_disc_params(2.75, 1.3, 2.3, 45.678)
    for kw, exp in _smooth_kats('map_img_6x5'):
        got = body.map_img(IMAGE, degree_interval=45, interpolation='smooth', **kw)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), kw
    # oversampling off == bilinear on the original grid
    assert np.allclose(
        body.map_img(IMAGE, degree_interval=45, interpolation='smooth', smooth_oversample_by=1),
        body.map_img(IMAGE, degree_interval=45, interpolation='linear'), equal_nan=True,
    )  # fmt: skip
    assert np.isnan(body.map_img(IMAGE * nan, degree_interval=45, interpolation='smooth')).all()

    body.set_img_size(90, 120)
    body.set_disc_params(32.1, 50, 12, 98.76)
    image = smooth_test_image()
    for kw, exp in _smooth_kats('map_img_90x120'):
        got = body.map_img(image, degree_interval=45, interpolation='smooth', **kw)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), kw


def test_observation_mapped_data(jupiter):
    """tests/test_observation.py golden: map_rectangular-linear / -nearest primary HDUs"""
    import os

    from conftest import GOLDEN

    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    obs = Observation(data=cube, geometry=jupiter, engine=OracleEngine())
    assert obs.get_img_size() == (7, 10)
    obs.set_disc_params(2.5, 3.1, 3.9, 123.456)
    for interp, name in (
        ('linear', 'map_rectangular_linear'), ('nearest', 'map_rectangular_nearest'),
        ('quadratic', 'map_rectangular_quadratic'), ('cubic', 'map_rectangular_cubic'),
        ('smooth', 'map_rectangular_smooth'),
    ):  # fmt: skip
        gold = np.load(os.path.join(GOLDEN, f'golden_{name}.npz'))['PRIMARY']
        m = obs.get_mapped_data(interpolation=interp, degree_interval=30)
        assert m.shape == (10, 6, 12)
        assert np.allclose(m, gold, rtol=1e-5, atol=1e-6, equal_nan=True)
        m[:] = 0  # a copy: the cache is untouched (observation.py:864-872)
        assert np.allclose(obs.get_mapped_data(interpolation=interp, degree_interval=30), gold, equal_nan=True)
    # map_rectangular-interpolation.fits: interpolation=(1, 3), spline_smoothing=2.34 (FITPACK
    # smoothing); planes 6 and 7 are the ones the reference itself compares loosely because the
    # smoothing of extreme data differs between scipy versions (tests/test_observation.py:1162-1170)
    gold = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_interpolation.npz'))['PRIMARY']
    m = obs.get_mapped_data(interpolation=(1, 3), spline_smoothing=2.34, degree_interval=30)
    assert np.array_equal(np.isnan(m), np.isnan(gold))
    for pl in range(10):
        rtol, atol = {6: (1e-1, 1e-1), 7: (10, 1)}.get(pl, (1e-6, 1e-5))
        assert np.allclose(m[pl], gold[pl], rtol=rtol, atol=atol, equal_nan=True), pl
    n = len(obs._engine.calls)
    obs.get_mapped_data(degree_interval=30)
    assert len(obs._engine.calls) == n  # cached
    obs.set_x0(2.6)
    obs.get_mapped_data(degree_interval=30)
    assert len(obs._engine.calls) > n  # disc change invalidates
    with pytest.raises(TypeError):
        obs.set_img_size(3, 3)
    with pytest.raises(TypeError):
        Observation(data=cube, geometry=jupiter, nx=3, engine=OracleEngine())
    with pytest.raises(ValueError):
        Observation(geometry=jupiter, engine=OracleEngine())
    # a 2D image is a one-plane cube
    obs2 = Observation(data=cube[0], geometry=jupiter, engine=OracleEngine())
    assert obs2.data.shape == (1, 10, 7) and obs2.get_mapped_data(degree_interval=90).shape == (1, 2, 4)


def test_xy_conversions_kat(jupiter):
    """tests/test_body_xy.py:267-486: disc (5, 8, 3, 45), 15 x 10 image"""
    from planetmapper_amd import NotFoundError

    body = BodyXY('Jupiter', geometry=jupiter, nx=15, ny=10, engine=OracleEngine())
    body.set_disc_params(5, 8, 3, 45)
    # xy, radec, lonlat, km, angular
    coordinates = [
        [(0, 0), (196.3684350770821, -5.581107015413806), (nan, nan), (-43515.54503863168, -220566.4464649765), (12.721709080506116, -55.12740601573759)],
        [(5, 8), (196.37198562427025, -5.565793847134351), (153.1235185909613, -3.0887371238645795), (0.0, 0.0), (0.0, 0.0)],
        [(4.1, 7.1), (196.37198562427025, -5.567914131973045), (164.3872136538264, -28.87847195832716), (-12411.924521414994, -27675.679236383432), (0.0, -7.633025448335383)],
        [(1.234, 5.678), (196.37369462098349, -5.572965121633222), (nan, nan), (-64181.931835415264, -83648.1756567178), (-6.1233826374518685, -25.81658829413859)],
    ]  # fmt: skip
    close = lambda a, b, **kw: np.allclose(a, b, equal_nan=True, **kw)  # noqa: E731
    for xy, radec, lonlat, km, angular in coordinates:
        assert close(body.xy2radec(*xy), radec)
        assert close(body.xy2lonlat(*xy), lonlat)
        assert close(body.xy2km(*xy), km, atol=1e-3)
        assert close(body.xy2angular(*xy), angular, atol=1e-6)
        assert close(body.radec2xy(*radec), xy, atol=1e-3)
        assert close(body.km2xy(*km), xy, atol=1e-3)
        assert close(body.angular2xy(*angular), xy, atol=1e-3)
        assert close(body.radec2km(*body.xy2radec(*xy)), km, atol=1e-3)
        assert close(body.km2angular(*km), angular, atol=1e-6)
        if not any(np.isnan(lonlat)):
            assert close(body.lonlat2xy(*lonlat), xy, atol=1e-3)
            assert close(body.radec2lonlat(*radec), lonlat, atol=1e-4)
    assert isinstance(body.xy2radec(0, 0)[0], float)
    ra, dec = body.xy2radec(np.array([0, 5]), 8.0)  # broadcasting -> arrays
    assert ra.shape == (2,) and close((ra[1], dec[1]), coordinates[1][1])
    for a in [(nan, nan), (nan, 0), (0, nan), (np.inf, np.inf)]:
        for fn in (body.xy2radec, body.xy2lonlat, body.xy2km, body.radec2xy, body.lonlat2xy, body.km2xy):
            assert not all(np.isfinite(fn(*a)))
    # altitude semantics: point altitude for lonlat -> xy, surface adjustment for xy -> lonlat
    for (lon, lat, alt), expected in [
        ((42, 23.4, 0), (7.781497231832574, 8.015145501618983)),
        ((42, 23.4, -123.456), (7.776650117803703, 8.014878507462662)),
        ((42, 23.4, 1234.567), (7.829968623728911, 8.017815455484365)),
        ((42, 23.4, nan), (nan, nan)),
    ]:
        assert close(body.lonlat2xy(lon, lat, alt=alt, not_visible_nan=False), expected)
    assert close(body.xy2lonlat(7.781497231832574, 8.015145501618983), (86.30139500952406, 21.109249946237032))
    assert close(
        body.xy2lonlat(7.781497231832574, 8.015145501618983, alt=123456.789), (134.58218536012419, 4.708273802335033)
    )
    with pytest.raises(NotFoundError):
        body.xy2lonlat(0, 0, not_found_nan=False)
    # far side: not visible -> NaN by default, coordinates with not_visible_nan=False
    assert np.isnan(body.lonlat2radec(0, 0)[0]) and np.isfinite(body.lonlat2radec(0, 0, not_visible_nan=False)[0])
    # tests/test_body.py:873-881 (radec2lonlat) through the API
    assert close(body.radec2lonlat(196.372, -5.566), (154.24480750302573, -5.475831082435726))
    # planetocentric round trip
    lc = body.xy2lonlat(4.1, 7.1, planetocentric=True)
    assert close(body.lonlat2xy(*lc, planetocentric=True), (4.1, 7.1), atol=1e-3)


@pytest.mark.parametrize(
    'name,kw',
    [
        ('map_orthographic_1', dict(projection='orthographic', size=10)),
        ('map_orthographic_2', dict(projection='orthographic', lat=90, size=5)),
        ('map_orthographic_3', dict(projection='orthographic', lat=-21.3, lon=-42, size=4)),
        ('map_azimuthal_1', dict(projection='azimuthal', size=10)),
        ('map_azimuthal_2', dict(projection='azimuthal', lat=-90, size=5)),
        ('map_azimuthal_3', dict(projection='azimuthal', lat=42, lon=12.345, size=4)),
    ],
)
def test_pyproj_free_projections_match_golden_maps(body, name, kw):
    """
    tests/test_observation.py:1123-1153: the orthographic / azimuthal golden maps (all
    map-space backplanes and the mapped cube) with the closed-form projection grids.
    """
    import os

    from conftest import GOLDEN

    gold = np.load(os.path.join(GOLDEN, f'golden_{name}.npz'))
    body.set_img_size(7, 10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    lons, lats, xx, yy, tr, info = body.generate_map_coordinates(**kw)
    assert np.array_equal(np.isnan(lons), np.isnan(gold['LON-GRAPHIC']))
    assert info['projection'] == kw['projection'] and info['size'] == kw['size']
    for n in body.backplanes:
        if n in gold.files:
            m = body.get_backplane_map(n, **kw)
            assert np.allclose(m, gold[n], rtol=1e-5, atol=1e-6, equal_nan=True), n
    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    assert np.allclose(body.map_img(cube, **kw), gold['PRIMARY'], rtol=1e-5, atol=1e-6, equal_nan=True)


def test_azimuthal_equal_area_grid(body):
    """`laea` on a sphere: equal areas - the lon/lat cell Jacobian is constant."""
    lons, lats, xx, yy, _, _ = body.generate_map_coordinates('azimuthal equal area', size=41, lat=30, lon=100)
    assert np.isnan(lons[0, 0]) and np.isfinite(lons[20, 20])
    assert lats[20, 20] == pytest.approx(30.0) and (lons[20, 20] % 360) == pytest.approx(100.0)


def test_disc_helpers_kats(jupiter):
    """tests/test_body_xy.py:588-595, 637-763: rotate_north_to_top, scale_img_size, add_img_border,
    add_arcsec_offset and the image limits, with the reference's expected values"""
    body = BodyXY('Jupiter', geometry=jupiter, nx=15, ny=10, engine=OracleEngine())
    body.set_disc_params(0, 0, 1, 0)
    body.rotate_north_to_top()
    assert body.get_rotation() == pytest.approx(24.15516987997688, abs=1e-7)
    assert body.get_disc_method() == 'rotate_north_to_top'

    def fresh():
        b = BodyXY('Jupiter', geometry=jupiter, nx=16, ny=10, engine=OracleEngine())
        b.set_disc_params(3, 4, 5, 6)
        return b

    for factor, size, disc in (
        (1, (16, 10), (3.0, 4.0, 5.0, 6.0)), (2, (32, 20), (6.5, 8.5, 10.0, 6.0)),
        (1.5, (24, 15), (4.75, 6.25, 7.5, 6.0)), (0.5, (8, 5), (1.25, 1.75, 2.5, 6.0)),
    ):  # fmt: skip
        b = fresh()
        b.scale_img_size(factor)
        assert b.get_img_size() == size and np.allclose(b.get_disc_params(), disc)
    with pytest.raises(ValueError):
        fresh().scale_img_size(0.25)
    with pytest.raises(ValueError):
        fresh().scale_img_size(-1)
    b = fresh()
    b.scale_img_size(0.25, allow_rounding=True)
    assert b.get_img_size() == (4, 3) and np.allclose(b.get_disc_params(), (0.375, 0.625, 1.25, 6.0))
    for border, size, disc in ((0, (16, 10), (3, 4, 5, 6)), (2, (20, 14), (5, 6, 5, 6)), (-1, (14, 8), (2, 3, 5, 6))):
        b = fresh()
        b.add_img_border(border)
        assert b.get_img_size() == size and np.allclose(b.get_disc_params(), disc)

    body = BodyXY('Jupiter', geometry=jupiter, nx=15, ny=10, engine=OracleEngine())
    body.set_disc_params(0, 0, 1, 0)
    body.add_arcsec_offset(0, 0)
    assert np.allclose(body.get_disc_params(), (0, 0, 1, 0))
    body.add_arcsec_offset(1, 2)
    assert np.allclose(body.get_disc_params(), (-0.05532064212457044, 0.11116537556358708, 1.0, 0.0))
    body.set_disc_params(7.5, 5.0, 4.5, 0.0)
    assert body.get_img_limits_xy() == ((-0.5, 14.5), (-0.5, 9.5))
    assert np.allclose(body.get_img_limits_radec(),
                       ((196.38091225891438, 196.36417481895663), (-5.571901975157448, -5.560796287842726)))  # fmt: skip
    assert np.allclose(body.get_img_limits_km(),
                       ((-151724.69753899056, 130727.50016257458), (-125236.31445765976, 117241.42226096484)))  # fmt: skip
    assert np.allclose(body.get_img_limits_angular(),
                       ((-31.984379466325663, 27.98633203326517), (-21.98926088314898, 17.99121344984992)))  # fmt: skip


def test_point_functions_kats(body):
    """
    The per-point siblings of the backplanes, value tables of the reference's tests/test_body.py:
    illumination :1826-1863, azimuth :1865-1898, local solar time :1900-1914, visibility
    :1732-1798, illuminated :1979-2006, radial velocity :2486-2519, distance :2521-2552,
    graphic <-> centric :2554-2595 (Body('Jupiter', observer='HST', utc='2005-01-01')).
    """
    close = lambda a, b, **kw: np.allclose(a, b, equal_nan=True, **kw)  # noqa: E731
    invalid = [(nan, nan), (nan, 0), (0, nan), (np.inf, np.inf)]

    def both_conventions(func, lonlat, expected, eq=close, **kw):
        assert eq(func(*lonlat, **kw), expected), (func.__name__, lonlat)
        for planetocentric in (False, True):
            ll = body.graphic2centric_lonlat(*lonlat, **{k: v for k, v in kw.items() if k == 'alt'}) if planetocentric else lonlat
            assert eq(func(*ll, planetocentric=planetocentric, **kw), expected), (func.__name__, lonlat, planetocentric)

    gie = [
        ((0, 0), (10.31594976458697, 163.2795134457034, 152.99822832991876)),
        ((123.456, -78.9), (10.316968817304499, 79.16351827229181, 77.68583738495468)),
    ] + [(ll, (nan, nan, nan)) for ll in invalid]
    for lonlat, angles in gie:
        both_conventions(body.illumination_angles_from_lonlat, lonlat, angles)
    az = [((0, 0), 177.66817822757469), ((123.456, -78.9), 169.57651996164563)] + [(ll, nan) for ll in invalid]
    for lonlat, angle in az:
        both_conventions(body.azimuth_angle_from_lonlat, lonlat, angle)
    rv = [((0, 0), -20.796924908179438), ((45, 45), -17.75706386255955)] + [(ll, nan) for ll in invalid]
    for lonlat, x in rv:
        both_conventions(body.radial_velocity_from_lonlat, lonlat, x)
    dist = [((0, 0), 819701772.0279644), ((45, 45), 819656453.7301536)] + [(ll, nan) for ll in invalid]
    for lonlat, x in dist:
        both_conventions(body.distance_from_lonlat, lonlat, x)
    assert isinstance(body.distance_from_lonlat(0, 0), float)
    ph, inc, em = body.illumination_angles_from_lonlat(np.array([0.0, 123.456]), np.array([0.0, -78.9]))  # arrays
    assert ph.shape == (2,) and close(em, (152.99822832991876, 77.68583738495468))

    lst = [
        (0, 22.89638888888889, '22:53:47'),
        (-90, 4.896388888888889, '04:53:47'),
        (123.456, 14.666111111111112, '14:39:58'),
        (999.999, 4.229722222222223, '04:13:47'),
        (nan, nan, ''),
        (np.inf, nan, ''),
    ]
    for lon, expected, s in lst:
        assert np.isclose(body.local_solar_time_from_lon(lon), expected, equal_nan=True), lon
        assert body.local_solar_time_string_from_lon(lon) == s

    same = lambda a, b: a == b  # noqa: E731
    for lonlat, visible in [((0, 0), False), ((180, 12), True), ((50, -80), True)] + [(ll, False) for ll in invalid]:
        both_conventions(body.test_if_lonlat_visible, lonlat, visible, eq=same)
    for (lon, lat, alt), visible in [
        ((0, 0, 0), False),
        ((0, 0, 1000000.0), True),
        ((153.1, -3.0, 0), True),
        ((153.1, -3.0, -1), False),
        ((153.1, -3.0, 1), True),
        ((153.1, nan, 1), False),
    ]:
        assert body.test_if_lonlat_visible(lon, lat, alt=alt) == visible, (lon, lat, alt)
        assert body.test_if_lonlat_visible(lon, lat, alt=alt, planetocentric=False) == visible
    for lonlat, lit in [((0, 0), False), ((180, 12), True), ((50, -80), False)] + [(ll, False) for ll in invalid]:
        both_conventions(body.test_if_lonlat_illuminated, lonlat, lit, eq=same)
    assert list(body.test_if_lonlat_illuminated(np.array([0.0, 180.0]), np.array([0.0, 12.0]))) == [False, True]

    pairs = [
        [(0, 0), (0, 0)],
        [(0, 90), (0, 90)],
        [(0, -90), (0, -90)],
        [(90, 0), (-90, 0)],
        [(123.4, 56.789), (-123.4, 53.17999536010973)],
        [
            (np.array([1.0, 2.0, 3.0, nan]), np.array([40.0, 50.0, 60.0, nan])),
            (np.array([-1.0, -2.0, -3.0, nan]), np.array([36.26969371, 46.18216311, 56.56575448, nan])),
        ],
    ]
    for graphic, centric in pairs:
        assert close(body.graphic2centric_lonlat(*graphic), centric), graphic
        assert close(body.centric2graphic_lonlat(*centric), graphic), centric
    for a in invalid:
        assert close(body.graphic2centric_lonlat(*a), (nan, nan))
        assert close(body.centric2graphic_lonlat(*a), (nan, nan))
    with pytest.raises(UnsupportedError):
        body.illumination_angles_from_lonlat(0, 0, alt=10.0)
    # device-resident results need an engine with a device (the CPU stand-in of these tests has none): said, not faked
    body.set_img_size(5, 4)
    with pytest.raises(UnsupportedError):
        body.get_lon_img(device=True)
    with pytest.raises(UnsupportedError):
        body.get_backplane_img('EMISSION', device=True)


def test_ring_and_limb_point_functions_kats(body):
    """tests/test_body.py:2008-2049 (ring plane) and :1683-1730 (limb, rtol 1e-5 there)"""
    close = lambda a, b, **kw: np.allclose(a, b, equal_nan=True, **kw)  # noqa: E731
    ring = [
        ((0, 0), (nan, nan, nan)),
        ((196.37198562427025, -5.565793847134351), (nan, nan, nan)),
        ((196.37347182693253, -5.561472466522512), (1377914.753652832, 152.91772706249577, 818261707.8278764)),
        ((196.3696997398314, -5.569843641306982), (nan, nan, nan)),
        ((196.3, -5.5), (9305877.091704229, 145.3644753085151, 810435703.2382222)),
        ((nan, 0), (nan, nan, nan)),
    ]
    for radec, expected in ring:
        assert close(body.ring_plane_coordinates(*radec), expected), radec
    assert close(
        body.ring_plane_coordinates(196.37198562427025, -5.565793847134351, only_visible=False),
        (4638.105239104683, 156.0690984698183, 819638074.3312378),
    )
    r, lo, d = body.ring_plane_coordinates(np.array([196.3, nan]), np.array([-5.5, 0.0]))
    assert r.shape == (2,) and close(r, (9305877.091704229, nan)) and close(lo, (145.3644753085151, nan))
    limb = [
        ((0, 0), (82.72145635455739, -7.331180721378409, 243226446.365406)),
        ((196.3719829300016, -5.565779946690757), (67.23274105785333, 58.34599234749429, -68089.8880967631)),
        ((196.372, -5.566), (248.13985326986065, -64.83923990338549, -64857.80811442864)),
        ((196.3, -5.5), (64.1290135632679, 20.79992677586983, 1320579.9259661217)),
    ]
    for radec, expected in limb:
        assert close(body.limb_coordinates_from_radec(*radec), expected, rtol=1e-5), radec
        lon_c, lat_c = body.graphic2centric_lonlat(*expected[:2])
        got = body.limb_coordinates_from_radec(*radec, planetocentric=True)
        assert close(got, (lon_c, lat_c, expected[2]), rtol=1e-5)
    assert close(body.limb_coordinates_from_radec(nan, 0), (nan, nan, nan))


def test_ring_and_grid_wireframe_coordinates_kats(body):
    """tests/test_body.py:2051-2081 (ring_radec) and :2107-2190 (visible_lonlat_grid_radec)"""
    close = lambda a, b, **kw: np.allclose(a, b, equal_nan=True, **kw)  # noqa: E731
    assert close(body.ring_radec(10000, npts=5), (np.full(5, nan), np.full(5, nan)))  # inside Jupiter
    assert close(
        body.ring_radec(100000, npts=5),
        ([nan, 196.36633034, 196.37500382, 196.37764017, nan], [nan, -5.56310623, -5.56681892, -5.56848105, nan]),
    )
    assert close(
        body.ring_radec(123456.789, npts=3, only_visible=False),
        ([196.36825958, 196.37571178, 196.36825958], [-5.56452821, -5.56705935, -5.56452821]),
    )
    assert close(body.ring_radec(nan, npts=2, only_visible=False), ([nan, nan], [nan, nan]))
    pole = (196.3700663, -5.57005326)
    expected = [
        ([pole[0], nan, nan, nan, nan], [pole[1], nan, nan, nan, nan]),
        ([pole[0], nan, nan, nan, nan], [pole[1], nan, nan, nan, nan]),
        ([pole[0], 196.36772166, 196.36794262, 196.37034361, nan], [pole[1], -5.56729981, -5.56387245, -5.56148116, nan]),
        ([pole[0], 196.36970087, 196.37065239, 196.37232288, nan], [pole[1], -5.56808941, -5.56495336, -5.56227057, nan]),
        ([pole[0], 196.37225066, 196.37414339, 196.37487263, nan], [pole[1], -5.56923855, -5.5665267, -5.56341971, nan]),
        ([pole[0], 196.37387716, 196.37637019, 196.37649901, nan], [pole[1], -5.57007398, -5.56767064, -5.56425534, nan]),
        ([pole[0], nan, nan, nan, nan], [pole[1], nan, nan, nan, nan]),
        ([pole[0], nan, nan, nan, nan], [pole[1], nan, nan, nan, nan]),
        ([pole[0]] * 5, [pole[1]] * 5),
        ([nan, 196.36772166, 196.37225066, nan, nan], [nan, -5.56729981, -5.56923855, nan, nan]),
        ([nan, 196.36794262, 196.37414339, nan, nan], [nan, -5.56387245, -5.5665267, nan, nan]),
        ([nan, 196.37034361, 196.37487263, nan, nan], [nan, -5.56148116, -5.56341971, nan, nan]),
    ]
    got = body.visible_lonlat_grid_radec(interval=45, npts=5)
    assert len(got) == len(expected)
    for g, e in zip(got, expected):
        assert close(g, e), (g, e)
    # the xy forms are radec2xy of the above (body_xy.py:1220-1249)
    body.set_img_size(15, 10)
    body.set_disc_params(5, 8, 3, 45)
    for (x, y), (ra, dec) in zip(body.visible_lonlat_grid_xy(interval=45, npts=5), got):
        assert close((x, y), body.radec2xy(ra, dec))
    assert close(body.ring_xy(123456.789, npts=3, only_visible=False), body.radec2xy(*body.ring_radec(123456.789, npts=3, only_visible=False)))


def test_result_planes_are_recycled_only_when_nobody_holds_them(jupiter):
    """
    `Engine.plane_buffer` / `recycle_plane` behind `BodyXY._img_planes` / `_clear_cache`: a cold getter after
    `set_disc_params` writes into the array the previous disc's plane lived in - unless the caller still holds that
    plane (the array itself, or any view of it), whose values must then stay what they were (the reference's arrays
    are independent of each other, body_xy.py:696-698 only drops the cache's reference).
    """
    from planetmapper_amd.engine import Engine

    class RecyclingOracleEngine(OracleEngine):
        """the oracle-backed test double with the real Engine's pool of result arrays (pageable here)"""

        plane_buffer, recycle_plane = Engine.plane_buffer, Engine.recycle_plane
        _calibrate_recycling, _forget_plane = Engine._calibrate_recycling, staticmethod(Engine._forget_plane)

        def __init__(self):
            super().__init__()
            self._plane_pool, self._plane_owned, self._plane_bytes, self._plane_limit = {}, {}, 0, 1 << 40
            self._free_counts = self._calibrate_recycling()
            self.lent = []

        def pinned_empty(self, shape, dtype=np.float64):  # (built like Engine.pinned_empty: a reshaped view of a buffer's array)
            return np.frombuffer(bytearray(int(np.prod(shape)) * 8), dtype=np.float64).reshape(shape)

        def backplanes_img(self, names, alt=0.0, *, recycled=False):
            out = super().backplanes_img(names, alt)
            if not recycled:
                return out
            res = {}
            for n, a in out.items():
                buf = self.plane_buffer(a.shape)
                buf[...] = a
                self.lent.append(id(buf))
                res[n] = buf
            return res

    eng = RecyclingOracleEngine()
    body = BodyXY('Jupiter', geometry=jupiter, nx=600, ny=500, engine=eng)  # (planes of 2.4 MB: above the pool's 1 MiB floor)
    body.set_disc_params(300, 250, 200, 10)
    lon1 = body.get_lon_img()
    assert not lon1.flags.writeable and body.get_lon_img() is lon1
    first = list(eng.lent)  # lon, lat
    del lon1
    body.set_disc_params(310, 250, 200, 10)  # nobody holds the planes: both arrays go back to the pool ...
    lon2 = body.get_lon_img()
    assert sorted(eng.lent[2:]) == sorted(first)  # ... and are written again
    keep = lon2[100:110]  # a VIEW of the plane, held by the user
    values = keep.copy()
    lat2 = body.get_lat_img()
    body.set_disc_params(320, 250, 200, 10)
    lon3 = body.get_lon_img()
    assert np.array_equal(keep, values, equal_nan=True) and np.array_equal(lat2, lat2.copy(), equal_nan=True)
    assert not np.array_equal(lon3[100:110], values, equal_nan=True)  # (another disc: other values, in another array)
    assert len(set(eng.lent[4:]) & set(eng.lent[2:4])) == 0  # neither held array was written again
    ref = BodyXY('Jupiter', geometry=jupiter, nx=600, ny=500, engine=OracleEngine())
    ref.set_disc_params(320, 250, 200, 10)
    assert np.array_equal(lon3, ref.get_lon_img(), equal_nan=True)
    body.set_img_size(64, 48)  # small planes: plain numpy arrays, nothing pooled
    body.set_disc_params(30, 20, 15, 0)
    assert body.get_emission_angle_img().shape == (48, 64)
