"""
GPU parity tests (run with `-m gpu` on an MI355X): the HIP engine, called through the
C ABI, against (1) the reference's golden vectors and (2) the CPU oracle on the same
seeded inputs.

Tolerances: tests/parity.py (masks bit-exact; angular planes 1e-9 deg x condition
number, >= 98 % of pixels inside the flat 1e-9 deg; km planes 2e-4 km on 8e8 km;
radial velocity 1e-9 km/s); mapped data within 1e-12.
"""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from parity import compare_planes, masks_agree

pytestmark = pytest.mark.gpu

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']


@pytest.fixture(scope='module')
def engine():
    from planetmapper_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module', params=['fast', 'general'])
def engine_fg(request, engine):
    """
    The module engine with the image planes routed through (a) the kernel the library selects
    (the spheroid fast path for every test geometry observed from afar) and (b) the general kernel
    `k_disc<FLAGS>` (`PM_OPT_GENERAL_KERNEL`): both must meet the same parity bars.
    """
    from planetmapper_amd import _lib

    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 1 if request.param == 'general' else 0)
    yield engine
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)


@pytest.fixture(scope='module')
def oracle():
    from oracle import oracle as o

    return o


def _compare(out, ref, names, g, r0=None, flat=True):
    """`flat=False` (the fuzz sweeps): no floor on the share of pixels inside the flat bar - the floors of
    tests/parity.py were measured on planets seen near their equator; a pole-on view (every pixel at high
    latitude, where the longitude's 1 / cos(lat) works) or a body at 30 au has another share (soak: 96-98 %)."""
    ps = None if r0 is None else g.diameter_arcsec / (2 * r0)  # BodyXY.get_plate_scale_arcsec
    return compare_planes(out, ref, names, g, plate_scale_arcsec=ps, min_flat_fraction=None if flat else 0.0)


import contextlib


@contextlib.contextmanager
def _library_choice(engine):
    """the module's engine with PM_OPT_GENERAL_KERNEL off, whatever `engine_fg` stretch of the session it is in"""
    from planetmapper_amd import _lib

    forced = engine.get_option(_lib.PM_OPT_GENERAL_KERNEL)
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
    try:
        yield engine
    finally:
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, forced)


def _check_golden(out, gold, names, g, what):
    """
    HIP against the reference's golden planes DIRECTLY (real CSPICE output): NaN masks identical, the reference's own
    comparison rule (tests/test_observation.py:1255: atol 1e-6, rtol 1e-5), and per pixel the tighter of the conditioned
    1e-9 deg evaluated on the GOLDEN planes and the flat bars of rounds 1-5 (tests/parity.py check_against_golden; the
    worst pixel / bar of every file is printed).
    """
    from parity import check_against_golden

    return check_against_golden(out, gold, names, g, f'HIP {what}')


def _golden_setup(engine, g):
    engine.set_geometry(g)
    engine.set_disc(2.5, 3.1, 3.9, float(np.deg2rad(123.456) % (2 * np.pi)), 7, 10, True)


def test_golden_nav_all_planes(engine, oracle, jupiter):
    """tests/data/outputs/test_nav.fits at the reference's own tolerance, and the oracle."""
    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav.npz'))
    _golden_setup(engine, jupiter)
    out = engine.backplanes_img(oracle.PLANE_NAMES)
    _check_golden(out, gold, oracle.PLANE_NAMES, jupiter, 'test_nav')
    ref = oracle.backplanes_img(jupiter, oracle.make_disc(2.5, 3.1, 3.9, 123.456, 7, 10), oracle.PLANE_NAMES)
    _compare(out, ref, oracle.PLANE_NAMES, jupiter)


def test_golden_nav_alt(engine, oracle, jupiter):
    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav_alt.npz'))
    _golden_setup(engine, jupiter)
    out = engine.backplanes_img(oracle.PLANE_NAMES, alt=34567.8912)
    _check_golden(out, gold, oracle.PLANE_NAMES, jupiter, 'test_nav_alt')


@pytest.mark.parametrize(
    'name,interp,alt',
    [
        ('map_rectangular_linear', 'linear', 0.0),
        ('map_rectangular_nearest', 'nearest', 0.0),
        ('map_rectangular_nearest_alt', 'nearest', 34567.8912),
    ],
)
def test_golden_maps(engine, oracle, jupiter, name, interp, alt):
    gold = np.load(os.path.join(GOLDEN, f'golden_{name}.npz'))
    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    _golden_setup(engine, jupiter)
    lon, lat = gold['LON-GRAPHIC'], gold['LAT-GRAPHIC']
    out = engine.backplanes_map(oracle.PLANE_NAMES, lon, lat, alt=alt)
    _check_golden(out, gold, oracle.PLANE_NAMES, jupiter, name)
    disc = oracle.make_disc(2.5, 3.1, 3.9, 123.456, 7, 10)
    ref = oracle.backplanes_map(jupiter, disc, oracle.PLANE_NAMES, lon, lat, alt=alt)
    _compare(out, ref, oracle.PLANE_NAMES, jupiter, r0=3.9)
    mapped = engine.map_cube(cube, out['PIXEL-X'], out['PIXEL-Y'], interp, True)
    from parity import check_mapped_against_golden

    check_mapped_against_golden(mapped, gold['PRIMARY'], interp, f'HIP {name}')
    ref_mapped = oracle.map_cube(cube, out['PIXEL-X'], out['PIXEL-Y'], interp, True)
    assert np.array_equal(np.isnan(mapped), np.isnan(ref_mapped))
    assert np.nanmax(np.abs(mapped - ref_mapped), initial=0.0) <= 1e-12


@pytest.mark.parametrize('body', ['jupiter', 'saturn', 'triaxial_east'])
def test_map_planes_fine_grid_both_kernels_and_any_subset(engine, oracle, jupiter, saturn, body):
    """
    The map-space chain at throughput size (a 0.5 deg grid, 259 200 cells, with cells at the poles, on the seam and without
    coordinates): all 26 planes from the B0 kernel the library takes (`k_map_b0`) AND from the J2000 kernel behind
    PM_OPT_GENERAL_KERNEL (`k_map`) against the oracle; then what the B0 kernel's scalar branches must guarantee - a plane
    does not depend on which other planes were asked for with it (each group of the chain alone against the full request),
    and the device-resident form (`pm_backplanes_map`, PM_MEM_DEVICE: grids and planes in HBM) gives the same bits.
    """
    import torch

    from planetmapper_amd import _lib

    g = {'jupiter': jupiter, 'saturn': saturn}.get(body) or _variant(jupiter, radii=[71492.0, 70100.0, 68800.0], west_positive=0)
    nx, ny, x0, y0, r0, rot = 400, 300, 205.3, 148.1, 120.0, 0.7
    engine.set_geometry(g)
    engine.set_disc(x0, y0, r0, rot, nx, ny, True)
    d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
    d.rotation_rad = rot
    lons = np.arange(0.25, 360, 0.5)
    lats = np.arange(-89.75, 90, 0.5)
    lon, lat = np.meshgrid(lons[::-1] if g.west_positive else lons, lats)
    lon, lat = np.ascontiguousarray(lon), np.ascontiguousarray(lat)
    lat[0, :8] = [-90.0, 90.0, -90.0, 90.0, 89.999999, -89.999999, 0.0, 0.0]
    lon[0, :8] = [0.0, 0.0, 360.0, 180.0, 359.999999, 1e-9, 0.0, 360.0]
    lon[5, 5], lat[6, 6], lon[7, 7] = np.nan, np.inf, -np.inf
    names = list(oracle.PLANE_NAMES)
    ref = oracle.backplanes_map(g, d, names, lon, lat)
    full = engine.backplanes_map(names, lon, lat)
    _compare(full, ref, names, g, r0=r0)
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 1)
    try:
        _compare(engine.backplanes_map(names, lon, lat), ref, names, g, r0=r0)
    finally:
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
    groups = [['LON-GRAPHIC'], ['LOCAL-SOLAR-TIME', 'LAT-GRAPHIC'], ['LON-CENTRIC'], ['LAT-CENTRIC', 'EMISSION'], ['EMISSION'], ['DISTANCE'],
              ['PHASE'], ['INCIDENCE', 'AZIMUTH'], ['RADIAL-VELOCITY'], ['DOPPLER', 'PHASE'], ['RA'], ['DEC', 'PIXEL-Y'], ['PIXEL-X'],
              ['KM-X', 'ANGULAR-Y'], ['ANGULAR-X'], ['LIMB-LAT-GRAPHIC'], ['LIMB-DISTANCE', 'LIMB-LON-GRAPHIC'], ['RING-RADIUS'],
              ['RING-LON-GRAPHIC', 'RING-DISTANCE', 'KM-Y']]  # fmt: skip
    for grp in groups:
        part = engine.backplanes_map(grp, lon, lat)
        for n in grp:
            assert np.array_equal(np.isnan(part[n]), np.isnan(full[n])), (grp, n)
            # (another instantiation of the kernel template: the same operations, not necessarily the same contractions)
            scale = float(np.nanmax(np.abs(full[n]), initial=1.0))
            assert np.nanmax(np.abs(part[n] - full[n]), initial=0.0) <= 4e-16 * scale + 2e-13, (grp, n)
    dev = {n: torch.empty(lon.shape, dtype=torch.float64, device='cuda') for n in names}
    engine.backplanes_map_device(dev, torch.from_numpy(lon).cuda(), torch.from_numpy(lat).cuda(), *lon.shape)
    engine.synchronize()
    for n in names:
        assert np.array_equal(dev[n].cpu().numpy(), full[n], equal_nan=True), n


@pytest.mark.parametrize('sz,rot', [(128, 0.0), (1024, 0.0), (517, 33.3)])
def test_jupiter_full_set_vs_oracle(engine_fg, oracle, jupiter, sz, rot):
    """BASELINE configs 1-2: centred disc (BodyXY.centre_disc body_xy.py:791), all 26 planes."""
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    engine_fg.set_geometry(jupiter)
    engine_fg.set_disc(x0, y0, r0, float(np.deg2rad(rot) % (2 * np.pi)), sz, sz, True)
    out = engine_fg.backplanes_img(oracle.PLANE_NAMES)
    from planetmapper_amd import _lib

    general = engine_fg.get_option(_lib.PM_OPT_GENERAL_KERNEL)
    assert engine_fg.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == (3 if general else 1)
    ref = oracle.backplanes_img(jupiter, oracle.make_disc(x0, y0, r0, rot, sz, sz), oracle.PLANE_NAMES)
    _compare(out, ref, oracle.PLANE_NAMES, jupiter)
    frac = np.isfinite(out['LON-GRAPHIC']).mean()
    assert 0.55 < frac < 0.65


def test_no_optimize_speed_and_ragged_sizes(engine_fg, oracle, jupiter):
    """optimize_speed=False (no radius pre-mask) and a width that is not a multiple of 64."""
    engine_fg.set_geometry(jupiter)
    engine_fg.set_disc(40.2, 17.7, 30.0, 1.0, 131, 67, False)
    out = engine_fg.backplanes_img(oracle.PLANE_NAMES, alt=1234.5)
    d = oracle.make_disc(40.2, 17.7, 30.0, 0.0, 131, 67, optimize_speed=False)
    d.rotation_rad = 1.0
    ref = oracle.backplanes_img(jupiter, d, oracle.PLANE_NAMES, alt=1234.5)
    _compare(out, ref, oracle.PLANE_NAMES, jupiter)


def test_a_frame_taller_than_one_launch(engine_fg, oracle, jupiter):
    """
    70 000 x 64: more rows than gridDim.y holds (65 535) - the reference's loop (body_xy.py:3155-3166) has no limit; the
    library maps such a frame in row blocks. A strip through a disc of 30 000 px radius: limb at both ends, all 26 planes,
    whole frame and a row window that straddles the block boundary; into host memory (sparse bands) and device memory.
    """
    import torch

    nx, ny = 64, 70_000
    x0, y0, r0, rot = 20.3, 34_000.7, 30_000.0, 0.4
    engine_fg.set_geometry(jupiter)
    engine_fg.set_disc(x0, y0, r0, rot, nx, ny, True)
    d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
    d.rotation_rad = rot
    ref = oracle.backplanes_img(jupiter, d, oracle.PLANE_NAMES)
    out = engine_fg.backplanes_img(oracle.PLANE_NAMES)
    assert out['LON-GRAPHIC'].shape == (ny, nx) and 0.8 < np.isfinite(ref['LON-GRAPHIC']).mean() < 0.9
    # (at 30 000 px per radius the strip resolves the sub-observer point to 1e-5 rad of emission: within a few pixels of it
    #  the azimuth - pi - acos(q / (sin e sin i)) - is conditioned beyond the capped bar of tests/parity.py; those pixels,
    #  emission < 0.01 deg, are held to 0.05 deg, the rest of the plane to the bar like every other plane)
    core = ref['EMISSION'] < 0.01
    assert 0 < core.sum() < 200
    az_out, az_ref = out['AZIMUTH'].copy(), ref['AZIMUTH'].copy()
    assert np.array_equal(np.isnan(az_out), np.isnan(az_ref)) and np.nanmax(np.abs(az_out[core] - az_ref[core])) < 0.05
    az_out[core] = az_ref[core]
    _compare({**out, 'AZIMUTH': az_out}, ref, oracle.PLANE_NAMES, jupiter, r0=r0)
    names = ['LON-GRAPHIC', 'EMISSION', 'RA', 'RING-RADIUS']
    bufs = {n: torch.full((ny, nx), 7.0, dtype=torch.float64, device='cuda') for n in names}
    engine_fg.backplanes_img_device(bufs)
    engine_fg.synchronize()
    for n in names:
        assert np.array_equal(bufs[n].cpu().numpy(), out[n], equal_nan=True), n
    lo, n_rows = 30_000, 36_000  # (crosses row 32 768 of the frame and is itself more than one block)
    win = engine_fg.backplanes_img_rows(names, lo, n_rows)
    for n in names:
        assert np.array_equal(win[n], out[n][lo : lo + n_rows], equal_nan=True), n


def test_saturn_rings_divergent_path(engine_fg, oracle, saturn):
    """BASELINE config 4 geometry at 768^2: ring planes fill the frame (parity unpinned by
    the reference; GPU vs oracle only)."""
    sz = 768
    x0 = y0 = (sz - 1) / 2
    engine_fg.set_geometry(saturn)
    engine_fg.set_disc(x0, y0, 150.0, float(np.deg2rad(20.0)), sz, sz, True)
    names = HEADLINE + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE', 'DISTANCE']
    out = engine_fg.backplanes_img(names)
    ref = oracle.backplanes_img(saturn, oracle.make_disc(x0, y0, 150.0, 20.0, sz, sz), names)
    for n in names:
        assert np.array_equal(np.isnan(out[n]), np.isnan(ref[n])), n
    _compare(out, ref, names, saturn)
    assert np.isfinite(out['RING-RADIUS']).mean() > 0.3


def test_map_and_reprojection_vs_oracle(engine, oracle, jupiter):
    """BASELINE config 3 shape at reduced size: 1 deg rectangular map of a synthetic cube."""
    sz = 512
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, r0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 1.0)
    assert lon.shape == (180, 360) and lon[0, 0] == 359.5 and lat[0, 0] == -89.5
    disc = oracle.make_disc(x0, y0, r0, 0.0, sz, sz)
    out = engine.backplanes_map(oracle.PLANE_NAMES, lon, lat)
    ref = oracle.backplanes_map(jupiter, disc, oracle.PLANE_NAMES, lon, lat)
    _compare(out, ref, oracle.PLANE_NAMES, jupiter, r0=r0)
    rng = np.random.default_rng(20050101)
    yy, xx = np.mgrid[:sz, :sz]
    mu = np.sqrt(np.clip(1 - ((xx - x0) ** 2 + (yy - y0) ** 2) / r0**2, 0, None))
    cube = mu[None] + 0.05 * rng.standard_normal((4, sz, sz))
    cube[rng.random(cube.shape) < 1e-3] = np.nan
    cube[3, 100] = np.nan
    for interp in ('linear', 'nearest'):
        a = engine.map_cube(cube, ref['PIXEL-X'], ref['PIXEL-Y'], interp, True)
        b = oracle.map_cube(cube, ref['PIXEL-X'], ref['PIXEL-Y'], interp, True)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) <= 1e-12
    for dt in (np.float32, np.int16, np.uint8):
        c = (np.nan_to_num(cube) * 100).astype(dt)
        a = engine.map_cube(c, ref['PIXEL-X'], ref['PIXEL-Y'], 'linear', True)
        b = oracle.map_cube(c, ref['PIXEL-X'], ref['PIXEL-Y'], 'linear', True)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) <= 1e-9


def test_error_behaviour(engine, jupiter):
    """ValueError cases of the reference: empty image (body_xy.py:3167), unknown
    interpolation (:1630), image shape mismatch (:1587)."""
    engine.set_geometry(jupiter)
    engine.set_disc(0, 0, 1, 0, 0, 0, True)
    with pytest.raises(ValueError):
        engine.backplanes_img(['LON-GRAPHIC'])
    engine.set_disc(3, 3, 2, 0, 8, 8, True)
    with pytest.raises(ValueError):
        engine.map_cube(np.zeros((8, 8)), np.zeros((2, 2)), np.zeros((2, 2)), 'bicubic')
    with pytest.raises(ValueError):
        engine.map_cube(np.zeros((7, 8)), np.zeros((2, 2)), np.zeros((2, 2)), 'linear')
    with pytest.raises(ValueError):
        engine.set_disc(0, 0, -1, 0, 8, 8, True)
    # caller-supplied maps that leave the frame (or have a NaN y for a finite x) never read
    # outside the plane: such cells come back NaN for every interpolation
    img = np.arange(64.0).reshape(8, 8)
    xm = np.array([[3.2, 1e9, -1e9, 2.0, np.inf, 7.49], [-0.4, 3.0, 8.6, 1e300, -7.6, np.nan]])
    ym = np.array([[2.1, 3.0, 3.0, np.nan, 2.0, 7.49], [-0.4, -1e12, 2.0, 2.0, 1.0, 1.0]])
    for interp in ('nearest', 'linear', 'cubic', 'smooth'):
        for prop in (True, False):
            out = engine.map_cube(img, xm, ym, interp, prop)[0]
            assert np.isfinite(out[0, 0]) and np.isnan(out[1, 5]), (interp, prop)
            if interp == 'nearest':
                assert out[0, 0] == img[2, 3] and out[0, 5] == img[7, 7] and out[1, 0] == img[0, 0]
                assert np.isnan(out[0, 1:5]).all() and np.isnan(out[1, 1:4]).all()


def test_headline_frame_4096_vs_oracle_and_round_trip(engine, oracle, jupiter):
    """
    BASELINE headline config at full size: 4096^2, 5 planes. Mask bit-exact and
    conditioned tolerances against the (OpenMP) oracle, plus a size-independent
    property tying the image kernel to the map kernel: feeding the LON/LAT planes of
    the image back through the map direction (lonlat -> x, y) must return the pixel
    coordinates they came from.
    """
    sz = 4096
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    from planetmapper_amd import _lib

    oracle.set_num_threads(16)
    ref = oracle.backplanes_img(jupiter, oracle.make_disc(x0, y0, r0, 0.0, sz, sz), HEADLINE)
    # The general kernel first, the library's own choice last (`out` below is the headline kernel's). Share of on-disc pixels
    # inside the flat 1e-9 deg, measured (round 6, observer from the TLE ephemeris; the header-derived observer of rounds 1-5
    # gives the same to 0.005 %): library's choice LON 99.45 %, LAT 99.997 %, INC / EMI 99.79 %, PHASE 100 %; general kernel
    # LON 99.04 %, LAT 99.997 %, INC / EMI 99.50 %. (Until round 6 this test ran under whichever kernel the module's
    # parametrised fixture had left selected - the general one - against the fast kernel's floors: 99.4990 % failed 99.5.)
    floors = {1: {'LON-GRAPHIC': 0.985, 'LAT-GRAPHIC': 0.999, 'PHASE': 1.0, 'INCIDENCE': 0.99, 'EMISSION': 0.99},
              0: {'LON-GRAPHIC': 0.99, 'LAT-GRAPHIC': 0.999, 'PHASE': 1.0, 'INCIDENCE': 0.995, 'EMISSION': 0.995}}
    forced = engine.get_option(_lib.PM_OPT_GENERAL_KERNEL)
    try:
        for general in (1, 0):
            engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, general)
            engine.set_geometry(jupiter)
            engine.set_disc(x0, y0, r0, 0.0, sz, sz, True)
            out = engine.backplanes_img(HEADLINE)
            stats = _compare(out, ref, HEADLINE, jupiter)
            which = 'general kernel' if general else 'the kernel the library chooses'
            print(f'\n4096^2 HIP ({which}) vs oracle (max |diff| deg, fraction within flat 1e-9 deg):', stats)
            assert int(np.isfinite(out['LON-GRAPHIC']).sum()) == int(np.isfinite(ref['LON-GRAPHIC']).sum())
            for n in HEADLINE:
                assert stats[n][1] >= floors[general][n], (general, n, stats[n])
    finally:
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, forced)
    # round trip on a band of rows through the disc (emission < 85 deg: well conditioned)
    rows = slice(2040, 2056)
    lon, lat, emi = out['LON-GRAPHIC'][rows], out['LAT-GRAPHIC'][rows], out['EMISSION'][rows]
    xm, ym = engine.xy_map(np.ascontiguousarray(lon), np.ascontiguousarray(lat))
    yy, xx = np.mgrid[rows, 0:sz]
    ok = np.isfinite(lon) & (emi < 85.0)
    assert ok.sum() > 40000
    assert np.isfinite(xm[ok]).all()
    # not exact by design: Body._targvec2obsvec ignores the target's translation during the
    # light-time offset (body.py:917-948), worth ~0.04 km = 1e-3 px at this plate scale
    assert np.max(np.abs(xm[ok] - xx[ok])) < 5e-3 and np.max(np.abs(ym[ok] - yy[ok])) < 5e-3


def test_reference_api_on_gpu(jupiter):
    """The drop-in surface (BodyXY / Observation method names) on the real engine vs goldens."""
    from planetmapper_amd import BodyXY, Observation

    gold = np.load(os.path.join(GOLDEN, 'golden_test_nav.npz'))
    body = BodyXY('Jupiter', '2005-01-01T00:00:00', observer='HST', scenario='jupiter_hst_2005', nx=7, ny=10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    lon = body.get_lon_img()
    assert lon.shape == (10, 7) and lon.dtype == np.float64 and not lon.flags.writeable
    for name in body.backplanes:
        img = body.get_backplane_img(name)
        assert np.allclose(img, gold[name], rtol=1e-5, atol=1e-6, equal_nan=True), name
    assert np.array_equal(body.get_emission_angle_img(), body.backplanes['EMISSION'].get_img(), equal_nan=True)
    gmap = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_linear.npz'))
    for name in body.backplanes:
        m = body.get_backplane_map(name, degree_interval=30)
        assert np.allclose(m, gmap[name], rtol=1e-5, atol=1e-6, equal_nan=True), name
    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    obs = Observation(data=cube, geometry=jupiter)
    obs.set_disc_params(2.5, 3.1, 3.9, 123.456)
    assert np.allclose(obs.get_mapped_data(degree_interval=30), gmap['PRIMARY'], rtol=1e-5, atol=1e-6, equal_nan=True)
    gn = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_nearest.npz'))['PRIMARY']
    assert np.array_equal(obs.get_mapped_data('nearest', degree_interval=30), gn, equal_nan=True)
    # BASELINE config 1: 128 x 128 get_lon_img + get_emission_angle_img through the API
    b = BodyXY('jupiter', scenario='jupiter_hst_2005', sz=128)
    assert b.get_disc_params() == (63.5, 63.5, 57.15, 0.0)
    assert np.isfinite(b.get_lon_img()).sum() == np.isfinite(b.get_emission_angle_img()).sum() > 9000


def test_device_resident_getters_of_the_drop_in_surface(jupiter):
    """
    `get_*_img(device=True)`, `get_backplane_img(name, device=True)`, `get_mapped_data(..., device=True)`: the results
    where they were computed - `DeviceArray`s in HBM read through DLPack (torch.from_dlpack) and
    `__cuda_array_interface__` - equal to the numpy getters bit for bit, cached like them (base.py:115-138,
    body_xy.py:2586-2630), invalidated by `_clear_cache` without pulling memory from under a consumer that imported it.
    """
    import gc

    import torch

    from planetmapper_amd import Observation
    from planetmapper_amd._lib import UnsupportedError
    from planetmapper_amd.device_array import DeviceArray

    rng = np.random.default_rng(11)
    data = rng.standard_normal((3, 96, 120)).astype(np.float32)
    data[1, 40:44, 50:53] = np.nan
    obs = Observation(data=data, geometry=jupiter)
    obs.set_disc_params(60.3, 47.1, 40.0, 33.0)
    same = lambda d, h: np.array_equal(d.numpy(), h, equal_nan=True)  # noqa: E731
    lon_d = obs.get_lon_img(device=True)
    assert isinstance(lon_d, DeviceArray) and lon_d.shape == (96, 120) and lon_d.dtype == np.float64 and lon_d.valid
    assert obs.get_lon_img(device=True) is lon_d  # the cache entry itself, like the read-only numpy view
    t = torch.from_dlpack(lon_d)
    assert t.is_cuda and t.dtype == torch.float64 and tuple(t.shape) == (96, 120) and t.data_ptr() == lon_d.ptr
    host = obs.get_lon_img()
    assert np.array_equal(t.cpu().numpy(), host, equal_nan=True) and same(lon_d, host) and np.isfinite(host).sum() > 3000
    cai = lon_d.__cuda_array_interface__
    assert cai['data'] == (lon_d.ptr, True) and cai['shape'] == (96, 120) and cai['typestr'] == '<f8' and cai['version'] == 3
    assert lon_d.__dlpack_device__() == (10, 0)  # kDLROCM
    for name in ('EMISSION', 'RA', 'RING-RADIUS', 'LIMB-LAT-GRAPHIC', 'LOCAL-SOLAR-TIME', 'DOPPLER'):
        assert same(obs.get_backplane_img(name, device=True), obs.get_backplane_img(name)), name
    assert same(obs.get_backplane_img('LON-GRAPHIC', alt=2500.0, device=True), obs.get_backplane_img('LON-GRAPHIC', alt=2500.0))
    assert not same(obs.get_backplane_img('LON-GRAPHIC', alt=2500.0, device=True), host)
    for interp, kw in (('linear', {}), ('nearest', {}), ('cubic', {}), ('smooth', {}), ('linear', {'propagate_nan': False}),
                       ('cubic', {'spline_smoothing': 96 * 120.0})):
        m_d = obs.get_mapped_data(interp, degree_interval=10, device=True, **kw)
        m_h = obs.get_mapped_data(interp, degree_interval=10, **kw)
        assert isinstance(m_d, DeviceArray) and m_d.shape == (3, 18, 36) and same(m_d, m_h), (interp, kw)
        assert np.isfinite(m_h).sum() > 100
    assert obs.get_mapped_data('linear', degree_interval=10, device=True) is obs.get_mapped_data('linear', degree_interval=10, device=True)
    # map space: the planes of a map grid in HBM (the grid itself uploaded once per set of map keywords)
    for name in ('PIXEL-X', 'EMISSION', 'LON-CENTRIC', 'RADIAL-VELOCITY', 'LIMB-DISTANCE', 'RING-RADIUS', 'LOCAL-SOLAR-TIME'):
        for kw in ({'degree_interval': 10}, {'degree_interval': 15, 'alt': 1200.0}, {'projection': 'orthographic', 'size': 24}):
            assert same(obs.get_backplane_map(name, device=True, **kw), obs.get_backplane_map(name, **kw)), (name, kw)
    assert same(obs.get_emission_angle_map(device=True, degree_interval=10), obs.get_emission_angle_map(degree_interval=10))
    assert obs.get_x_map(device=True, degree_interval=10) is obs.get_backplane_map('PIXEL-X', device=True, degree_interval=10)
    emi_map_d, x_map_d = obs.get_emission_angle_map(device=True, degree_interval=10), obs.get_x_map(device=True, degree_interval=10)
    obs.register_backplane('MINE', 'a user function', lambda: np.zeros((96, 120)), lambda **kw: np.zeros((18, 36)))
    with pytest.raises(UnsupportedError):
        obs.get_backplane_img('MINE', device=True)
    # ---- a new disc: the handles of the old one are dead, a consumer's import is not
    before = t.clone()
    mapped_old = obs.get_mapped_data('linear', degree_interval=10, device=True)
    obs.set_x0(61.3)
    assert not lon_d.valid and not mapped_old.valid
    assert not x_map_d.valid and emi_map_d.valid  # x / y maps go with the disc; the other map planes do not depend on it
    assert obs.get_emission_angle_map(device=True, degree_interval=10) is emi_map_d
    assert same(obs.get_x_map(device=True, degree_interval=10), obs.get_x_map(degree_interval=10))
    for dead in (lambda: lon_d.ptr, lambda: lon_d.__cuda_array_interface__, lambda: torch.from_dlpack(lon_d), lon_d.numpy):
        with pytest.raises(ValueError, match='cache entry that has been cleared'):
            dead()
    lon_new = obs.get_lon_img(device=True)
    assert lon_new is not lon_d and lon_new.ptr != t.data_ptr()  # (the imported memory was NOT handed out again)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(t), torch.nan_to_num(before))  # ... nor overwritten
    assert same(lon_new, obs.get_lon_img()) and not np.array_equal(lon_new.numpy(), host, equal_nan=True)
    pooled = obs._engine._device_pool_bytes
    assert lon_d._exports == 1 and lon_d in obs._engine._device_waiting
    del t
    gc.collect()
    assert lon_d._exports == 0
    obs._engine._device_sweep()
    assert obs._engine._device_pool_bytes == pooled + 96 * 120 * 8  # the consumer let go: the plane is in the engine's pool
    nxt = obs.get_lat_img(device=True)  # (LAT of the new disc was computed with LON: cached)
    obs.set_x0(60.3)
    assert not nxt.valid and obs._engine._device_pool_bytes > pooled + 96 * 120 * 8
    again = obs.get_lon_img(device=True)
    assert same(again, host)  # the first disc again, out of pooled memory
    # a capsule nobody consumes releases its hold when it is dropped
    cap = again.__dlpack__()
    assert again._exports == 1
    del cap
    gc.collect()
    assert again._exports == 0
    # an engine that is closed takes the memory of its live arrays with it (their handles go invalid); an import a consumer
    # still holds keeps ITS block until the consumer lets go
    from planetmapper_amd.engine import Engine

    e2 = Engine(0)
    kept, imported = e2.device_array((4, 5)), e2.device_array((3,))
    t2 = torch.from_dlpack(imported)
    e2.close()
    assert not kept.valid and not imported.valid and kept._ptr == 0 and imported._ptr == 0
    t2.fill_(1.0)
    torch.cuda.synchronize()
    assert float(t2.sum()) == 3.0
    del t2, kept, imported
    gc.collect()
    # a consumer that still holds its import when the interpreter goes down: the import's deleter is C code of the library
    # (a Python callback there is a call into an interpreter that no longer exists)
    import subprocess
    import sys

    code = ('import numpy as np, torch\n'
            'from planetmapper_amd import BodyXY\n'
            'b = BodyXY("jupiter", scenario="jupiter_hst_2005", sz=64)\n'
            'keep = [torch.from_dlpack(b.get_lon_img(device=True)), torch.from_dlpack(b.get_emission_angle_img(device=True))]\n'
            'b.set_x0(30.0)\n'  # one of the imports outlives its cache entry as well
            'keep.append(torch.from_dlpack(b.get_lon_img(device=True)))\n'
            'print("alive", len(keep), flush=True)\n')
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and 'alive 3' in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-1500:])


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
def test_projected_golden_maps_through_the_hip_map_kernel(jupiter, oracle, name, kw):
    """
    The reference's six orthographic / azimuthal golden maps (tests/test_observation.py:1123-1153;
    tests/data/outputs/map_orthographic-*.fits, map_azimuthal-*.fits): the projected lon/lat grid
    (closed-form, planetmapper_amd/projections.py) goes through the HIP `k_map` kernel for all 26
    map-space planes and through `k_reproject` for the 10-plane cube, at the reference's tolerance,
    with bit-exact NaN masks, and against the oracle on the same grid.
    """
    from planetmapper_amd import BodyXY

    gold = np.load(os.path.join(GOLDEN, f'golden_{name}.npz'))
    body = BodyXY('Jupiter', '2005-01-01T00:00:00', observer='HST', geometry=jupiter, nx=7, ny=10)
    body.set_disc_params(2.5, 3.1, 3.9, 123.456)
    lons, lats, *_ = body.generate_map_coordinates(**kw)
    assert np.array_equal(np.isnan(lons), np.isnan(gold['LON-GRAPHIC']))
    d = oracle.make_disc(2.5, 3.1, 3.9, 123.456, 7, 10)
    ref = oracle.backplanes_map(jupiter, d, oracle.PLANE_NAMES, lons, lats)
    got = {}
    for n in body.backplanes:
        if n in gold.files:
            got[n] = body.get_backplane_map(n, **kw)
    _check_golden(got, gold, list(got), jupiter, name)
    _compare(got, ref, list(got), jupiter, r0=3.9)
    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    mapped = body.map_img(cube, **kw)
    from parity import check_mapped_against_golden

    check_mapped_against_golden(mapped, gold['PRIMARY'], 'linear', f'HIP {name}')


def test_nan_preclean_and_median_paths(engine, oracle, jupiter):
    """
    propagate_nan=False and +-inf pixels: the NaN pre-clean of
    BodyXY._replace_nans_with_interpolated_values (body_xy.py:1871-1904) incl. the plane
    nanmedian (GPU radix select), vs the oracle; plus the reference KAT
    tests/test_body_xy.py:1177-1190 through the real engine.
    """
    nan = np.nan
    image = np.array(
        [
            [0.0, 100.0, -1.0, 2.2, 3.3, 4.4],
            [0.0, 75.0, 999.0, 50.0, 1.0, 123.456789],
            [0.0, 25.0, 0.0, 123.45, nan, 3],
            [0.0, 0.123, 0.0, 3.0, 0.1, nan],
            [100.0, -100.0, 100.0, -100.0, 100.0, nan],
        ]
    )
    # fmt: off
    no_propagation = [[nan, nan, 83.42502054006614, 61.410255547165704, 1.0972142916279704, nan, nan, nan], [nan, nan, nan, 61.591824124152424, 488.0893412811879, 4.181692402514696, 3.8032713799190443, nan], [nan, nan, nan, 3.678385742930187, 94.03788871233297, 35.721226497463014, 94.00305287602345, nan], [nan, nan, nan, -25.28910210942658, -1.6502703714050462, 4.265385156596395, nan, nan]]
    # fmt: on
    from planetmapper_amd import BodyXY

    body = BodyXY('Jupiter', geometry=jupiter, nx=6, ny=5)
    body.set_disc_params(2.75, 1.3, 2.3, 45.678)
    got = body.map_img(image, degree_interval=45, propagate_nan=False)
    assert np.allclose(got, no_propagation, rtol=1e-5, atol=1e-8, equal_nan=True)

    sz = 256
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.9 * x0, 0.3, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 2.0)
    d = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
    d.rotation_rad = 0.3
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    rng = np.random.default_rng(7)
    cube = rng.standard_normal((7, sz, sz)) * 10 + 3
    cube[0][rng.random((sz, sz)) < 0.02] = np.nan  # isolated NaNs: window means only
    cube[1][60:140, 80:200] = np.nan  # a hole: its interior needs the plane median
    cube[2][:] = np.nan  # all NaN
    cube[3][:] = np.inf  # nothing finite, but not all NaN: median 0.0
    cube[4][rng.random((sz, sz)) < 0.01] = np.inf  # isolated infs
    cube[4][100:120, 100:130] = -np.inf
    cube[5][::2] = np.nan  # even count / striped
    cube[6][rng.random((sz, sz)) < 0.5] = np.nan
    for prop in (False, True):
        a = engine.map_cube(cube, xm, ym, 'linear', prop)
        b = oracle.map_cube(cube, xm, ym, 'linear', prop)
        assert np.array_equal(np.isnan(a), np.isnan(b)), prop
        assert np.array_equal(np.isinf(a), np.isinf(b)), prop
        fin = np.isfinite(b)
        assert np.max(np.abs(a[fin] - b[fin])) <= 1e-11, prop
    assert np.isnan(engine.map_cube(cube[2], xm, ym, 'linear', False)).all()
    # float32 and int16 cubes through the median path
    c32 = cube[1].astype(np.float32)
    a = engine.map_cube(c32, xm, ym, 'linear', False)
    b = oracle.map_cube(c32, xm, ym, 'linear', False)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) <= 1e-5


def test_async_map_cube_is_finished_before_its_flags_are_regrown(engine, oracle, jupiter):
    """
    A device-mode pm_map_cube stays pending until pm_synchronize() (planes that need the nanmedian
    are replayed there). A second call with MORE planes regrows the per-plane flag array the pending
    call points at: the library must finish the first call before it frees that array - the first
    output is then final, with the median applied (an -inf block: interior pixels need it).
    """
    import torch

    from planetmapper_amd.engine import Engine

    eng = Engine(0)  # fresh context: its flag array starts empty and grows with the calls below
    try:
        sz = 128
        x0 = y0 = (sz - 1) / 2
        eng.set_geometry(jupiter)
        eng.set_disc(x0, y0, 0.9 * x0, 0.0, sz, sz, True)
        lon, lat = oracle.rectangular_grid(jupiter, 4.0)
        d = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
        xm, ym = oracle.xy_map(jupiter, d, lon, lat)
        n0, n1 = xm.shape
        rng = np.random.default_rng(11)
        small = rng.standard_normal((2, sz, sz)) + 5.0
        small[1][40:90, 30:100] = -np.inf
        big = rng.standard_normal((9, sz, sz))
        big[7][20:60, 50:110] = np.inf
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        dxm, dym, dsmall, dbig = t(xm), t(ym), t(small), t(big)
        o_small = torch.full((2, n0, n1), -7.0, dtype=torch.float64, device='cuda')
        o_big = torch.full((9, n0, n1), -7.0, dtype=torch.float64, device='cuda')
        eng.map_cube_device(dsmall, np.float64, 2, dxm, dym, n0, n1, o_small)  # pending, flags: 2 entries
        eng.map_cube_device(dbig, np.float64, 9, dxm, dym, n0, n1, o_big)  # regrow to 9 entries
        eng.synchronize()
        for got, cube in ((o_small, small), (o_big, big)):
            ref = oracle.map_cube(cube, xm, ym, 'linear', True)
            got = got.cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            fin = np.isfinite(ref)
            assert fin.any() and np.max(np.abs(got[fin] - ref[fin])) <= 1e-11
    finally:
        eng.close()


def test_host_path_pageable_pinned_and_zero_copy_agree(engine, oracle, jupiter):
    """
    The host <-> HBM leg of PM_MEM_HOST calls (pm_hostpipe.hip): a pageable cube through the
    three-slot copy pipeline (tiny chunks here: many ring reuses and output drains), a pinned cube
    copied by DMA, a pinned cube gathered in place (zero copy), pageable / pinned outputs, the
    staged D2H of frames - all bit-identical to each other and within the bar of the oracle; planes
    that need the nanmedian (-inf block) are redone behind the pipeline.
    """
    from planetmapper_amd import _lib

    sz, planes = 256, 37
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.9 * x0, 0.2, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 2.0)
    d = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
    d.rotation_rad = 0.2
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    rng = np.random.default_rng(99)
    cube = rng.standard_normal((planes, sz, sz)) * 4 + 1
    cube[rng.random(cube.shape) < 2e-3] = np.nan
    cube[5][90:150, 70:190] = -np.inf
    cube[30][:] = np.nan
    try:
        engine.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, 1 << 20)  # 2 planes of 512 KiB per chunk
        for interp in ('linear', 'nearest'):
            ref = oracle.map_cube(cube, xm, ym, interp, True)
            a = engine.map_cube(cube, xm, ym, interp, True)
            assert np.array_equal(np.isnan(a), np.isnan(ref)) and np.array_equal(np.isinf(a), np.isinf(ref))
            fin = np.isfinite(ref)
            assert np.max(np.abs(a[fin] - ref[fin])) <= 1e-11
            pc = engine.pinned_copy(cube)
            # pageable cube: whole planes through the copy pipeline (0), block table collected by the
            # copy threads (3; what the default picked above for this coarse map)
            for zc in (0, 3):
                engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
                assert np.array_equal(a, engine.map_cube(cube, xm, ym, interp, True), equal_nan=True), (interp, zc)
            engine.set_option(_lib.PM_OPT_ZERO_COPY, 0)
            b = engine.map_cube(pc, xm, ym, interp, True)
            assert np.array_equal(a, b, equal_nan=True), interp
            # gathered in place (1), through the table of sampled blocks fetched by the GPU (2) /
            # collected by the copy threads (3), the library's choice (-1)
            # and the hybrid of the last two (4)
            for zc in (1, 2, 3, 4, -1):
                engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
                c = engine.map_cube(pc, xm, ym, interp, True)
                assert np.array_equal(a, c, equal_nan=True), (interp, zc)
            # pinned output: the zero-copy kernel stores straight into it
            po = engine.pinned_empty(a.shape)
            po[...] = -3.0
            engine._check(engine._lib.pm_map_cube(
                engine._ctx, pc.ctypes.data, 0, planes, xm.ctypes.data, ym.ctypes.data, xm.shape[0], xm.shape[1],
                _lib.PM_INTERP_LINEAR if interp == 'linear' else _lib.PM_INTERP_NEAREST, 1, po.ctypes.data, _lib.PM_MEM_HOST,
            ))  # fmt: skip
            assert np.array_equal(po, a, equal_nan=True), interp
            del pc, po
        # int16 planes, default chunking; every way of reading the pinned cube
        engine.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, 32 << 20)
        ci = (rng.standard_normal((planes, sz, sz)) * 1000).astype(np.int16)
        pci = engine.pinned_copy(ci)
        ri = engine.map_cube(ci, xm, ym)
        for zc in (-1, 0, 1, 2, 3, 4):
            engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
            assert np.array_equal(ri, engine.map_cube(pci, xm, ym), equal_nan=True), zc
            assert np.array_equal(ri, engine.map_cube(ci, xm, ym), equal_nan=True), zc
        engine.set_option(_lib.PM_OPT_ZERO_COPY, -1)
        assert np.max(np.abs(np.nan_to_num(engine.map_cube(ci, xm, ym) - oracle.map_cube(ci, xm, ym)))) <= 1e-9
        # planes that are not a whole number of 256-byte blocks (uint8, 250 x 250), and a map fine
        # enough to touch all of a plane (the library then copies whole planes)
        c8 = rng.integers(0, 255, (5, 250, 250), dtype=np.uint8)
        d8 = oracle.make_disc(124.5, 124.5, 110.0, 0.0, 250, 250)
        xm8, ym8 = oracle.xy_map(jupiter, d8, lon, lat)
        lonf, latf = oracle.rectangular_grid(jupiter, 0.25)
        xmf, ymf = oracle.xy_map(jupiter, d, lonf, latf)
        p8, p6 = engine.pinned_copy(c8), engine.pinned_copy(cube[:6])
        fine = engine.map_cube(cube[:6], xmf, ymf)
        for zc in (-1, 1, 2, 3, 4):
            engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
            assert np.array_equal(fine, engine.map_cube(p6, xmf, ymf), equal_nan=True), zc
            assert np.array_equal(fine, engine.map_cube(cube[:6], xmf, ymf), equal_nan=True), zc
        engine.set_disc(124.5, 124.5, 110.0, 0.0, 250, 250, True)
        small = engine.map_cube(c8, xm8, ym8)
        assert np.max(np.abs(np.nan_to_num(small - oracle.map_cube(c8, xm8, ym8)))) <= 1e-11
        for zc in (-1, 1, 2, 3, 4):
            engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
            assert np.array_equal(small, engine.map_cube(p8, xm8, ym8), equal_nan=True), zc
            assert np.array_equal(small, engine.map_cube(c8, xm8, ym8), equal_nan=True), zc
        engine.set_option(_lib.PM_OPT_ZERO_COPY, -1)
        # frames: staged D2H into pageable arrays == DMA into pinned arrays
        names = ['LON-GRAPHIC', 'EMISSION', 'RA', 'RING-RADIUS']
        engine.set_disc(700.3, 511.0, 480.0, 0.4, 1400, 1100, True)  # 12 MB planes: several staging pieces
        pag = engine.backplanes_img(names)
        import ctypes

        from planetmapper_amd.engine import PLANE_INDEX, plane_mask

        pin = {n: engine.pinned_empty((1100, 1400)) for n in names}
        ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
        for n, arr in pin.items():
            ptrs[PLANE_INDEX[n]] = arr.ctypes.data
        engine._check(engine._lib.pm_backplanes_img(engine._ctx, plane_mask(names), 0.0, ptrs, _lib.PM_MEM_HOST))
        dev = {n: __import__('torch').empty((1100, 1400), dtype=__import__('torch').float64, device='cuda') for n in names}
        engine.backplanes_img_device(dev)
        engine.synchronize()
        for n in names:
            assert np.array_equal(pag[n], pin[n], equal_nan=True), n
            assert np.array_equal(pag[n], dev[n].cpu().numpy(), equal_nan=True), n
    finally:
        engine.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, 32 << 20)
        engine.set_option(_lib.PM_OPT_ZERO_COPY, -1)


def test_sparse_frame_transfer_equals_the_whole_planes(engine_fg, jupiter, saturn):
    """
    Image planes into host memory (pm_hostpipe.hip, PM_OPT_SPARSE_FRAME): the disc planes are NaN
    outside the radius pre-mask, so only bands of rows around that circle are copied (rectangles) and the
    copy threads write the NaN - bit-identical to whole-plane copies and to the device buffers, for discs
    in the middle of, on the edge of and outside the frame, row blocks, pinned and pageable arrays, with
    planes that are NOT NaN outside the circle (RA, ring radius) in the same call.
    """
    engine = engine_fg  # both image kernels: the pre-mask is theirs
    import ctypes

    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import PLANE_INDEX, plane_mask

    names = ['LON-GRAPHIC', 'EMISSION', 'AZIMUTH', 'DISTANCE', 'DOPPLER', 'RA', 'RING-RADIUS', 'LIMB-DISTANCE', 'LOCAL-SOLAR-TIME']
    cases = [  # nx, ny, x0, y0, r0
        (2048, 2048, 1023.5, 1023.5, 700.0), (1800, 1300, 400.0, 900.0, 350.0), (1500, 1100, -200.0, 500.0, 400.0),
        (1400, 1200, 700.0, 1600.0, 300.0), (1300, 1024, 3000.0, 3000.0, 100.0), (900, 1700, 450.0, 850.0, 40.0),
        (1024, 1024, 511.5, 511.5, 3000.0),
    ]  # fmt: skip
    try:
        for k, (nx, ny, x0, y0, r0) in enumerate(cases):
            engine.set_geometry(saturn if k % 3 == 2 else jupiter)
            engine.set_disc(x0, y0, r0, 0.3 * k, nx, ny, True)
            dev = {n: torch.empty((ny, nx), dtype=torch.float64, device='cuda') for n in names}
            engine.backplanes_img_device(dev)
            engine.synchronize()
            ref = {n: dev[n].cpu().numpy() for n in names}
            for mode in (1, 0, -1):
                engine.set_option(_lib.PM_OPT_SPARSE_FRAME, mode)
                got = engine.backplanes_img(names)
                for n in names:
                    assert np.array_equal(got[n], ref[n], equal_nan=True), (k, mode, n)
            # pinned arrays, and a block of rows
            engine.set_option(_lib.PM_OPT_SPARSE_FRAME, 1)
            pin = {n: engine.pinned_empty((ny, nx)) for n in names}
            ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
            for n, arr in pin.items():
                arr[...] = -7.0
                ptrs[PLANE_INDEX[n]] = arr.ctypes.data
            engine._check(engine._lib.pm_backplanes_img(engine._ctx, plane_mask(names), 0.0, ptrs, _lib.PM_MEM_HOST))
            for n in names:
                assert np.array_equal(pin[n], ref[n], equal_nan=True), (k, 'pinned', n)
            r_lo, r_n = ny // 3, ny // 2
            blk = {n: np.full((r_n, nx), -7.0) for n in names}
            for n, arr in blk.items():
                ptrs[PLANE_INDEX[n]] = arr.ctypes.data
            engine._check(engine._lib.pm_backplanes_img_rows(engine._ctx, plane_mask(names), 0.0, r_lo, r_n, ptrs, _lib.PM_MEM_HOST))
            for n in names:
                assert np.array_equal(blk[n], ref[n][r_lo : r_lo + r_n], equal_nan=True), (k, 'rows', n)
    finally:
        engine.set_option(_lib.PM_OPT_SPARSE_FRAME, -1)


@pytest.mark.parametrize('dtype', [np.float64, np.float32, np.uint16])
def test_host_cube_block_table_at_config5_plane_size(engine, oracle, jupiter, dtype):
    """
    BASELINE config 5 geometry (1024^2 planes, 1 deg map) with 24 planes from host memory: the default
    route (table of the sampled 16-byte blocks, marked by k_mark_blocks, collected by the copy threads)
    against whole planes by DMA and against the device-resident kernel - bit-identical - and against
    the oracle on a few planes; NaN pixels, a -inf patch (plane redone with its nanmedian) and an
    all-NaN plane included.
    """
    import torch

    from planetmapper_amd import _lib

    sz, planes = 1024, 24
    x0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 1.0)
    xm, ym = engine.xy_map(lon, lat)
    rng = np.random.default_rng(512)
    if np.issubdtype(dtype, np.floating):
        cube = (rng.standard_normal((planes, sz, sz)) * 3 + 2).astype(dtype)
        cube[rng.random(cube.shape) < 1e-3] = np.nan
        cube[7][400:520, 380:640] = -np.inf
        cube[19][:] = np.nan
    else:
        cube = rng.integers(0, 60000, (planes, sz, sz)).astype(dtype)
    try:
        for interp in ('linear', 'nearest'):
            engine.set_option(_lib.PM_OPT_ZERO_COPY, 0)
            whole = engine.map_cube(cube, xm, ym, interp, True)
            for zc in (-1, 3):
                engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
                assert np.array_equal(whole, engine.map_cube(cube, xm, ym, interp, True), equal_nan=True), (interp, zc)
            # a pinned cube: fetched by the GPU (2) and the hybrid of fetched and collected chunks (4: 24 planes are
            # a fetched chunk of 8, a collected one of 8 and another fetched one)
            pinned = engine.pinned_copy(cube)
            for zc in (2, 4):
                engine.set_option(_lib.PM_OPT_ZERO_COPY, zc)
                assert np.array_equal(whole, engine.map_cube(pinned, xm, ym, interp, True), equal_nan=True), (interp, zc)
                assert engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) == zc
            del pinned
            n0, n1 = xm.shape
            out = torch.empty((planes, n0, n1), dtype=torch.float64, device='cuda')
            raw = cube.view(np.int16) if dtype == np.uint16 else cube  # (bytes only: torch need not know uint16)
            dcube, dxm, dym = torch.from_numpy(raw).cuda(), torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
            torch.cuda.synchronize()
            engine.map_cube_device(dcube, dtype, planes, dxm, dym, n0, n1, out, interp, True)
            engine.synchronize()
            assert np.array_equal(whole, out.cpu().numpy(), equal_nan=True), interp
            pick = [0, 7, 19, 23]
            ref = oracle.map_cube(cube[pick], xm, ym, interp, True)
            got = whole[pick]
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
            fin = np.isfinite(ref)
            assert np.max(np.abs(got[fin] - ref[fin])) <= 1e-9 * max(1.0, float(np.max(np.abs(ref[fin]))))
    finally:
        engine.set_option(_lib.PM_OPT_ZERO_COPY, -1)


def test_c_abi_sharded_cube_with_an_rccl_communicator(engine, oracle, jupiter):
    """
    pm_comm_* / pm_map_cube_sharded (the C-ABI form of the plane sharding; multi-rank runs need a
    multi-GPU node, so here: a real RCCL communicator of ONE rank, the block arithmetic, the NaN
    padding of short blocks, device and host-fed cubes, gather on and off).
    """
    import torch

    from planetmapper_amd.distributed import Comm, shard_bounds

    sz, planes = 128, 5
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.9 * x0, 0.1, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 6.0)
    d = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
    d.rotation_rad = 0.1
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    n0, n1 = xm.shape
    rng = np.random.default_rng(3)
    cube = rng.standard_normal((planes, sz, sz))
    cube[rng.random(cube.shape) < 1e-2] = np.nan
    ref = oracle.map_cube(cube, xm, ym, 'linear', True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    dxm, dym, dcube = t(xm), t(ym), t(cube)
    comm = Comm(engine, 1, 0, Comm.unique_id())
    try:
        a, b, per_rank = shard_bounds(planes, 1, 0)
        out = torch.full((1, per_rank, n0, n1), -1.0, dtype=torch.float64, device='cuda')
        comm.map_cube_sharded(dcube, np.float64, planes, dxm, dym, n0, n1, out)
        engine.synchronize()
        got = out.cpu().numpy()[0]
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.nanmax(np.abs(got - ref)) <= 1e-12
        out.fill_(-1.0)
        comm.map_cube_sharded(engine.pinned_copy(cube), np.float64, planes, dxm, dym, n0, n1, out, host_cube=True, gather=False)
        engine.synchronize()
        assert np.array_equal(out.cpu().numpy()[0], got, equal_nan=True)
        out.fill_(-1.0)
        comm.map_cube_sharded(cube, np.float64, planes, dxm, dym, n0, n1, out, host_cube=True)  # pageable block
        engine.synchronize()
        assert np.array_equal(out.cpu().numpy()[0], got, equal_nan=True)
    finally:
        comm.close()


def test_fused_mapped_data_equals_the_two_calls(engine, oracle, jupiter):
    """
    pm_mapped_data (x/y map + reprojection in one launch for <= 8 planes) against pm_xy_map followed by
    pm_map_cube: identical x/y maps and mapped planes for nearest / linear with both NaN policies,
    planes that need their nanmedian (finished by pm_synchronize), an altitude, and the fall-back to
    the two calls beyond 8 planes.
    """
    import torch

    sz = 200
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0 + 3.3, y0 - 7.1, 0.8 * x0, 0.6, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 3.0)
    lon[5, 7] = np.nan  # a hole in the grid (manual grids may have them)
    n0, n1 = lon.shape
    rng = np.random.default_rng(17)
    cube = rng.standard_normal((11, sz, sz)) + 2.0
    cube[rng.random(cube.shape) < 5e-3] = np.nan
    cube[2][60:120, 50:150] = np.inf
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    dlon, dlat, dcube = t(lon), t(lat), t(cube)
    for alt in (0.0, 2500.0):
        for interp in ('linear', 'nearest'):
            for prop in (True, False):
                for planes in (1, 3, 8, 11):
                    xa, ya = (torch.empty((n0, n1), dtype=torch.float64, device='cuda') for _ in range(2))
                    xb, yb = torch.full_like(xa, -5.0), torch.full_like(xa, -5.0)
                    oa = torch.full((planes, n0, n1), -9.0, dtype=torch.float64, device='cuda')
                    ob = torch.full_like(oa, -9.0)
                    engine.xy_map_device(dlon, dlat, n0, n1, xa, ya, alt=alt)
                    engine.map_cube_device(dcube, np.float64, planes, xa, ya, n0, n1, oa, interp, prop)
                    engine.synchronize()
                    engine.mapped_data_device(dcube, np.float64, planes, dlon, dlat, n0, n1, xb, yb, ob, interp, prop, alt=alt)
                    engine.synchronize()
                    key = (alt, interp, prop, planes)
                    assert torch.equal(torch.nan_to_num(xa, nan=-1.0), torch.nan_to_num(xb, nan=-1.0)), key
                    assert torch.equal(torch.nan_to_num(ya, nan=-1.0), torch.nan_to_num(yb, nan=-1.0)), key
                    assert torch.equal(torch.nan_to_num(oa, nan=-1.0, posinf=7e300, neginf=-7e300),
                                       torch.nan_to_num(ob, nan=-1.0, posinf=7e300, neginf=-7e300)), key
    # and against the oracle
    d = oracle.make_disc(x0 + 3.3, y0 - 7.1, 0.8 * x0, 0.0, sz, sz)
    d.rotation_rad = 0.6
    xr, yr = oracle.xy_map(jupiter, d, lon, lat, alt=2500.0)
    assert np.array_equal(np.isnan(xr), np.isnan(xb.cpu().numpy()))
    ref = oracle.map_cube(cube, xr, yr, 'nearest', False)
    got = ob.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))


def test_point_transforms_vs_oracle(engine, oracle, jupiter):
    """
    pm_transform (reference xy2lonlat, lonlat2radec, ... on arrays) against the oracle:
    every pair of coordinate systems on a cloud of points, altitude and flag variants.
    """
    nx, ny = 300, 200
    engine.set_geometry(jupiter)
    engine.set_disc(140.5, 90.25, 80.0, float(np.deg2rad(33.0)), nx, ny, True)
    d = oracle.make_disc(140.5, 90.25, 80.0, 33.0, nx, ny)
    rng = np.random.default_rng(11)
    n = 20000
    xy = (rng.uniform(-20, nx + 20, n), rng.uniform(-20, ny + 20, n))
    pts = {'xy': xy}
    for cs in ('radec', 'angular', 'km', 'lonlat'):
        pts[cs] = oracle.transform(jupiter, d, 'xy', cs, *xy)
    pts['lonlat'] = (rng.uniform(0, 360, n), rng.uniform(-90, 90, n))
    ps = jupiter.diameter_arcsec / 160.0
    tol = {'xy': 2e-9 / ps * 4, 'radec': 1e-12, 'angular': 3e-9, 'km': 1e-5}
    for src in pts:
        for dst in ('xy', 'radec', 'angular', 'km', 'lonlat'):
            for kw in ({}, {'alt': 543.21}, {'planetocentric': True}, {'not_visible_nan': True}):
                if 'not_visible_nan' in kw and src != 'lonlat':
                    continue
                if kw and 'lonlat' not in (src, dst):
                    continue
                a = engine.transform(src, dst, *pts[src], **kw)
                b = oracle.transform(jupiter, d, src, dst, *pts[src], **kw)
                assert np.array_equal(np.isnan(a[0]), np.isnan(b[0])), (src, dst, kw)
                fin = np.isfinite(b[0])
                assert fin.sum() > 100, (src, dst, kw)
                if dst == 'lonlat':
                    # conditioned like the image planes: compare on the sphere, scaled by the
                    # emission-angle amplification
                    lon_a, lat_a, lon_b, lat_b = (np.deg2rad(v[fin]) for v in (a[0], a[1], b[0], b[1]))
                    dang = np.hypot((lon_a - lon_b + np.pi) % (2 * np.pi) - np.pi, lat_a - lat_b) * np.cos(lat_b)
                    em = np.deg2rad(
                        oracle.backplanes_map(jupiter, d, ['EMISSION'], b[0][fin][None], b[1][fin][None])['EMISSION'][0]
                    ) if 'planetocentric' not in kw and 'alt' not in kw else None
                    if em is not None:
                        assert np.all(np.rad2deg(dang) <= 5e-9 / np.clip(np.abs(np.cos(em)), 1e-7, None)), (src, dst)
                    else:
                        assert np.median(np.rad2deg(dang)) < 1e-9 and np.max(np.rad2deg(dang)) < 1e-5, (src, dst, kw)
                else:
                    for u, v in zip(a, b):
                        dd = np.abs(u[fin] - v[fin])
                        if dst == 'radec':
                            dd = np.minimum(dd, 360 - dd)
                        assert dd.max() <= tol[dst], (src, dst, kw, dd.max())
    # scalars and broadcasting through the reference-compatible API
    from planetmapper_amd import BodyXY

    body = BodyXY('Jupiter', geometry=jupiter, nx=15, ny=10)
    body.set_disc_params(5, 8, 3, 45)
    assert np.allclose(body.xy2lonlat(5, 8), (153.1235185909613, -3.0887371238645795))
    assert np.allclose(body.lonlat2xy(42, 23.4, alt=1234.567, not_visible_nan=False), (7.829968623728911, 8.017815455484365))
    assert np.isnan(body.xy2lonlat(0, 0)[0])


def test_spline_interpolation_vs_oracle_and_goldens(engine, oracle, jupiter):
    """
    'quadratic' / 'cubic' / mixed-degree map_img (RectBivariateSpline s=0,
    body_xy.py:1651-1702): golden FITS map_rectangular-quadratic / -cubic and the oracle
    (itself equal to scipy to 1e-15) on a larger cube with NaNs.
    """
    from planetmapper_amd import Observation

    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    obs = Observation(data=cube, geometry=jupiter)
    obs.set_disc_params(2.5, 3.1, 3.9, 123.456)
    for name in ('quadratic', 'cubic'):
        gold = np.load(os.path.join(GOLDEN, f'golden_map_rectangular_{name}.npz'))['PRIMARY']
        got = obs.get_mapped_data(name, degree_interval=30)
        from parity import check_mapped_against_golden

        check_mapped_against_golden(got, gold, name, f'HIP map_rectangular_{name}')
    sz = 200
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.9 * x0, 0.2, sz + 13, sz, True)
    d = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz + 13, sz)
    d.rotation_rad = 0.2
    lon, lat = oracle.rectangular_grid(jupiter, 3.0)
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    rng = np.random.default_rng(3)
    cube = rng.standard_normal((5, sz, sz + 13)) * 5
    cube[0][rng.random((sz, sz + 13)) < 0.01] = np.nan
    cube[1][50:90, 60:120] = np.nan
    cube[2][:] = np.nan
    cube[3][rng.random((sz, sz + 13)) < 0.01] = np.inf
    for interp in ('quadratic', 'cubic', (1, 3), (2, 1), 4):
        for prop in (True, False):
            a = engine.map_cube(cube, xm, ym, interp, prop)
            b = oracle.map_cube(cube, xm, ym, interp if not isinstance(interp, int) else (interp, interp), prop)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (interp, prop)
            fin = np.isfinite(b)
            scale = np.maximum(1.0, np.abs(b[fin]))
            assert np.max(np.abs(a[fin] - b[fin]) / scale) <= 1e-9, (interp, prop)
    c32 = cube[0].astype(np.float32)
    a = engine.map_cube(c32, xm, ym, 'cubic', True)
    b = oracle.map_cube(c32, xm, ym, 'cubic', True)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) <= 1e-6
    with pytest.raises(ValueError):
        engine.map_cube(cube, xm, ym, (0, 1), True)
    with pytest.raises(ValueError):
        engine.map_cube(cube, xm, ym, 'bicubic', True)


def test_smooth_interpolation_vs_oracle_kats_and_golden(engine, oracle, jupiter):
    """
    'smooth' map_img (PCHIP oversampling + bilinear, body_xy.py:1704-1853): the reference's
    own expected values (tests/test_body_xy.py:1290-1372), its golden FITS
    map_rectangular-smooth, and the oracle (whose PCHIP equals scipy's bit for bit) on a
    larger cube with NaN pixels / blocks / rows / columns, +-inf, integer and f32 planes.
    The GPU evaluates only the four fine-grid nodes around each sample, the oracle builds
    the whole oversampled image like the reference: agreement to 1e-9 relative.
    """
    from planetmapper_amd import BodyXY, Observation
    from test_api_host import IMAGE, _smooth_kats, smooth_test_image

    body = BodyXY('Jupiter', geometry=jupiter, engine=engine)
    body.set_img_size(6, 5)
    body.set_disc_params(2.75, 1.3, 2.3, 45.678)
    for kw, exp in _smooth_kats('map_img_6x5'):
        got = body.map_img(IMAGE, degree_interval=45, interpolation='smooth', **kw)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), kw
    assert np.isnan(body.map_img(IMAGE * np.nan, degree_interval=45, interpolation='smooth')).all()
    body.set_img_size(90, 120)
    body.set_disc_params(32.1, 50, 12, 98.76)
    image = smooth_test_image()
    for kw, exp in _smooth_kats('map_img_90x120'):
        got = body.map_img(image, degree_interval=45, interpolation='smooth', **kw)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), kw

    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    obs = Observation(data=cube, geometry=jupiter, engine=engine)
    obs.set_disc_params(2.5, 3.1, 3.9, 123.456)
    gold = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_smooth.npz'))['PRIMARY']
    got = obs.get_mapped_data('smooth', degree_interval=30)
    from parity import check_mapped_against_golden

    check_mapped_against_golden(got, gold, 'smooth', 'HIP map_rectangular_smooth')

    sz = 200
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.7 * x0, 0.2, sz + 13, sz, True)
    d = oracle.make_disc(x0, y0, 0.7 * x0, 0.0, sz + 13, sz)
    d.rotation_rad = 0.2
    lon, lat = oracle.rectangular_grid(jupiter, 3.0)
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    rng = np.random.default_rng(11)
    cube = rng.standard_normal((7, sz, sz + 13)) * 5
    cube[0][rng.random((sz, sz + 13)) < 0.02] = np.nan
    cube[1][50:90, 60:120] = np.nan
    cube[2][:] = np.nan
    cube[3][rng.random((sz, sz + 13)) < 0.01] = np.inf
    cube[4][100, :] = np.nan
    cube[4][:, 77] = np.nan
    cube[5][:, :] = np.nan
    cube[5][60:140:7, 40:180:5] = rng.standard_normal(cube[5][60:140:7, 40:180:5].shape)  # sparse samples
    for kw in ({}, dict(smooth_oversample_by=1), dict(smooth_oversample_by=3),
               dict(smooth_oversample_by=10, smooth_max_oversampled_img_size=700)):  # fmt: skip
        for prop in (True, False):
            a = engine.map_cube(cube, xm, ym, 'smooth', prop, **kw)
            b = oracle.map_cube(cube, xm, ym, 'smooth', prop, **kw)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (kw, prop)
            assert np.isfinite(b).sum() > 1000
            fin = np.isfinite(b)
            scale = np.maximum(1.0, np.abs(b[fin]))
            assert np.max(np.abs(a[fin] - b[fin]) / scale) <= 1e-9, (kw, prop)
    for dt in (np.float32, np.int16, np.uint8):
        c = (np.nan_to_num(cube[0]) * 10).astype(dt) if dt is not np.float32 else cube[0].astype(dt)
        a = engine.map_cube(c, xm, ym, 'smooth', True)
        b = oracle.map_cube(c, xm, ym, 'smooth', True)
        assert np.array_equal(np.isnan(a), np.isnan(b)), dt
        assert np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= 1e-9, dt
    # a map with no visible cell: all-NaN output (the Python layer raises like the reference)
    a = engine.map_cube(cube[:2], xm * np.nan, ym * np.nan, 'smooth', True)
    assert np.isnan(a).all()
    with pytest.raises(IndexError):
        body.map_img(image, projection='manual', lon_coords=np.full((2, 2), np.nan), lat_coords=np.full((2, 2), np.nan),
                     interpolation='smooth')  # fmt: skip
    # device-resident call (the body above re-bound its own disc to the shared engine)
    import torch

    engine.set_disc(x0, y0, 0.7 * x0, 0.2, sz + 13, sz, True)
    dc = torch.from_numpy(cube).cuda()
    dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    dout = torch.empty((cube.shape[0],) + xm.shape, dtype=torch.float64, device='cuda')
    engine.set_smooth_options(5, 10_000)
    engine.map_cube_device(dc, np.float64, cube.shape[0], dx, dy, xm.shape[0], xm.shape[1], dout, 'smooth', True)
    engine.synchronize()
    assert np.array_equal(dout.cpu().numpy(), engine.map_cube(cube, xm, ym, 'smooth', True), equal_nan=True)


def test_save_observation_and_map_on_gpu(engine, oracle, jupiter, tmp_path):
    """
    The callers that want every plane (observation.py:1184-1474) on the real engine: a 257 x 193
    frame written with all 26 backplanes and its 5 deg map, read back and compared with the oracle.
    """
    from planetmapper_amd import Observation, fits_io

    rng = np.random.default_rng(8)
    cube = rng.standard_normal((3, 193, 257)).astype(np.float32)
    obs = Observation(data=cube, target='jupiter', geometry=jupiter, engine=engine)
    obs.set_disc_params(120.3, 99.1, 80.5, 33.0)
    out = os.path.join(tmp_path, 'nav.fits')
    obs.save_observation(out, print_info=False)
    hdus = fits_io.read(out)
    assert [h.name for h in hdus[1:]] == list(obs.backplanes)
    assert hdus[0].data.dtype == np.float32 and np.array_equal(hdus[0].data, cube)
    d = oracle.make_disc(120.3, 99.1, 80.5, 33.0, 257, 193)
    ref = oracle.backplanes_img(jupiter, d, oracle.PLANE_NAMES)
    _compare({h.name: h.data for h in hdus[1:]}, ref, oracle.PLANE_NAMES, jupiter, r0=80.5)
    out = os.path.join(tmp_path, 'map.fits')
    obs.save_mapped_observation(out, degree_interval=5, print_info=False)
    hdus = fits_io.read(out)
    lon, lat = oracle.rectangular_grid(jupiter, 5.0)
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    # white noise has unit gradients per pixel: the 1e-10 px differences of the x/y maps near
    # the limb show up at that level in the samples
    assert np.array_equal(np.isnan(hdus[0].data), np.isnan(oracle.map_cube(cube, xm, ym)))
    assert np.allclose(hdus[0].data, oracle.map_cube(cube, xm, ym), rtol=1e-7, atol=1e-7, equal_nan=True)
    gx, gy = obs.get_x_map(degree_interval=5), obs.get_y_map(degree_interval=5)
    assert np.allclose(hdus[0].data, oracle.map_cube(cube, gx, gy), rtol=1e-12, atol=1e-12, equal_nan=True)
    assert hdus[0].header['PLANMAP MAP DEGREE-INTERVAL'] == 5 and hdus[0].header['CDELT1'] == -5.0
    assert [h.name for h in hdus[1:]] == list(obs.backplanes)


def _variant(g, **changes):
    v = g.copy()
    for k, val in changes.items():
        if isinstance(val, (list, tuple)):
            arr = getattr(v, k)
            for i, x in enumerate(val):
                arr[i] = x
        else:
            setattr(v, k, val)
    return v


def _with_its_own_sub_observer_point(g):
    """
    The sub-observer fields of a geometry block (subpnt 'INTERCEPT/ELLIPSOID', body.py:538-555: the point, its ray, its
    epoch and distance - the pivot of PM's obsvec <-> targvec transforms) recomputed from the block's own motion model,
    for variants whose radii are not the fixture's: a pivot 60 000 km off the surface of a 9000-km body is not a geometry
    the reference can produce, and it levers one epoch quantum of the limb / ring planes up by |pivot| / |point|.
    """
    v = g.copy()
    R0 = np.array(v.R0[:]).reshape(3, 3)
    T0, VT, AT = (np.array(x[:]) for x in (v.T0, v.VT, v.AT))
    radii = np.array(v.radii[:])
    t0 = v.et - v.lt_c

    def rot_at(t):
        c, s = np.cos(v.wdot * (t - t0)), np.sin(v.wdot * (t - t0))
        return np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]]) @ R0

    lt = v.lt_c
    for _ in range(12):
        te = v.et - lt
        d = te - t0
        Rk = rot_at(te)
        obs_b = -(Rk @ (T0 + VT * d + 0.5 * AT * d * d))
        sp = obs_b / np.sqrt(np.sum((obs_b / radii) ** 2))
        new = float(np.linalg.norm(sp - obs_b)) / v.clight
        if new == lt:
            break
        lt = new
    ray = sp - obs_b
    ov = Rk.T @ ray
    for i in range(3):
        v.sub_sp[i], v.sub_ray[i], v.sub_obsvec[i] = sp[i], ray[i], ov[i]
    v.sub_et = v.et - lt
    v.sub_dist = float(np.linalg.norm(ray))
    return v


@pytest.mark.parametrize('case', ['east_positive', 'triaxial', 'triaxial_east_small', 'io_like', 'io_like_fast'])
def test_other_body_shapes_and_longitude_conventions(engine, oracle, jupiter, case):
    """
    Bodies the Jupiter / Saturn fixtures do not exercise: east-positive planetographic
    longitudes (prograde convention off: Sun, Earth, Moon in the reference, body.py:520-536)
    and a triaxial ellipsoid (a != b), which takes the general kernel `k_disc`. Geometry blocks
    derived from the Jupiter fixture; every image plane, the map chain and the point
    transforms against the oracle ("parity unpinned" by reference goldens for these).
    """
    if case == 'east_positive':
        g = _variant(jupiter, west_positive=0)
    elif case == 'triaxial':
        g = _variant(jupiter, radii=[71492.0, 69000.0, 66854.0])
    elif case.startswith('io_like'):
        # a real moon's shape and spin (Io: 1829.4 x 1819.4 x 1815.7 km, 1.769 d): the triaxial variant's closed-form
        # light time (Params::tri_cf; the Jupiter-sized triaxial bodies above turn too much under the ray for it and
        # keep the sequence + Newton step). `io_like_fast`: the same body turning 40 times faster - still inside
        # the closed form's guard, the first-order turn 40 times larger (2e-4 km).
        ratio = 1829.4 / jupiter.radii[0]
        g = _variant(jupiter, radii=[1829.4, 1819.4, 1815.7], wdot=4.11e-5 * (40.0 if case.endswith('fast') else 1.0),
                     diameter_arcsec=jupiter.diameter_arcsec * ratio)
    else:
        g = _variant(jupiter, radii=[71492.0, 70100.0, 68800.0], west_positive=0)
    sz, nxs = 301, 333
    x0, y0, r0, rot = 160.2, 141.9, 110.0, 71.0
    engine.set_geometry(g)
    engine.set_disc(x0, y0, r0, float(np.deg2rad(rot)), nxs, sz, True)
    d = oracle.make_disc(x0, y0, r0, rot, nxs, sz)
    d.rotation_rad = float(np.deg2rad(rot))
    from planetmapper_amd import _lib

    # (the module's engine may be inside an `engine_fg[general]` stretch of the session: this test is about the library's
    #  own choice of kernel)
    forced = engine.get_option(_lib.PM_OPT_GENERAL_KERNEL)
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
    try:
        _other_body_checks(engine, oracle, g, d, case, x0, y0, r0, rot, nxs, sz)
    finally:
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, forced)


def _other_body_checks(engine, oracle, g, d, case, x0, y0, r0, rot, nxs, sz):
    for alt in (0.0, 2500.0 if g.radii[0] > 1e4 else 60.0):
        out = engine.backplanes_img(oracle.PLANE_NAMES, alt=alt)
        ref = oracle.backplanes_img(g, d, oracle.PLANE_NAMES, alt=alt)
        _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0, flat=g.radii[0] > 1e4)
        assert 0.2 < np.isfinite(out['LON-GRAPHIC']).mean() < 0.5
        if case.startswith('io_like'):
            from planetmapper_amd import _lib

            assert engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == 2
    lon, lat = oracle.rectangular_grid(g, 4.0)
    assert (lon[0, 0] > lon[0, 1]) == bool(g.west_positive)
    out = engine.backplanes_map(oracle.PLANE_NAMES, lon, lat)
    ref = oracle.backplanes_map(g, d, oracle.PLANE_NAMES, lon, lat)
    _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0)
    rng = np.random.default_rng(2)
    px, py = rng.uniform(0, nxs, 4000), rng.uniform(0, sz, 4000)
    a, b = engine.transform('xy', 'lonlat', px, py)
    ra, rb = oracle.transform(g, d, 'xy', 'lonlat', px, py)
    assert np.array_equal(np.isnan(a), np.isnan(ra)) and np.isfinite(a).sum() > 500
    assert np.nanmax(np.abs(((a - ra + 180) % 360) - 180)) < 1e-6 and np.nanmax(np.abs(b - rb)) < 1e-6
    fin = np.isfinite(a)
    bx, by = engine.transform('lonlat', 'xy', a[fin], b[fin])
    ox, oy = oracle.transform(g, d, 'lonlat', 'xy', a[fin], b[fin])
    assert np.array_equal(np.isnan(bx), np.isnan(ox))
    assert np.nanmax(np.hypot(bx - ox, by - oy)) < 1e-6
    if g.radii[0] == g.radii[1]:
        # (recpgr / pgrrec work on the spheroid (a, f): for a triaxial body the round trip is
        # not the identity in the reference either)
        assert np.nanmax(np.hypot(bx - px[fin], by - py[fin])) < 5e-3  # the reference's own round-trip error


def test_nan_preclean_reference_kats_on_gpu(engine, jupiter):
    """
    The reference's value table for `_replace_nans_with_interpolated_values`
    (tests/test_body_xy.py:1479-1536) through the GPU: with `propagate_nan=False` a bilinear
    sample at an integer pixel position returns that pixel's cleaned value, so an identity map
    reads the whole cleaned image back (window means evaluated on the fly in `k_reproject`,
    plane nanmedian by the radix-select kernels). The all-NaN image is the exception: `map_img`
    returns an all-NaN map before cleaning (body_xy.py:1668-1670).
    """
    import json

    with open(os.path.join(GOLDEN, 'kat_replace_nans.json'), encoding='utf-8') as f:
        cases = json.load(f)['cases']
    engine.set_geometry(jupiter)
    for c in cases:
        img = np.array(c['image'], dtype=float)
        exp = np.array(c['cleaned'], dtype=float)
        ny, nx = img.shape
        engine.set_disc(1.0, 1.0, 1.0, 0.0, nx, ny, True)
        xm, ym = np.meshgrid(np.arange(nx, dtype=float), np.arange(ny, dtype=float))
        got = engine.map_cube(img, xm, ym, 'linear', False)[0]
        if np.isnan(img).all():
            assert np.isnan(got).all()
        else:
            assert np.allclose(got, exp, rtol=1e-12, atol=0), c['image']
        for dt in (np.float32,):
            if np.isfinite(img).all():
                assert np.allclose(engine.map_cube(img.astype(dt), xm, ym, 'linear', False)[0], exp)


def _near_field_geometry(distance_km):
    """Jupiter seen from `distance_km` (an observer like the reference's 'amalthea' test case,
    tests/test_body_xy.py:2592-2607), built by the package's own host geometry provider"""
    from planetmapper_amd.ephem import Ephemeris, RotationModel
    from planetmapper_amd.geometry import CLIGHT, GeometryBuilder
    from planetmapper_amd.scenarios import _load_json

    d = _load_json('jupiter_hst_2005')
    gb = GeometryBuilder(Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck']), d['target_id'])
    h = d['header']
    return gb.build(
        d['et'] + 3 * 3600.0,
        observer_velocity=[-10.0, 28.0, 3.0],
        target_ra_dec_dist_lt=(h['PLANMAP TARGET RA'] + 40.0, h['PLANMAP TARGET DEC'] + 20.0, distance_km,
                               distance_km / CLIGHT),  # fmt: skip
    )


@pytest.mark.parametrize('distance_km', [107_238.0, 181_000.0, 1.2e6])
def test_near_field_observer(engine_fg, oracle, distance_km):
    """
    A close observer: the disc subtends tens of degrees, so the small-angle shortcuts of the
    fast path (series sincos of the view angles, spin angle over the light-time span) are
    outside their range for part of the frame and the wave-uniform fallbacks take over;
    parallax makes the visible cap much smaller than a hemisphere. At 107 238 km = 1.5 equatorial
    radii from the centre the observer is inside two scaled radii, where the library itself
    dispatches the general kernel (`y2 <= 4` in pm_backplanes_img_rows). All planes + map chain vs
    the oracle (no reference golden for this geometry: "parity unpinned").
    """
    g = _near_field_geometry(distance_km)
    assert g.diameter_arcsec > 5 * 3600
    nx, ny = 261, 197
    x0, y0, r0, rot = 128.0, 101.5, 70.0, 200.0
    engine_fg.set_geometry(g)
    engine_fg.set_disc(x0, y0, r0, float(np.deg2rad(rot)), nx, ny, True)
    d = oracle.make_disc(x0, y0, r0, rot, nx, ny)
    d.rotation_rad = float(np.deg2rad(rot))
    out = engine_fg.backplanes_img(oracle.PLANE_NAMES)
    from planetmapper_amd import _lib

    if distance_km < 2 * 66854.0:  # inside two scaled radii: the library's own dispatch
        assert engine_fg.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == 3
    ref = oracle.backplanes_img(g, d, oracle.PLANE_NAMES)
    _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0)
    assert 0.1 < np.isfinite(out['LON-GRAPHIC']).mean() < 0.6
    lon, lat = oracle.rectangular_grid(g, 5.0)
    om = engine_fg.backplanes_map(oracle.PLANE_NAMES, lon, lat)
    rm = oracle.backplanes_map(g, d, oracle.PLANE_NAMES, lon, lat)
    _compare(om, rm, oracle.PLANE_NAMES, g, r0=r0)
    # test_mapping_visible_areas: a cell is mapped exactly where its emission angle is <= 90 deg
    vis = om['EMISSION'] <= 90
    assert np.isfinite(om['RA'][vis]).all() and np.isnan(om['RA'][~vis]).all()
    assert 0.02 < vis.mean() < 0.5


def _compare_allowing_epoch_quantum_flips(out, ref, names, g, r0, label=''):
    """
    `_compare` for geometries on which ONE QUANTUM of the epoch et - lt (ulp(et): 3e-8 s in 2005, 1.2e-7 s from 2015 on)
    is visible: which quantum an epoch rounds to is decided by the last bits of a light time that two implementations
    compute by different routes (1e-12 s apart: a few pixels in 1e5 flip - CSPICE would show as much against itself on
    another machine). The library follows the reference's epochs (Params::plain_lt / cf_iter / turn_quantum); what is left
    is that noise: per plane at most 2e-3 of the pixels beyond the ordinary bar, each within one conditioned quantum of
    it, nothing beyond. Masks identical.
    """
    from parity import base_deg, tolerances

    tol = tolerances(ref, g, plate_scale_arcsec=g.diameter_arcsec / (2 * r0))
    quantum = float(np.spacing(abs(g.et)))  # (np.spacing of a negative number is negative: epochs before 2000)
    q_deg = float(np.rad2deg(quantum * (abs(g.wdot) + np.linalg.norm(g.VT[:]) / min(g.radii[:]))))
    # (the limb and ring planes go through PM's obsvec -> targvec transform, whose epoch sub_et - dd / c is a double too)
    turning = ('LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC', 'LAT-CENTRIC', 'INCIDENCE', 'EMISSION', 'AZIMUTH',
               'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC', 'RING-LON-GRAPHIC')
    flipped = {}
    for n in names:
        assert masks_agree(n, out[n], ref[n]), n
        fin = np.isfinite(ref[n])
        diff = np.abs(out[n] - ref[n])
        if 'LON' in n or n == 'RA':
            diff = np.minimum(diff, 360.0 - diff)
        if n == 'LOCAL-SOLAR-TIME':
            assert np.nanmax(diff) <= 1.0 / 3600 + 1e-12
            continue
        t = np.broadcast_to(tol[n], diff.shape)
        if n in ('RADIAL-VELOCITY', 'DOPPLER'):
            # the intercept is defined to the rounding of the ray (1e-7 km on 7e4 km: 1e-12): so is wdot x r, which
            # is 380 km/s for the fast-spin bodies - the bar of 1e-9 km/s was set for bodies that turn at 12 km/s
            # (measured on a Jupiter-sized body: the median difference grows in proportion to the spin, 1e-11 -> 3e-10 km/s)
            # (in absolute terms: five half-ulps of the unit ray at the target's distance, over cos(emission) as the
            #  intercept slides along a slanted ray, times the spin - 2e-9 km/s for a 400-km body turning in 16 minutes)
            kappa = np.broadcast_to(tol['LAT-GRAPHIC'], diff.shape) / base_deg(g)
            noise = 5.0 * 1.11e-16 * float(np.linalg.norm(g.T0[:])) * abs(g.wdot) * kappa
            # (... and the reference's own value is defined to the rounding of its epoch, half a quantum: wdot^2 r per second
            #  of it. Below the library's visibility threshold - 1e-9 deg of turn per quantum - the kernels do not round the
            #  epoch at all and sit CLOSER to the binary128 value than the oracle: a 62 000-km body turning 47 times faster than
            #  Jupiter in the year 2000, quantum 1.9e-9 s: oracle up to 8e-9 km/s from the exact value, HIP 1.8e-9)
            noise = noise + 0.5 * quantum * g.wdot**2 * max(g.radii[:])
            t = t + (noise if n == 'RADIAL-VELOCITY' else noise / g.clight)
        bad = fin & (diff > t)
        if n in turning:
            allow = t * (1.0 + 1.5 * q_deg / base_deg(g))
        elif n in ('RADIAL-VELOCITY', 'DOPPLER'):
            # the point's velocity turns with the body: wdot^2 r per second of epoch (the target's own acceleration is nothing);
            # and the body moves under the fixed ray, |VT| per second: the intercept slides by that over cos(emission), the
            # spin velocity wdot x r with it (the larger term on a small fast body: 5e-9 km/s on 300 km turning in an hour)
            kappa = np.broadcast_to(tol['LAT-GRAPHIC'], diff.shape) / base_deg(g)
            # (TWO rounded epochs: the intercept is body-fixed at sincpt's epoch, its state is taken at spkcpt's - the spin
            #  velocity follows their difference; measured 1.95 quanta at the centre of a disc, each implementation one side)
            dv = 2.5 * quantum * (g.wdot**2 * max(g.radii[:]) + float(np.linalg.norm(g.VT[:])) * abs(g.wdot) * kappa)
            allow = t + (dv if n == 'RADIAL-VELOCITY' else dv / g.clight)
        elif n == 'DISTANCE':
            allow = t + 1.5 * quantum * float(np.linalg.norm(g.VT[:])) * np.broadcast_to(tol['LAT-GRAPHIC'], diff.shape) / base_deg(g)
        elif n == 'RING-RADIUS':
            # PM's obsvec -> targvec turns the offset from the SUB-OBSERVER POINT: a quantum of its epoch moves the point by
            # wdot x |pivot| (4e-5 km on a Saturn-sized body turning 30 times faster than Saturn)
            allow = t + 1.5 * quantum * abs(g.wdot) * float(np.linalg.norm(g.sub_sp[:]))
        elif n == 'LIMB-DISTANCE':
            # ... and the latitude of the limb point with it: the local radius changes by (a - c) sin(2 lat) per radian
            allow = t + 1.5 * quantum * abs(g.wdot) * (max(g.radii[:]) - min(g.radii[:]))
        else:
            allow = t
        assert not (bad & (diff > allow)).any(), (n, float(np.nanmax(np.where(bad, diff / allow, 0.0))))
        flipped[n] = (int(bad.sum()), int(fin.sum()))
    print(f'\n[{label}] pixels one epoch quantum away (of finite pixels):', {k: v for k, v in flipped.items() if v[0]})
    for n, (nb, nf) in flipped.items():
        assert nb <= max(3, int(2e-3 * nf)), (n, nb, nf)


@pytest.mark.parametrize('case', ['fast_spin', 'fast_spin_triaxial', 'large_acceleration', 'spin_x20', 'spin_x20_triaxial'])
def test_fast_spin_and_large_acceleration_take_the_general_kernel(engine, oracle, jupiter, case):
    """
    The other two conditions under which the library itself leaves the fast paths (pm_backplanes_img_rows): a spin
    angle over a disc's light-time span of 1e-3 rad and more (here: Jupiter turning 30 times faster - the angle goes
    through the range-tiered sincos of k_disc_sph<FLAGS, 2> instead of a series), and a target acceleration whose
    A d^2 / 2 over that span is visible (here: 3e4 times Jupiter's - the quadratic terms of target and Sun).
    All 26 planes against the oracle, which evaluates the same motion model with full rotation matrices.
    """
    from planetmapper_amd import _lib

    if case == 'large_acceleration':
        g = _variant(jupiter, AT=[a * 3e4 for a in jupiter.AT[:]], AS=[a * 3e4 for a in jupiter.AS[:]])
    else:
        # (x20: the spin angle over a light-time span stays below 1e-3 rad, but one epoch quantum is 6e-9 deg of turn)
        g = _variant(jupiter, wdot=jupiter.wdot * (20.0 if case.startswith('spin_x20') else 30.0))
        if case.endswith('triaxial'):
            g = _variant(g, radii=[71492.0, 69800.0, 66854.0])
    nx, ny = 301, 233
    x0, y0, r0, rot = 150.2, 118.0, 96.0, 0.7
    d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
    d.rotation_rad = rot
    with _library_choice(engine):
        engine.set_geometry(g)
        engine.set_disc(x0, y0, r0, rot, nx, ny, True)
        out = engine.backplanes_img(oracle.PLANE_NAMES)
        kernel = engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL)
    print(f'\n[{case}] image kernel {kernel}')
    if not case.startswith('spin_x20'):
        assert kernel == 3  # the library's own dispatch
    ref = oracle.backplanes_img(g, d, oracle.PLANE_NAMES)
    if case == 'large_acceleration':
        _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0, flat=False)
    else:
        # Epochs et - lt are doubles (one quantum: 3e-8 s). At this spin a quantum turns the body by 9e-9 deg, nine
        # times the bar - and which quantum an epoch rounds to is decided by the last bits of a light time that two
        # implementations compute by different routes (here: the scaled rejection form against |T + R^T sp| / c;
        # 1e-12 s apart, so about 3e-5 of the pixels flip - CSPICE would show as much against itself on another
        # machine). The library follows the reference's epochs (Params::plain_lt: its sequence for the intercept, the
        # fixed point for illumf_c / spkcpt_c); what is left is that noise: a handful of pixels exactly one
        # conditioned quantum away, nothing beyond.
        _compare_allowing_epoch_quantum_flips(out, ref, oracle.PLANE_NAMES, g, r0, label='fast spin')
    assert np.isfinite(out['LON-GRAPHIC']).sum() > 20000


@pytest.mark.parametrize('case', ['jupiter_2023', 'small_body_2023', 'saturn_2023'])
def test_closed_form_steps_through_the_reference_iterates_where_a_quantum_is_visible(engine, oracle, jupiter, saturn, case):
    """
    Epochs `et - lt` are doubles; one quantum of them is ulp(et): 3e-8 s for the 2005 fixtures, 1.2e-7 s from 2015
    on - in which Jupiter moves 3e-6 km = 2.6e-9 deg of its own longitude. From then on (and for every small body at
    any epoch) which quantum the reference's final epoch rounds to is visible at the level of the bar, and only its own
    sequence of light-time iterates lands on it. The closed form steps through that sequence from the fixed point
    (`Params::cf_iter`). The same geometry blocks moved to 2023 (every epoch field shifted alike: positions are
    relative, only the quantum changes), against the oracle - which walks the sequence: masks identical, every
    plane inside the ordinary bars, and the image kernel still the fast path.
    """
    from planetmapper_amd import _lib

    shift = 7.2e8 - jupiter.et
    base = saturn if case == 'saturn_2023' else jupiter
    g = _variant(base, et=base.et + shift, ts0=base.ts0 + shift, sub_et=base.sub_et + shift)
    if case == 'small_body_2023':
        ratio = 1200.0 / jupiter.radii[0]  # a 1200-km spheroid: one quantum is 1.6e-7 deg of longitude on it
        g = _variant(g, radii=[1200.0, 1200.0, 1170.0], diameter_arcsec=jupiter.diameter_arcsec * ratio)
    sz = 1024
    x0, y0, r0, rot = 505.3, 517.9, 440.0, 0.3
    names = HEADLINE + ['LON-CENTRIC', 'LAT-CENTRIC', 'AZIMUTH', 'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER', 'LOCAL-SOLAR-TIME']
    d = oracle.make_disc(x0, y0, r0, 0.0, sz, sz)
    d.rotation_rad = rot
    with _library_choice(engine):
        engine.set_geometry(g)
        engine.set_disc(x0, y0, r0, rot, sz, sz, True)
        out = engine.backplanes_img(names)
        assert engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == 1
        assert engine.get_option(_lib.PM_OPT_LAST_LT_PATH) & 3 == 3  # closed form, stepping through the reference's iterates
    oracle.set_num_threads(16)
    ref = oracle.backplanes_img(g, d, names)
    _compare_allowing_epoch_quantum_flips(out, ref, names, g, r0, label=case)
    # the plane sets with their own compiled kernel (BASELINE's headline five, config 4's eight, the intercept group of
    # save_observation) have a QUANT instantiation each: the same planes, bit for bit, as the general-mask kernel gave
    ring = ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
    disc_group = HEADLINE + ring + ['LON-CENTRIC', 'LAT-CENTRIC', 'AZIMUTH', 'LOCAL-SOLAR-TIME', 'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER']
    with _library_choice(engine):
        for subset in (HEADLINE, HEADLINE + ring, disc_group):
            sub = engine.backplanes_img(subset)
            for n in subset:
                if n in out:
                    assert np.array_equal(sub[n], out[n], equal_nan=True), (len(subset), n)
            if subset is disc_group:
                ref_ring = oracle.backplanes_img(g, d, ring)
                _compare_allowing_epoch_quantum_flips({n: sub[n] for n in ring}, ref_ring, ring, g, r0, label=case + ' rings')
    # the map direction at that epoch (its light-time offsets go through PM's own transforms: body.py:917-1006)
    lon, lat = oracle.rectangular_grid(g, 5.0)
    with _library_choice(engine):
        om = engine.backplanes_map(oracle.PLANE_NAMES, lon, lat)
    rm = oracle.backplanes_map(g, d, oracle.PLANE_NAMES, lon, lat)
    _compare_allowing_epoch_quantum_flips(om, rm, oracle.PLANE_NAMES, g, r0, label=case + ' map')
    # ... and the plain sequence (PM_OPT_LT_MODE 1) gives the same answer as the stepped closed form
    from planetmapper_amd.engine import Engine

    e2 = Engine(0)
    try:
        e2.set_option(_lib.PM_OPT_LT_MODE, 1)
        e2.set_geometry(g)
        e2.set_disc(x0, y0, r0, rot, sz, sz, True)
        seq = e2.backplanes_img(names)
    finally:
        e2.close()
    quantum_deg = float(np.rad2deg(np.spacing(g.et) * np.linalg.norm(g.VT[:]) / min(g.radii[:])))
    lon = np.abs(out['LON-GRAPHIC'] - seq['LON-GRAPHIC'])
    lon = np.minimum(lon, 360.0 - lon)
    # (away from the limb and the poles, where a quantum stands out from the rounding of the intercept itself)
    on = np.isfinite(lon) & (ref['EMISSION'] < 60.0) & (np.abs(ref['LAT-GRAPHIC']) < 60.0)
    # (the two walk the same epochs: the few pixels that differ by a quantum are the borderline lanes)
    share = float(np.mean(lon[on] > 0.5 * quantum_deg))
    print(f'\n[{case}] one quantum = {quantum_deg:.2e} deg of longitude; pixels where closed form and sequence chose another: {share:.2e}')
    # (measured: Jupiter 0, Saturn 9e-4 - where the stepped closed form is the one closer to the oracle, 0.57 % against 0.93 %
    #  of these pixels beyond half a quantum -, the small body 2e-3)
    assert share <= 3e-3, share


@pytest.mark.parametrize('leg', ['fixed_seed', 'fresh_seed'])
def test_random_epochs_body_sizes_and_spins_fuzz(engine, oracle, jupiter, saturn, leg):
    """
    What decides HOW the image kernels solve the light time - the epoch (ulp(et)), the size of the body (how far one
    quantum of its motion shows on it), its spin, its shape - swept at random: epochs 1995 ... 2040, equatorial radii
    300 ... 70 000 km, spheroids and triaxial bodies, spins from a tenth to fifty times the fixture's. Every case goes
    through the library's own choice of kernel and light-time path (closed form; stepping through the reference's iterates;
    its plain sequence; Newton step; first-order turn of a triaxial shape; general kernel), all 26 planes against the
    oracle: masks identical, values inside the bars but for the handful of pixels one epoch quantum away that two
    implementations of a rounded epoch cannot avoid (`_compare_allowing_epoch_quantum_flips`).
    """
    from conftest import fresh_seed
    from planetmapper_amd import _lib

    seed = 271828 if leg == 'fixed_seed' else fresh_seed('test_random_epochs_body_sizes_and_spins_fuzz')
    rng = np.random.default_rng(seed)
    seen = set()
    for i in range(14):
        base = saturn if i % 4 == 3 else jupiter
        et = float(rng.uniform(-1.6e8, 1.26e9))  # 1995 ... 2040
        shift = et - base.et
        a = float(10 ** rng.uniform(np.log10(300.0), np.log10(7e4)))
        c = a * float(rng.uniform(0.85, 1.0))
        b = a if i % 3 else a * float(rng.uniform(0.96, 0.9995))
        spin = base.wdot * float(10 ** rng.uniform(-1.0, 1.7))
        g = _with_its_own_sub_observer_point(_variant(base, et=et, ts0=base.ts0 + shift, radii=[a, b, c], wdot=spin,
                                                      diameter_arcsec=base.diameter_arcsec * a / base.radii[0]))
        nx, ny = int(rng.integers(150, 330)), int(rng.integers(150, 330))
        r0 = float(rng.uniform(0.25, 0.48) * min(nx, ny))
        x0, y0 = float(rng.uniform(0.4, 0.6) * nx), float(rng.uniform(0.4, 0.6) * ny)
        rot = float(rng.uniform(0, 2 * np.pi))
        d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
        d.rotation_rad = rot
        with _library_choice(engine):
            engine.set_geometry(g)
            engine.set_disc(x0, y0, r0, rot, nx, ny, True)
            out = engine.backplanes_img(oracle.PLANE_NAMES)
            seen.add((engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL), engine.get_option(_lib.PM_OPT_LAST_LT_PATH)))
        ref = oracle.backplanes_img(g, d, oracle.PLANE_NAMES)
        try:
            _compare_allowing_epoch_quantum_flips(out, ref, oracle.PLANE_NAMES, g, r0, label=f'fuzz {i}')
        except AssertionError as e:
            raise AssertionError(f'seed {seed} case {i}: et {et:.4g}, radii ({a:.5g}, {b:.5g}, {c:.5g}), spin x{spin / base.wdot:.3g}, '
                                 f'kernel / light-time path {sorted(seen)[-1]}: {e}') from e
    print(f'\n[seed {seed}] (kernel, light-time path) combinations met: {sorted(seen)}')
    assert len(seen) >= 3


def test_frame_that_spans_most_of_the_sky(engine, oracle, jupiter):
    """
    A plate scale of a degree per pixel: the frame spans 200 x 150 degrees, view angles pass 1.5 rad
    (`Params::view_direct` off: KM / ANGULAR planes through the reference's own RA / Dec round trip,
    `Body._obsvec2angular` body.py:1345) and leave every short sincos tier; the disc is a fraction of one pixel.
    All 26 planes against the oracle: the sky planes over the whole frame, the disc planes NaN but for the pixel
    or two that hold the planet.
    """
    nx, ny = 200, 150
    r0 = jupiter.diameter_arcsec / (2 * 3600.0)  # one degree per pixel
    x0, y0, rot = 97.3, 71.8, 0.4
    for opt in (True, False):
        engine.set_geometry(jupiter)
        engine.set_disc(x0, y0, r0, rot, nx, ny, opt)
        d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny, optimize_speed=opt)
        d.rotation_rad = rot
        out = engine.backplanes_img(oracle.PLANE_NAMES)
        ref = oracle.backplanes_img(jupiter, d, oracle.PLANE_NAMES)
        # (the bars of the km / arcsec planes are absolute, 1e-5 km and 3e-9 arcsec, set for fields the size of a disc: a
        #  coordinate of 1.3e9 km is not defined to 1e-5 km in binary64. Here they are relative to the coordinate -
        #  2e-12, i.e. a direction good to 3e-12 rad = 2e-10 deg; measured 6.5e-13 at 89 deg from the centre)
        wide = ('KM-X', 'KM-Y', 'ANGULAR-X', 'ANGULAR-Y')
        _compare(out, ref, [n for n in oracle.PLANE_NAMES if n not in wide], jupiter, r0=r0, flat=False)
        for n in wide:
            assert np.isfinite(out[n]).all()
            base = 1e-5 if n.startswith('KM') else 3e-9
            assert (np.abs(out[n] - ref[n]) <= base + 2e-12 * np.abs(ref[n])).all(), n
        assert np.isfinite(out['RA']).all() and np.ptp(out['ANGULAR-X']) > 150 * 3600.0


def test_observer_inside_the_body(engine, oracle):
    """
    surfpt_c's other branch: an observer INSIDE the ellipsoid (0.42 equatorial radii from the centre) sees the far
    intersection of every ray - no limb, emission angles beyond 90 deg. The reference cannot build such a Body (its
    angular diameter is arcsin of a number above 1), but the C ABI takes any block and the general kernel
    (k_disc_sph<FLAGS, 2>: the signed root) must do what CSPICE does: every pixel on the body, planes inside the bars
    against the oracle. The library's own dispatch sends it there (y2 <= 4).
    """
    from planetmapper_amd import _lib

    g = _near_field_geometry(30_000.0)
    g.diameter_arcsec = 3600.0 * 120  # (NaN from the provider: any positive number fixes the pixel scale)
    g.km_per_arcsec = 1.0
    nx, ny = 131, 97
    x0, y0, r0, rot = 60.0, 50.5, 45.0, 33.0
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC', 'LAT-CENTRIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'DISTANCE',
             'RADIAL-VELOCITY', 'DOPPLER']
    for opt in (False, True):
        engine.set_geometry(g)
        engine.set_disc(x0, y0, r0, float(np.deg2rad(rot)), nx, ny, opt)
        d = oracle.make_disc(x0, y0, r0, rot, nx, ny, optimize_speed=opt)
        d.rotation_rad = float(np.deg2rad(rot))
        out = engine.backplanes_img(names)
        assert engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == 3
        ref = oracle.backplanes_img(g, d, names)
        _compare(out, ref, names, g, flat=False)
        if not opt:
            assert np.isfinite(out['LON-GRAPHIC']).all() and np.nanmin(out['EMISSION']) > 90.0


@pytest.mark.parametrize('force_names', [None, ['LON-GRAPHIC', 'EMISSION', 'RING-RADIUS', 'RA', 'LIMB-DISTANCE', 'PIXEL-Y']])
def test_row_blocks_equal_the_full_frame(engine_fg, oracle, jupiter, saturn, force_names):
    """
    pm_backplanes_img_rows (the unit of row-block sharding over GPUs): any row block is
    bit-identical to the same rows of the full-frame launch, for the spheroid fast path, the
    ring path, the sky planes and ragged blocks; invalid windows are rejected.
    """
    names = force_names or oracle.PLANE_NAMES
    for g, (nx, ny) in ((jupiter, (203, 157)), (saturn, (130, 95))):
        engine_fg.set_geometry(g)
        engine_fg.set_disc(nx / 2.1, ny / 1.9, 0.3 * nx, 0.7, nx, ny, True)
        full = engine_fg.backplanes_img(names)
        for a, n in ((0, ny), (0, 1), (ny - 1, 1), (ny // 4, ny // 2 + 1), (5, ny - 6), (ny, 0)):
            block = engine_fg.backplanes_img_rows(names, a, n)
            for k in names:
                assert block[k].shape == (n, nx)
                assert np.array_equal(block[k], full[k][a : a + n], equal_nan=True), (k, a, n)
        for a, n in ((-1, 3), (ny - 2, 3), (0, -1)):
            with pytest.raises(ValueError):
                engine_fg.backplanes_img_rows(names, a, n)
    import torch

    dev = {k: torch.empty((64, 130), dtype=torch.float64, device='cuda') for k in names}
    engine_fg.backplanes_img_rows_device(dev, 20, 64)
    engine_fg.synchronize()
    for k in names:
        assert np.array_equal(dev[k].cpu().numpy(), full[k][20:84], equal_nan=True), k


def test_smoothing_splines_vs_oracle_kats_and_golden(engine, oracle, jupiter):
    """
    `spline_smoothing > 0` (FITPACK regrid smoothing, body_xy.py:1673-1680): the reference's
    own expected values (tests/test_body_xy.py:1194-1231), its golden FITS
    map_rectangular-interpolation ((1, 3), s = 2.34; planes 6 and 7 loosely, as the reference
    compares them) and the oracle - whose knots and coefficients equal scipy's - on a larger
    cube: the fits run on the GPU (corrected semi-normal equations against the host's band QR),
    so agreement is to 1e-7 of the data scale rather than to the last digits.
    """
    from planetmapper_amd import BodyXY, Observation
    from test_api_host import IMAGE, _smoothing_kats

    body = BodyXY('Jupiter', geometry=jupiter, engine=engine)
    body.set_img_size(6, 5)
    body.set_disc_params(2.75, 1.3, 2.3, 45.678)
    for sm, exp in _smoothing_kats():
        got = body.map_img(IMAGE, interpolation='linear', degree_interval=45, spline_smoothing=sm)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-8, equal_nan=True), sm
    factors = [1, 2.345, -10, 3456.789, np.nan]
    kwargs = dict(interpolation='cubic', degree_interval=45, spline_smoothing=1)
    mapped = body.map_img([IMAGE * f for f in factors], **kwargs)
    for f, m in zip(factors, mapped):
        assert np.allclose(m, body.map_img(IMAGE * f, **kwargs), rtol=1e-9, atol=1e-9, equal_nan=True), f

    cube = np.load(os.path.join(GOLDEN, 'input_cube.npz'))['data']
    obs = Observation(data=cube, geometry=jupiter, engine=engine)
    obs.set_disc_params(2.5, 3.1, 3.9, 123.456)
    gold = np.load(os.path.join(GOLDEN, 'golden_map_rectangular_interpolation.npz'))['PRIMARY']
    m = obs.get_mapped_data(interpolation=(1, 3), spline_smoothing=2.34, degree_interval=30)
    assert np.array_equal(np.isnan(m), np.isnan(gold))
    for pl in range(10):
        rtol, atol = {6: (1e-1, 1e-1), 7: (10, 1)}.get(pl, (1e-6, 1e-5))
        assert np.allclose(m[pl], gold[pl], rtol=rtol, atol=atol, equal_nan=True), pl

    sz = 150
    x0 = y0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, 0.7 * x0, 0.2, sz + 13, sz, True)
    d = oracle.make_disc(x0, y0, 0.7 * x0, 0.0, sz + 13, sz)
    d.rotation_rad = 0.2
    lon, lat = oracle.rectangular_grid(jupiter, 3.0)
    xm, ym = oracle.xy_map(jupiter, d, lon, lat)
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:sz, 0 : sz + 13]
    base = np.sin(xx / 9.0) * np.cos(yy / 13.0) * 4
    cube = base[None] + rng.standard_normal((4, sz, sz + 13))
    cube[0][rng.random((sz, sz + 13)) < 0.01] = np.nan
    cube[1][40:60, 50:90] = np.nan
    cube[2][:] = np.nan
    npx = sz * (sz + 13)
    # (s is kept at or above the noise level n_pixels * sigma^2: far below it FITPACK itself runs
    #  into numerically singular knot sets and scipy returns coefficients of 1e120)
    for interp, s in (('linear', 0.9 * npx), ('cubic', 1.0 * npx), ('cubic', 0.93 * npx), ((2, 3), 1.05 * npx), ('quadratic', 30.0 * npx), (5, 1e3 * npx)):
        for prop in (True, False):
            a = engine.map_cube(cube, xm, ym, interp, prop, spline_smoothing=s)
            b = oracle.map_cube(cube, xm, ym, (interp, interp) if isinstance(interp, int) else interp, prop,
                                spline_smoothing=s)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (interp, s, prop)
            fin = np.isfinite(b)
            assert fin.sum() > 1000
            assert np.max(np.abs(a[fin] - b[fin])) <= 1e-7 * max(1.0, np.abs(b[fin]).max()), (interp, s, prop)
    a = engine.map_cube(cube[0].astype(np.float32), xm, ym, 'cubic', True, spline_smoothing=npx)
    b = oracle.map_cube(cube[0].astype(np.float32), xm, ym, 'cubic', True, spline_smoothing=npx)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) <= 1e-6
    with pytest.raises(ValueError):
        engine.map_cube(cube, xm, ym, 'cubic', True, spline_smoothing=-1.0)
    # s = 0 afterwards is the interpolating spline again
    assert np.array_equal(engine.map_cube(cube[:1], xm, ym, 'cubic', True),
                          engine.map_cube(cube[:1], xm, ym, 'cubic', True, spline_smoothing=0.0), equal_nan=True)


def test_config4_saturn_rings_full_size(engine_fg, oracle, saturn):
    """
    BASELINE config 4 at its full size: Saturn + rings, 4096^2, r0 = 800 px, rotation 20 deg,
    8 planes. NaN masks (disc limb, ring-plane hits, rings hidden behind the disc) bit-exact and
    the conditioned tolerances against the OpenMP oracle ("parity unpinned" by reference goldens).
    """
    sz = 4096
    x0 = y0 = (sz - 1) / 2
    engine_fg.set_geometry(saturn)
    engine_fg.set_disc(x0, y0, 800.0, float(np.deg2rad(20.0)), sz, sz, True)
    names = HEADLINE + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
    out = engine_fg.backplanes_img(names)
    oracle.set_num_threads(16)
    ref = oracle.backplanes_img(saturn, oracle.make_disc(x0, y0, 800.0, 20.0, sz, sz), names)
    for n in names:
        assert np.array_equal(np.isnan(out[n]), np.isnan(ref[n])), n
    stats = _compare(out, ref, names, saturn, r0=800.0)
    from parity import base_deg

    # (max |diff|, share inside the conditioned suite's flat bar for THIS geometry - base_deg(saturn) = 1.53e-9 deg: 12
    #  half-ulps of the unit ray at 1.2e9 km -, share inside the north star's own flat 1e-9 deg)
    print(f'\nconfig 4 (4096^2 Saturn + rings) HIP vs oracle; flat bar of the suite for this geometry {base_deg(saturn):.3e} deg '
          '(not 1e-9), third figure = share inside the north star\'s flat 1e-9 deg:', stats)
    # measured (fast / general kernel): LON 96.79 / 96.42 %, LAT 99.58 / 99.46 %, PHASE 100 %, INC and EMI 98.87 / 98.62 %
    # of the 1.84 M on-disc pixels inside 1e-9 deg; asserted at that - 0.3 %
    floor_1e9 = {'LON-GRAPHIC': 0.961, 'LAT-GRAPHIC': 0.9915, 'PHASE': 1.0, 'INCIDENCE': 0.983, 'EMISSION': 0.983}
    for n, f in floor_1e9.items():
        assert stats[n][2] >= f, (n, stats[n], f)
    assert 0.1 < np.isfinite(out['LON-GRAPHIC']).mean() < 0.14  # pi * 800^2 * (1 - f) / 4096^2
    assert np.isfinite(out['RING-RADIUS']).mean() > 0.5


@pytest.mark.parametrize('case', ['edge_on', 'observer_in_ring_plane', 'disc_partly_off_frame'])
def test_saturn_rings_special_geometries_full_size(engine_fg, oracle, saturn, case):
    """
    The divergent paths of config 4 at 4096^2: rings seen (nearly) edge-on - the ring plane's horizon crosses
    the frame, most rays meet the plane at grazing angles or not at all - the observer IN the ring plane
    (plane constant 0: every ray that is not parallel to the plane 'hits' it at distance 0, CSPICE's inrypl_c
    rule), and a disc half out of the frame. Masks bit-exact, values inside the bars.
    """
    from planetmapper_amd.geometry import PMGeometry  # noqa: F401

    sz = 4096
    g = saturn.copy()
    x0 = y0 = (sz - 1) / 2
    r0 = 800.0
    if case != 'disc_partly_off_frame':
        # tilt the ring plane until the observer is `b` above it: rotate its normal about the axis
        # perpendicular to the normal and the line of sight
        n = np.array(g.ring_n[:])
        t0 = np.array(g.T0[:])
        los = t0 / np.linalg.norm(t0)
        axis = np.cross(n, los)
        axis /= np.linalg.norm(axis)
        b_now = np.arcsin(float(n @ los))
        b_new = 0.0 if case == 'observer_in_ring_plane' else np.deg2rad(0.02)
        ang = b_now - b_new
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        n2 = Rm @ n
        if case == 'observer_in_ring_plane':
            n2 = n2 - (n2 @ los) * los  # exactly perpendicular to the line of sight: plane constant 0
            n2 /= np.linalg.norm(n2)
        k2 = float(n2 @ t0)
        if k2 < 0:
            n2, k2 = -n2, -k2
        for i in range(3):
            g.ring_n[i] = n2[i]
        g.ring_k = 0.0 if case == 'observer_in_ring_plane' else k2
    else:
        x0, y0 = 150.0, sz - 300.0
    engine_fg.set_geometry(g)
    engine_fg.set_disc(x0, y0, r0, float(np.deg2rad(20.0)), sz, sz, True)
    names = HEADLINE + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
    out = engine_fg.backplanes_img(names)
    oracle.set_num_threads(16)
    ref = oracle.backplanes_img(g, oracle.make_disc(x0, y0, r0, 20.0, sz, sz), names)
    for n in names:
        assert np.array_equal(np.isnan(out[n]), np.isnan(ref[n])), (case, n)
    _compare(out, ref, names, g, r0=r0)
    frac_ring = np.isfinite(out['RING-RADIUS']).mean()
    if case == 'edge_on':
        assert 0.05 < frac_ring < 0.97  # both sides of the plane's horizon are in the frame
    elif case == 'observer_in_ring_plane':
        assert np.nanmax(np.abs(out['RING-DISTANCE'])) == 0.0 or frac_ring == 0.0
    else:
        assert 0.02 < np.isfinite(out['LON-GRAPHIC']).mean() < 0.08


def test_config3_cube_2048_full_size(engine, oracle, jupiter):
    """
    BASELINE config 3 at its full size (SURVEY 8d recipe): P = 8 planes of 2048^2 f64 - limb-darkened
    disc + 0.05 sigma noise from default_rng(20050101), 0.1 % NaN per plane, one full NaN row in
    plane 3 - disc (1023.5, 1023.5, 921.15), 1 deg map, bilinear, propagate_nan. x/y maps against
    the oracle, mapped cube against the oracle on the GPU's own maps to 1e-12.
    """
    sz, P = 2048, 8
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    rng = np.random.default_rng(20050101)
    yy, xx = np.mgrid[0:sz, 0:sz]
    mu = np.sqrt(np.clip(1 - ((xx - x0) ** 2 + (yy - y0) ** 2) / r0**2, 0, None))
    cube = mu[None] + 0.05 * rng.standard_normal((P, sz, sz))
    cube[rng.random((P, sz, sz)) < 1e-3] = np.nan
    cube[3, 1000, :] = np.nan
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, r0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 1.0)
    assert lon.shape == (180, 360) and lon[0, 0] == 359.5 and lon[0, -1] == 0.5
    xm, ym = engine.xy_map(lon, lat)
    d = oracle.make_disc(x0, y0, r0, 0.0, sz, sz)
    ox, oy = oracle.xy_map(jupiter, d, lon, lat)
    assert np.array_equal(np.isnan(xm), np.isnan(ox))
    # the 1e-9 deg bar is 1e-9 * 3600 / plate scale = 1.8e-4 px here (x 1 / cos(emission) towards
    # the limb); measured: 100x below it at the limb, 1e-9 px in the median
    assert np.nanmax(np.abs(xm - ox)) < 2e-6 and np.nanmax(np.abs(ym - oy)) < 2e-6
    assert np.nanmedian(np.abs(xm - ox)) < 1e-8
    got = engine.map_cube(cube, xm, ym, 'linear', True)
    ref = oracle.map_cube(cube, xm, ym, 'linear', True)
    assert got.shape == (P, 180, 360)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) <= 1e-12
    assert 0.3 < np.isfinite(got[0]).mean() < 0.5 and np.isfinite(got[3]).sum() < np.isfinite(got[2]).sum()


def test_config5_cube_512_planes_properties(engine, oracle, jupiter):
    """
    BASELINE config 5 at its full size, device-resident: 512 planes of 1024^2 f64 (4 GiB in HBM)
    mapped in one call. Size-independent properties: (1) linearity - plane p holds
    a_p * base + b_p, so on every finite cell mapped[p] = a_p * mapped_base + b_p; (2) NaN masks of
    all planes equal the base plane's; (3) planes chosen at random equal the oracle's map of
    that plane bit-for-bit-close (1e-12); (4) a checksum over planes is invariant to the order in
    which they are mapped (plane blocks as the multi-GPU shards would see them).
    """
    import torch

    sz, P = 1024, 512
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    engine.set_geometry(jupiter)
    engine.set_disc(x0, y0, r0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 1.0)
    xm, ym = engine.xy_map(lon, lat)
    gen = torch.Generator(device='cuda').manual_seed(5)
    base = torch.randn((sz, sz), generator=gen, device='cuda', dtype=torch.float64)
    base[torch.rand((sz, sz), generator=gen, device='cuda') < 1e-3] = float('nan')
    a = torch.linspace(-3.0, 3.0, P, device='cuda', dtype=torch.float64)
    a[a.abs() < 0.01] = 0.5
    b = torch.linspace(100.0, -100.0, P, device='cuda', dtype=torch.float64)
    cube = a[:, None, None] * base[None] + b[:, None, None]  # 4 GiB
    dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    out = torch.empty((P, 180, 360), dtype=torch.float64, device='cuda')
    engine.map_cube_device(cube, np.float64, P, dx, dy, 180, 360, out)
    engine.synchronize()
    mb = torch.empty((1, 180, 360), dtype=torch.float64, device='cuda')
    engine.map_cube_device(base, np.float64, 1, dx, dy, 180, 360, mb)
    engine.synchronize()
    fin = torch.isfinite(mb[0])
    assert 20000 < int(fin.sum()) < 32400
    assert bool((torch.isfinite(out) == fin[None]).all())  # (2)
    lin = a[:, None, None] * mb + b[:, None, None]
    err = (out - lin)[:, fin].abs().max()
    assert float(err) < 1e-11, float(err)  # (1): bilinear weights sum to 1 within rounding
    host_base = base.cpu().numpy()
    for p in (0, 77, 300, 511):  # (3)
        plane = float(a[p]) * host_base + float(b[p])
        ref = oracle.map_cube(plane, xm, ym, 'linear', True)[0]
        got = out[p].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.nanmax(np.abs(got - ref)) <= 1e-11, p
    out2 = torch.empty_like(out)  # (4): 8 blocks of 64 planes, last block first
    for blk in reversed(range(8)):
        s = slice(64 * blk, 64 * blk + 64)
        engine.map_cube_device(cube[s], np.float64, 64, dx, dy, 180, 360, out2[s])
    engine.synchronize()
    assert torch.equal(torch.nan_to_num(out2), torch.nan_to_num(out))


@pytest.mark.parametrize('leg', ['fixed_seed', 'fresh_seed'])
def test_random_discs_and_frames_fuzz(engine_fg, oracle, jupiter, saturn, leg):
    """
    Sweep over frame shapes (1 x 1 up to ragged 200-pixel sides), disc positions inside,
    on the edge of and outside the frame, radii from sub-pixel to frame-filling, any rotation,
    with and without the radius pre-mask and altitude offsets: all 26 planes, masks bit-exact.
    Once with a fixed seed, once with a seed of this run (logged: conftest.fresh_seed).
    """
    from conftest import fresh_seed

    seed = 20260101 if leg == 'fixed_seed' else fresh_seed('test_random_discs_and_frames_fuzz')
    rng = np.random.default_rng(seed)
    cases = [(1, 1, 0.0, 0.0, 0.6, 0.0), (2, 1, 0.5, 0.0, 5.0, 1.0), (64, 1, 31.5, 0.0, 40.0, 3.0), (1, 70, 0.0, 30.0, 25.0, 5.9)]
    for _ in range(26):
        nx, ny = int(rng.integers(3, 200)), int(rng.integers(3, 200))
        r0 = float(10 ** rng.uniform(-0.3, 2.4))
        x0, y0 = float(rng.uniform(-0.3 * nx, 1.3 * nx)), float(rng.uniform(-0.3 * ny, 1.3 * ny))
        cases.append((nx, ny, x0, y0, r0, float(rng.uniform(0, 2 * np.pi))))
    for i, (nx, ny, x0, y0, r0, rot) in enumerate(cases):
        g = saturn if i % 5 == 4 else jupiter
        opt = bool(i % 3)
        alt = [0.0, 0.0, 1500.0, -300.0][i % 4]
        engine_fg.set_geometry(g)
        engine_fg.set_disc(x0, y0, r0, rot, nx, ny, opt)
        d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny, optimize_speed=opt)
        d.rotation_rad = rot
        out = engine_fg.backplanes_img(oracle.PLANE_NAMES, alt=alt)
        ref = oracle.backplanes_img(g, d, oracle.PLANE_NAMES, alt=alt)
        for n in oracle.PLANE_NAMES:
            assert masks_agree(n, out[n], ref[n]), (seed, i, n, nx, ny, x0, y0, r0, rot)
        try:
            _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0, flat=False)
        except AssertionError as e:
            raise AssertionError(f'seed {seed} case {i} {(nx, ny, x0, y0, r0, rot, opt, alt)}: {e}') from e


@pytest.mark.parametrize('leg', ['fixed_seed', 'fresh_seed'])
def test_random_reprojection_fuzz(engine, oracle, jupiter, leg):
    """
    Sweep over small frames, discs, NaN / inf patterns, dtypes and every interpolation mode
    of map_img (both NaN policies), on rectangular maps of random resolution: mapped planes against
    the oracle on the GPU's own x/y maps (NaN masks identical, values to 1e-9 of the data scale).
    Once with a fixed seed, once with a seed of this run (logged: conftest.fresh_seed).
    """
    from conftest import fresh_seed

    seed = 77 if leg == 'fixed_seed' else fresh_seed('test_random_reprojection_fuzz')
    rng = np.random.default_rng(seed)
    modes = ['nearest', 'linear', 'quadratic', 'cubic', (1, 3), (4, 2), 'smooth']
    for i in range(14):
        nx, ny = int(rng.integers(7, 60)), int(rng.integers(7, 60))
        r0 = float(rng.uniform(0.2, 0.7) * min(nx, ny))
        x0, y0 = float(rng.uniform(0.3, 0.7) * nx), float(rng.uniform(0.3, 0.7) * ny)
        engine.set_geometry(jupiter)
        engine.set_disc(x0, y0, r0, float(rng.uniform(0, 6.28)), nx, ny, True)
        lon, lat = oracle.rectangular_grid(jupiter, float(rng.choice([5.0, 9.0, 15.0, 30.0])))
        xm, ym = engine.xy_map(lon, lat)
        if not np.isfinite(xm).any():
            continue
        cube = rng.standard_normal((3, ny, nx)) * 10
        cube[0][rng.random((ny, nx)) < 0.05] = np.nan
        cube[1][rng.random((ny, nx)) < 0.03] = np.inf
        cube[2][ny // 3 : ny // 3 + 2, :] = np.nan
        if i % 4 == 1:
            cube = np.nan_to_num(cube, posinf=0).astype(np.float32)
        elif i % 4 == 2:
            cube = np.nan_to_num(cube, posinf=0).astype(np.int16)
        for interp in modes:
            for prop in (True, False):
                a = engine.map_cube(cube, xm, ym, interp, prop)
                b = oracle.map_cube(cube, xm, ym, interp, prop)
                assert np.array_equal(np.isnan(a), np.isnan(b)), (seed, i, nx, ny, interp, prop)
                fin = np.isfinite(b)
                if fin.any():
                    scale = max(1.0, float(np.abs(b[fin]).max()))
                    assert np.max(np.abs(a[fin] - b[fin])) <= 1e-9 * scale, (seed, i, nx, ny, interp, prop)


def test_transforms_with_non_finite_inputs(engine, oracle, jupiter):
    """Non-finite inputs short-circuit to NaN like the reference (body.py:1023-1029, 1918-1924):
    NaN / +-inf / 1e300 in either coordinate, every pair of coordinate systems."""
    nx, ny = 120, 90
    engine.set_geometry(jupiter)
    engine.set_disc(60.0, 45.0, 30.0, 0.4, nx, ny, True)
    d = oracle.make_disc(60.0, 45.0, 30.0, 0.0, nx, ny)
    d.rotation_rad = 0.4
    bad = [np.nan, np.inf, -np.inf, 1e300, -1e300]
    base = {'xy': (60.0, 45.0), 'radec': oracle.transform(jupiter, d, 'xy', 'radec', [60.0], [45.0]),
            'angular': (1.0, -2.0), 'km': (1000.0, -2000.0), 'lonlat': (150.0, -3.0)}  # fmt: skip
    for src, (u0, v0) in base.items():
        a = np.array([float(np.ravel(u0)[0])] + bad + [float(np.ravel(u0)[0])] * len(bad))
        b = np.array([float(np.ravel(v0)[0])] + [float(np.ravel(v0)[0])] * len(bad) + bad)
        for dst in base:
            got = engine.transform(src, dst, a, b)
            ref = oracle.transform(jupiter, d, src, dst, a, b)
            assert np.array_equal(np.isnan(got[0]), np.isnan(ref[0])), (src, dst, got[0], ref[0])
            assert np.array_equal(np.isnan(got[1]), np.isnan(ref[1])), (src, dst)
            assert np.isfinite(got[0][0]) or src == dst or np.isnan(ref[0][0])


def test_point_functions_on_gpu(jupiter):
    """
    Body.*_from_lonlat on the real engine against the reference's value tables
    (tests/test_body.py:1826-1835, 1866-1868, 1901-1905, 1733-1737, 1764-1771, 1980-1983,
    2487-2489, 2522-2524) - the same tables tests/test_api_host.py runs on the oracle.
    """
    from planetmapper_amd import BodyXY

    body = BodyXY('Jupiter', '2005-01-01T00:00:00', observer='HST', geometry=jupiter)
    nan = np.nan
    close = lambda a, b: np.allclose(a, b, equal_nan=True)  # noqa: E731
    assert close(body.illumination_angles_from_lonlat(0, 0), (10.31594976458697, 163.2795134457034, 152.99822832991876))
    assert close(
        body.illumination_angles_from_lonlat(123.456, -78.9), (10.316968817304499, 79.16351827229181, 77.68583738495468)
    )
    assert close(body.illumination_angles_from_lonlat(nan, 0), (nan, nan, nan))
    c = body.graphic2centric_lonlat(123.456, -78.9)
    assert close(
        body.illumination_angles_from_lonlat(*c, planetocentric=True),
        (10.316968817304499, 79.16351827229181, 77.68583738495468),
    )
    assert close(body.azimuth_angle_from_lonlat(0, 0), 177.66817822757469)
    assert close(body.azimuth_angle_from_lonlat(123.456, -78.9), 169.57651996164563)
    assert close(body.radial_velocity_from_lonlat(np.array([0.0, 45.0]), np.array([0.0, 45.0])), (-20.796924908179438, -17.75706386255955))
    assert close(body.distance_from_lonlat(45, 45), 819656453.7301536)
    for lon, expected, s in [(0, 22.89638888888889, '22:53:47'), (999.999, 4.229722222222223, '04:13:47'), (nan, nan, '')]:
        assert np.isclose(body.local_solar_time_from_lon(lon), expected, equal_nan=True)
        assert body.local_solar_time_string_from_lon(lon) == s
    assert [body.test_if_lonlat_visible(*ll) for ll in [(0, 0), (180, 12), (50, -80), (nan, 0)]] == [False, True, True, False]
    assert [body.test_if_lonlat_visible(lo, la, alt=al) for lo, la, al in [(0, 0, 1e6), (153.1, -3.0, -1), (153.1, -3.0, 1)]] == [True, False, True]
    assert [body.test_if_lonlat_illuminated(*ll) for ll in [(0, 0), (180, 12), (50, -80), (np.inf, np.inf)]] == [False, True, False, False]


def test_radec_query_vs_oracle_and_kats(engine, oracle, jupiter, saturn):
    """
    pm_radec_query (Body.ring_plane_coordinates / limb_coordinates_from_radec / radec2lonlat at
    sky points): the reference's value tables (tests/test_body.py:2008-2049, 1683-1730) and the
    oracle on a grid of sky points around the body, both ring visibility rules, with altitude.
    """
    engine.set_geometry(jupiter)
    engine.set_disc(2.5, 3.1, 3.9, 0.0, 7, 10, True)
    q = engine.radec_query([196.37347182693253, 196.3, 196.37198562427025, np.nan], [-5.561472466522512, -5.5, -5.565793847134351, 0.0])
    assert np.allclose(q[2:5, 0], (1377914.753652832, 152.91772706249577, 818261707.8278764))
    assert np.allclose(q[2:5, 1], (9305877.091704229, 145.3644753085151, 810435703.2382222))
    assert np.isnan(q[2:5, 2:]).all() and np.isnan(q[:, 3]).all()
    q = engine.radec_query(196.37198562427025, -5.565793847134351, ring_only_visible=False)
    assert q.shape == (8,) and np.allclose(q[2:5], (4638.105239104683, 156.0690984698183, 819638074.3312378))
    q = engine.radec_query([0, 196.3719829300016, 196.372, 196.3], [0, -5.565779946690757, -5.566, -5.5])
    exp = [
        (82.72145635455739, -7.331180721378409, 243226446.365406),
        (67.23274105785333, 58.34599234749429, -68089.8880967631),
        (248.13985326986065, -64.83923990338549, -64857.80811442864),
        (64.1290135632679, 20.79992677586983, 1320579.9259661217),
    ]
    assert np.allclose(q[5:8].T, exp, rtol=1e-5)
    rng = np.random.default_rng(7)
    for g in (jupiter, saturn):
        engine.set_geometry(g)
        engine.set_disc(2.5, 3.1, 3.9, 0.0, 7, 10, True)
        t0 = np.array(g.T0[:])  # observer -> target centre, J2000
        ra0 = np.rad2deg(np.arctan2(t0[1], t0[0])) % 360.0
        dec0 = np.rad2deg(np.arcsin(t0[2] / np.linalg.norm(t0)))
        span = 1.2 * g.diameter_arcsec / 3600.0
        ra = ra0 + rng.uniform(-span, span, 4000) / np.cos(np.deg2rad(dec0))
        dec = dec0 + rng.uniform(-span, span, 4000)
        ra[::97] = np.nan
        dec[::89] = np.inf
        # both evaluations: the B0 kernel (the library's choice for a spheroid seen from outside) and the J2000 one behind
        # PM_OPT_GENERAL_KERNEL, which every other body takes. Bars at the measured deviations x 3 (profiles/r05_radec_query_deviation.txt):
        # angles of the intercept and the limb point are conditioned like the image planes' (1 / cos e at the limb)
        from planetmapper_amd import _lib

        for general in (0, 1):
            engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, general)
            try:
                for alt, vis in ((0.0, True), (0.0, False), (2500.0, True)):
                    got = engine.radec_query(ra, dec, alt=alt, ring_only_visible=vis)
                    ref = oracle.radec_query(g, ra, dec, alt=alt, ring_only_visible=vis).T
                    assert np.array_equal(np.isnan(got), np.isnan(ref)), (general, alt, vis)
                    fin = np.isfinite(ref)
                    d = np.abs(got - ref)
                    d[[0, 3, 5]] = np.minimum(d[[0, 3, 5]], 360.0 - d[[0, 3, 5]])
                    scale = np.array([2e-7, 1e-7, 5e-6, 2e-9, 5e-6, 2e-7, 1e-7, 5e-6])[:, None]  # deg / km
                    assert np.all(d[fin] <= np.broadcast_to(scale, d.shape)[fin]), (general, alt, vis, np.nanmax(d / scale, axis=1))
                    for k in (0, 1, 5, 6):
                        assert np.mean(d[k][fin[k]] < 3e-9) > 0.97, (general, alt, vis, k)
                    assert fin[0].sum() > 100 and fin[2].sum() > 1000
            finally:
                engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)


@pytest.mark.parametrize('leg', ['fixed_seed', 'fresh_seed'])
def test_random_geometries_fuzz(engine, oracle, leg):
    """
    (Once with a fixed seed, once with a seed of this run - logged: conftest.fresh_seed.)
    Sweep over observers: distances from 2.6 radii to 30 au (across the thresholds at
    which the launcher leaves the spheroid fast path: observer within two radii of the surface,
    spin angle over a light-time span, acceleration term), any aspect angle, observer velocities
    up to 60 km/s, oblate / nearly spherical / triaxial shapes, both longitude conventions,
    altitude offsets: all 26 image planes and the map chain against the oracle, masks bit-exact.
    Geometry blocks from the package's own host provider ("parity unpinned" by goldens).
    """
    from planetmapper_amd.ephem import Ephemeris, RotationModel
    from planetmapper_amd.geometry import CLIGHT, GeometryBuilder
    from planetmapper_amd.scenarios import _load_json

    d = _load_json('jupiter_hst_2005')
    gb = GeometryBuilder(Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck']), d['target_id'])
    h = d['header']
    from conftest import fresh_seed

    seed = 314159 if leg == 'fixed_seed' else fresh_seed('test_random_geometries_fuzz')
    rng = np.random.default_rng(seed)
    r_eq = 71492.0
    dists = [2.6 * r_eq, 2.95 * r_eq, 3.05 * r_eq, 5.0 * r_eq] + list(10 ** rng.uniform(5.6, 9.65, 16))
    for i, dist in enumerate(dists):
        g = gb.build(
            d['et'] + float(rng.uniform(-2.0, 4.0)) * 3600.0,
            observer_velocity=list(rng.uniform(-1, 1, 3) * rng.uniform(0, 60)),
            target_ra_dec_dist_lt=(
                (h['PLANMAP TARGET RA'] + float(rng.uniform(-180, 180))) % 360.0,
                float(np.clip(h['PLANMAP TARGET DEC'] + rng.uniform(-60, 60), -85, 85)),
                float(dist),
                float(dist) / CLIGHT,
            ),
        )
        shape = i % 4
        if shape == 1:
            g = _variant(g, radii=[71492.0, 71492.0, 71400.0])  # nearly spherical
        elif shape == 2:
            g = _variant(g, radii=[71492.0, 70300.0, 66854.0])  # triaxial
        elif shape == 3:
            g = _variant(g, radii=[71492.0, 71492.0, 57000.0], west_positive=0)  # very oblate, east-positive
        nx, ny = int(rng.integers(90, 170)), int(rng.integers(90, 170))
        r0 = float(rng.uniform(0.2, 0.6) * min(nx, ny))
        x0, y0 = float(rng.uniform(0.3, 0.7) * nx), float(rng.uniform(0.3, 0.7) * ny)
        if dist < 2e6:
            # a disc of tens of degrees: keep the frame within ~30 deg of the target direction (the
            # sky-plane tolerances of tests/parity.py are absolute and meant for tangent-plane fields)
            r0 = float(rng.uniform(0.6, 0.9) * min(nx, ny))
            x0, y0 = float(rng.uniform(0.4, 0.6) * nx), float(rng.uniform(0.4, 0.6) * ny)
        rot = float(rng.uniform(0, 2 * np.pi))
        alt = [0.0, 0.0, 800.0][i % 3]
        engine.set_geometry(g)
        engine.set_disc(x0, y0, r0, rot, nx, ny, True)
        dd = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
        dd.rotation_rad = rot
        out = engine.backplanes_img(oracle.PLANE_NAMES, alt=alt)
        ref = oracle.backplanes_img(g, dd, oracle.PLANE_NAMES, alt=alt)
        for n in oracle.PLANE_NAMES:
            assert masks_agree(n, out[n], ref[n]), (seed, i, n, dist)
        try:
            _compare(out, ref, oracle.PLANE_NAMES, g, r0=r0, flat=False)
        except AssertionError as e:
            raise AssertionError(f'seed {seed} case {i}: distance {dist:.4g} km, shape {shape}, alt {alt}: {e}') from e
        assert np.isfinite(out['LON-GRAPHIC']).sum() > 500, (i, dist)
        lon, lat = oracle.rectangular_grid(g, 10.0)
        om = engine.backplanes_map(oracle.PLANE_NAMES, lon, lat, alt=alt)
        rm = oracle.backplanes_map(g, dd, oracle.PLANE_NAMES, lon, lat, alt=alt)
        try:
            _compare(om, rm, oracle.PLANE_NAMES, g, r0=r0, flat=False)
        except AssertionError as e:
            raise AssertionError(f'seed {seed} case {i} (map): distance {dist:.4g} km, shape {shape}, alt {alt}: {e}') from e


def test_mem_argument_is_validated_by_every_entry_point(engine, jupiter):
    """
    PM_MEM_HOST_CUBE is meaningful for cube mapping only (the cube is the one host buffer); every
    other entry point must refuse it - and any unknown value - instead of treating it as PM_MEM_HOST
    (which would hand device pointers to the CPU copy threads).
    """
    import ctypes

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import plane_mask

    engine.set_geometry(jupiter)
    engine.set_disc(3.5, 3.5, 3.0, 0.0, 8, 8, True)
    lib, ctx = engine._lib, engine._ctx
    buf = np.zeros((4, 64))
    ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
    ptrs[0] = buf[0].ctypes.data
    p = lambda k: buf[k].ctypes.data  # noqa: E731
    for bad in (_lib.PM_MEM_HOST_CUBE, 7, -1):
        assert lib.pm_backplanes_img(ctx, plane_mask(['LON-GRAPHIC']), 0.0, ptrs, bad) == _lib.PM_ERR_INVALID_ARGUMENT
        assert lib.pm_backplanes_img_rows(ctx, plane_mask(['LON-GRAPHIC']), 0.0, 0, 8, ptrs, bad) == _lib.PM_ERR_INVALID_ARGUMENT
        assert lib.pm_backplanes_map(ctx, plane_mask(['LON-GRAPHIC']), p(1), p(2), 8, 8, 0.0, ptrs, bad) == _lib.PM_ERR_INVALID_ARGUMENT
        assert lib.pm_xy_map(ctx, p(1), p(2), 8, 8, 0.0, p(0), p(3), bad) == _lib.PM_ERR_INVALID_ARGUMENT
        assert lib.pm_transform(ctx, 0, 1, 64, p(1), p(2), 0.0, 0, p(0), p(3), bad) == _lib.PM_ERR_INVALID_ARGUMENT
        assert lib.pm_radec_query(ctx, 8, p(1), p(2), 0.0, 1, p(0), bad) == _lib.PM_ERR_INVALID_ARGUMENT
        with pytest.raises(ValueError):
            engine._check(lib.pm_transform(ctx, 0, 1, 64, p(1), p(2), 0.0, 0, p(0), p(3), bad))
    for bad in (7, -1):
        assert lib.pm_map_cube(ctx, p(0), 0, 1, p(1), p(2), 8, 8, 1, 1, p(3), bad) == _lib.PM_ERR_INVALID_ARGUMENT
    # the engine still works after the refusals
    assert np.isfinite(engine.backplanes_img(['LON-GRAPHIC'])['LON-GRAPHIC']).any()


def test_a_point_alone_equals_the_same_point_inside_an_array(engine, oracle, jupiter):
    """
    Wave votes in the kernels (`__all` / `__any`) may skip work but never decide a lane's result in
    the vsep / latitude helpers: the incidence / emission / latitude of one map point must come out
    bit-identical whether it is evaluated alone, in a full wave of neighbours, or next to points on
    the other side of the body (other tiers of the elementary functions are wave-uniform by design and
    are held to 2e-13 deg here: pm_fastmath.hip.h sincos_auto).
    """
    engine.set_geometry(jupiter)
    engine.set_disc(140.5, 90.25, 80.0, 0.3, 300, 200, True)
    rng = np.random.default_rng(12345)
    names = ['LAT-GRAPHIC', 'INCIDENCE', 'EMISSION', 'PHASE', 'PIXEL-X', 'PIXEL-Y']
    lon = rng.uniform(0, 360, 512)
    lat = rng.uniform(-90, 90, 512)
    # a few points hard against the poles / the sub-observer point: the branch lanes
    lat[:8] = [89.9999, -89.9999, 89.5, -89.5, 60.0, -60.0, 30.0, -30.0]
    whole = engine.backplanes_map(names, lon[None, :], lat[None, :])
    order = rng.permutation(512)
    shuffled = engine.backplanes_map(names, lon[order][None, :], lat[order][None, :])
    for i in range(0, 512, 37):
        alone = engine.backplanes_map(names, lon[i : i + 1][None, :], lat[i : i + 1][None, :])
        for n in names:
            a, w = alone[n][0, 0], whole[n][0, i]
            assert (np.isnan(a) and np.isnan(w)) or abs(a - w) <= 2e-13, (n, i, a, w)
    inv = np.argsort(order)
    for n in names:
        a, b = whole[n][0], shuffled[n][0][inv]
        assert np.array_equal(np.isnan(a), np.isnan(b)), n
        assert np.nanmax(np.abs(a - b), initial=0.0) <= 2e-13, n
    # the image kernel: a 1-row frame through the disc centre against the same row of the full frame
    engine.set_disc(140.5, 90.0, 80.0, 0.0, 300, 181, True)
    full = engine.backplanes_img(['LAT-GRAPHIC', 'INCIDENCE', 'EMISSION'])
    rows = engine.backplanes_img_rows(['LAT-GRAPHIC', 'INCIDENCE', 'EMISSION'], 90, 1)
    for n in rows:
        assert np.array_equal(full[n][90], rows[n][0], equal_nan=True), n


def test_block_table_cache_routes_measured_and_chunk_callback(engine, oracle, jupiter):
    """
    The per-context state of the host-cube path (round 3): (a) the block table is reused by later calls
    with the same x/y map and rebuilt when the map (one cell!), the disc or the dtype changes, and can be
    switched off; (b) left to itself the library feeds three short chunks through each candidate route
    on the first large call, commits to the fastest, and every route gives the same bits;
    (c) the chunk callback reports every plane once, in order, and the number of planes redone.
    """
    import torch

    from planetmapper_amd import _lib

    sz, planes = 512, 200
    x0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 1.0)
    xm, ym = engine.xy_map(lon, lat)
    n0, n1 = xm.shape
    rng = np.random.default_rng(99)
    cube = engine.pinned_empty((planes, sz, sz))
    cube[...] = rng.standard_normal((planes, sz, sz))
    cube[rng.random(cube.shape) < 1e-3] = np.nan
    dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    out = torch.empty((planes, n0, n1), dtype=torch.float64, device='cuda')
    ref = torch.empty_like(out)
    engine.map_cube_device(torch.from_numpy(cube).cuda(), np.float64, planes, dxm, dym, n0, n1, ref)
    engine.synchronize()
    ref = ref.cpu().numpy()
    hits = lambda: engine.get_option(_lib.PM_OPT_BLOCK_TABLE_HITS)  # noqa: E731

    def fed(c=cube, xm_=dxm, ym_=dym, o=out):
        engine.map_cube_host_to_device(c, xm_, ym_, n0, n1, o[: c.shape[0]])
        engine.synchronize()
        return o[: c.shape[0]].cpu().numpy()

    try:
        # ---- (b) first call: exploration, then committed
        engine.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)
        assert np.array_equal(fed(), ref, equal_nan=True)
        ns = {r: engine.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(4)}
        assert ns[0] > 0 and ns[2] > 0 and ns[3] > 0 and ns[1] == 0, ns  # pinned cube: three candidates were timed
        best = min((v, r) for r, v in ns.items() if v > 0)[1]
        assert engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) == best
        h0 = hits()
        assert np.array_equal(fed(), ref, equal_nan=True)  # committed route, cached table
        assert engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) == best and hits() > h0
        # a pageable copy of the cube is another problem (no GPU fetch possible): measured afresh, same bits
        pageable = np.array(cube)
        assert np.array_equal(fed(pageable), ref, equal_nan=True)
        ns2 = {r: engine.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(4)}
        assert ns2[0] > 0 and ns2[3] > 0 and ns2[2] == 0, ns2
        # every explicit route: the same bits
        for route in (0, 1, 2, 3):
            engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, route)
            assert np.array_equal(fed(), ref, equal_nan=True), route
            assert engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) in ((route,) if route != 1 else (1, -1, best, 0, 2, 3))
        engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, 3)
        # ---- (a) cache: hit on an identical map, miss when ONE cell moves (and the moved map's result is right)
        h0 = hits()
        fed(cube[:16])
        assert hits() == h0 + 1
        xm2 = xm.copy()
        k = np.flatnonzero(np.isfinite(xm2))[1234]
        xm2.flat[k] += 2.75
        dxm2 = torch.from_numpy(xm2).cuda()
        got2 = fed(cube[:16], dxm2)
        assert hits() == h0 + 1  # rebuilt
        want2 = torch.empty((16, n0, n1), dtype=torch.float64, device='cuda')
        engine.map_cube_device(torch.from_numpy(cube[:16]).cuda(), np.float64, 16, dxm2, dym, n0, n1, want2)
        engine.synchronize()
        assert np.array_equal(got2, want2.cpu().numpy(), equal_nan=True)
        assert not np.array_equal(got2, ref[:16], equal_nan=True)
        fed(cube[:16], dxm2)
        assert hits() == h0 + 2
        # another dtype with the same map: its own table
        c32 = cube[:16].astype(np.float32)
        got32 = fed(c32, dxm2)
        assert hits() == h0 + 2
        engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, 0)
        assert np.array_equal(got32, fed(c32, dxm2), equal_nan=True)
        engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, 3)
        # switched off: never a hit, same results
        engine.set_option(_lib.PM_OPT_BLOCK_TABLE_CACHE, 0)
        h1 = hits()
        assert np.array_equal(fed(cube[:16]), ref[:16], equal_nan=True) and np.array_equal(fed(cube[:16]), ref[:16], equal_nan=True)
        assert hits() == h1
        engine.set_option(_lib.PM_OPT_BLOCK_TABLE_CACHE, 1)
        # ---- (c) chunk callback: every plane once, in order; redone planes counted
        for route in (0, 2, 3, -1):
            engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, route)
            seen = []
            engine.set_chunk_callback(lambda first, n: seen.append((first, n)))
            try:
                fed()
            finally:
                engine.set_chunk_callback(None)
            assert seen and seen[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(seen, seen[1:])), (route, seen[:4])
            assert seen[-1][0] + seen[-1][1] == planes
            assert engine.last_redo_planes() == 0
        seen = []
        engine.set_chunk_callback(lambda first, n: seen.append((first, n)))
        try:
            dirty = np.array(cube[:12])
            dirty[5][200:300, 200:330] = np.inf
            dirty[9][100:140, 250:300] = -np.inf
            fed(dirty)
            assert engine.last_redo_planes() == 2 and seen[-1][0] + seen[-1][1] == 12
            seen.clear()
            engine.map_cube_device(torch.from_numpy(dirty).cuda(), np.float64, 12, dxm, dym, n0, n1, out[:12])
            assert seen == [(0, 12)]
            engine.synchronize()
            assert engine.last_redo_planes() == 2
        finally:
            engine.set_chunk_callback(None)
    finally:
        engine.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, -1)
        engine.set_option(_lib.PM_OPT_BLOCK_TABLE_CACHE, 1)
        engine.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)


def test_c_abi_sharded_cube_into_one_host_array_without_a_collective(engine, oracle, jupiter):
    """pm_map_cube_sharded(mem = PM_MEM_HOST, gather = 0): the ranks of a job write their planes into ONE
    (P, n0, n1) host array - rehearsed here as the 3 'ranks' of a communicator-less call, one after the other."""
    import ctypes

    from planetmapper_amd import _lib
    from planetmapper_amd.distributed import shard_bounds

    sz, planes = 96, 7
    x0 = (sz - 1) / 2
    engine.set_geometry(jupiter)
    engine.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon, lat = oracle.rectangular_grid(jupiter, 10.0)
    xm, ym = engine.xy_map(lon, lat)
    n0, n1 = xm.shape
    rng = np.random.default_rng(8)
    cube = rng.standard_normal((planes, sz, sz))
    whole = engine.map_cube(cube, xm, ym)
    out = np.full((planes, n0, n1), -5.0)
    lib, ctx = engine._lib, engine._ctx
    # (comm = NULL is a single rank: it owns every plane)
    engine._check(lib.pm_map_cube_sharded(ctx, None, cube.ctypes.data, 0, planes, xm.ctypes.data, ym.ctypes.data, n0, n1, 1, 1,
                                          out.ctypes.data, _lib.PM_MEM_HOST, 0))
    assert np.array_equal(out, whole, equal_nan=True)
    with pytest.raises(ValueError):
        engine._check(lib.pm_map_cube_sharded(ctx, None, cube.ctypes.data, 0, planes, xm.ctypes.data, ym.ctypes.data, n0, n1, 1, 1,
                                              out.ctypes.data, _lib.PM_MEM_HOST, 1))
    assert shard_bounds(planes, 3, 2) == (6, 7, 3)


@pytest.mark.parametrize('which', ['jupiter', 'saturn'])
def test_all_planes_from_one_launch_equal_the_launch_per_group(engine, oracle, jupiter, saturn, which):
    """
    save_observation's request (observation.py:1269-1279): all 26 planes of a frame. With PM_OPT_FUSE_PLANES
    the spheroid path computes them in ONE launch (k_disc_sph<FLAGS, TRI, SKY>: the sky / limb
    planes ride along with the intercept planes); without, one launch per group as before. The same code
    runs in both: every plane bit-identical - ragged frame, row block, subsets of planes - and inside the
    bars against the oracle.
    """
    from planetmapper_amd import _lib

    g, sz_x, sz_y, r0, rot = (jupiter, 517, 389, 170.0, 0.58) if which == 'jupiter' else (saturn, 700, 640, 120.0, 0.35)
    engine.set_geometry(g)
    engine.set_disc(sz_x / 2.1, sz_y / 1.9, r0, rot, sz_x, sz_y, True)
    subsets = [oracle.PLANE_NAMES, ['LON-GRAPHIC', 'RA', 'PIXEL-Y'], ['EMISSION', 'LIMB-DISTANCE', 'KM-X', 'RING-RADIUS'],
               ['DISTANCE', 'ANGULAR-X', 'LIMB-LAT-GRAPHIC', 'DEC']]  # fmt: skip
    was_general = engine.get_option(_lib.PM_OPT_GENERAL_KERNEL)  # (the module's engine may be inside an engine_fg leg)
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
    try:
        for names in subsets:
            engine.set_option(_lib.PM_OPT_FUSE_PLANES, 1)
            one = engine.backplanes_img(names)
            assert engine.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) == 1
            rows_one = engine.backplanes_img_rows(names, 101, 57)
            engine.set_option(_lib.PM_OPT_FUSE_PLANES, 0)
            per_group = engine.backplanes_img(names)
            rows_group = engine.backplanes_img_rows(names, 101, 57)
            for n in names:
                assert np.array_equal(one[n], per_group[n], equal_nan=True), (which, n)
                assert np.array_equal(rows_one[n], rows_group[n], equal_nan=True), (which, n)
                assert np.array_equal(rows_one[n], one[n][101:158], equal_nan=True), (which, n)
        engine.set_option(_lib.PM_OPT_FUSE_PLANES, 1)
        d = oracle.make_disc(sz_x / 2.1, sz_y / 1.9, r0, 0.0, sz_x, sz_y)
        d.rotation_rad = rot
        _compare(engine.backplanes_img(oracle.PLANE_NAMES), oracle.backplanes_img(g, d, oracle.PLANE_NAMES), oracle.PLANE_NAMES, g, r0=r0)
    finally:
        engine.set_option(_lib.PM_OPT_FUSE_PLANES, 0)
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, was_general)


def test_cache_invalidation_matrix_through_the_real_engine(engine, jupiter):
    """
    The reference's cache contract (tests/test_body_xy.py:2495-2590) on the HIP engine: for five kinds of
    change (disc parameters, image size, an altitude, and both with an altitude) x all 26 backplanes x
    image / map space, a body that had its cache filled, was changed and changed back must give what a
    clean body with the same parameters gives - and, back at the start, exactly what it gave before.
    """
    from planetmapper_amd import BodyXY

    def make_body():
        body = BodyXY('Jupiter', geometry=jupiter, nx=6, ny=5, engine=engine)
        body.set_disc_params(2.5, 2, 2, 45)
        return body

    changes = {
        'set_disc_params': (lambda b: b.set_disc_params(3, 1.5, 2.5, 42), lambda b: b.set_disc_params(5, 3, 2, 123), 0.0),
        'set_img_size': (lambda b: b.set_img_size(6, 2), lambda b: b.set_img_size(3, 4), 0.0),
        'alt': (lambda b: None, lambda b: None, 123.456),
        'set_disc_params+alt': (lambda b: b.set_disc_params(3, 1.5, 2.5, 42), lambda b: b.set_disc_params(5, 3, 2, 123), 123.456),
        'set_img_size+alt': (lambda b: b.set_img_size(6, 2), lambda b: b.set_img_size(3, 4), 123.456),
    }  # fmt: skip
    names = list(make_body().backplanes.keys())
    assert len(names) == 26
    n_checked = 0
    for change_name, (reset, change, alt) in changes.items():
        for space in ('img', 'map'):
            def get(b, name, a):
                return b.get_backplane_img(name, alt=a) if space == 'img' else b.get_backplane_map(name, alt=a, degree_interval=45)

            # one body per (change, space) walks all planes: its cache holds every plane family at each step
            body = make_body()
            reset(body)
            before = {n: get(body, n, 0.0) for n in names}
            clean = make_body()
            change(body)
            change(clean)
            for n in names:
                assert np.allclose(get(body, n, alt), get(clean, n, alt), rtol=1e-5, atol=1e-8, equal_nan=True), (change_name, space, n)
            clean = make_body()
            reset(body)
            reset(clean)
            for n in names:
                back = get(body, n, 0.0)
                assert np.allclose(back, get(clean, n, 0.0), rtol=1e-5, atol=1e-8, equal_nan=True), (change_name, space, n)
                assert np.array_equal(back, before[n], equal_nan=True), (change_name, space, n)
                n_checked += 1
    assert n_checked == 5 * 2 * 26


def test_mapping_visible_areas_through_the_real_engine(engine, oracle):
    """
    tests/test_body_xy.py:2592-2607 restated: a map cell is mapped exactly where it is visible - RA and the
    mapped image finite where the emission angle is <= 90 deg, NaN beyond. Near observer (4 radii) so that
    the horizon cuts across the map well away from the 90 deg meridians.
    """
    from planetmapper_amd import BodyXY
    from test_motion_model import load

    _, g = load('jupiter_near_field')
    body = BodyXY('Jupiter', geometry=g, sz=10, engine=engine)
    body.set_disc_params(5, 5, 3, 0)
    emission = body.get_backplane_map('EMISSION', degree_interval=15)
    ra = body.get_backplane_map('RA', degree_interval=15)
    img = body.map_img(np.ones((10, 10)), degree_interval=15)
    assert (emission <= 90).sum() > 20 and (emission > 90).sum() > 100
    assert np.isfinite(ra[emission <= 90]).all() and not np.isfinite(ra[emission > 90]).any()
    # (map_img needs the cell's pixel inside the frame as well: the disc, r0 = 3 in a 10 x 10 frame, is)
    assert np.isfinite(img[emission <= 90]).all() and not np.isfinite(img[emission > 90]).any()


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['jupiter', 'saturn', 'near_field'])
def test_closed_form_light_time_against_the_reference_sequence(oracle, jupiter, saturn, which):
    """
    k_disc_sph settles the waves clear of the limb with the light time in closed form (profiles/EXPERIMENTS.md, round 3); the
    library can be told to walk the reference's own sequence of epochs everywhere instead (PM_OPT_LT_MODE 1) or to
    take the round-2 path (2: Newton step on the seed). Same frame through all three:
    NaN masks identical - the closed form may not decide a single limb pixel differently -, values inside the
    bars against the oracle in every mode, and the closed form no further from the reference's sequence than
    one epoch quantum allows (the two can round `et - lt` to neighbouring doubles: 3e-8 s, 3e-10 deg of
    longitude on Jupiter, amplified towards the limb as every error is).
    """
    from planetmapper_amd.engine import Engine

    g = {'jupiter': jupiter, 'saturn': saturn}.get(which) or _near_field_geometry(1.2e6)
    sz = 1536
    x0, y0, r0, rot = sz / 2 - 3.25, sz / 2 + 11.5, 0.43 * sz, 17.0
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'DISTANCE', 'RADIAL-VELOCITY']
    outs = {}
    from planetmapper_amd import _lib

    for mode in ('0', '1', '2'):
        eng = Engine(0)
        try:
            eng.set_option(_lib.PM_OPT_LT_MODE, int(mode))
            eng.set_geometry(g)
            eng.set_disc(x0, y0, r0, float(np.deg2rad(rot)), sz, sz, True)
            outs[mode] = eng.backplanes_img(names)
        finally:
            eng.close()
    d = oracle.make_disc(x0, y0, r0, rot, sz, sz)
    d.rotation_rad = float(np.deg2rad(rot))
    ref = oracle.backplanes_img(g, d, names)
    for mode in ('0', '1', '2'):
        for n in names:
            assert np.array_equal(np.isnan(outs[mode][n]), np.isnan(ref[n])), (which, mode, n)
        _compare(outs[mode], ref, names, g, r0=r0)
    on = np.isfinite(ref['EMISSION'])
    assert 0.3 < on.mean() < 0.7
    # one epoch quantum as an angle on the body: the spin plus the target's motion across the line of sight
    quantum = np.spacing(g.et)
    vt = float(np.linalg.norm(np.asarray(g.VT)))
    per_quantum = np.rad2deg(quantum * (abs(g.wdot) + vt / min(g.radii)))
    kappa = 1.0 / np.maximum(np.cos(np.deg2rad(ref['EMISSION'][on])), 1e-3)
    for n in ('LON-GRAPHIC', 'LAT-GRAPHIC', 'INCIDENCE', 'EMISSION'):
        diff = np.abs(outs['0'][n][on] - outs['1'][n][on])
        diff = np.minimum(diff, 360.0 - diff) if n == 'LON-GRAPHIC' else diff
        lat_k = 1.0 / np.maximum(np.cos(np.deg2rad(ref['LAT-GRAPHIC'][on])), 1e-3) if n == 'LON-GRAPHIC' else 1.0
        assert (diff <= (1.5 * per_quantum + 2e-10) * kappa * lat_k).all(), (which, n, float(diff.max()), per_quantum)
        # ... and most pixels chose the same quantum: the two differ by rounding noise only
        assert np.median(diff) <= 1.5e-10, (which, n, float(np.median(diff)))


@pytest.mark.gpu
def test_limb_bound_is_never_tighter_than_the_limb(engine, oracle, jupiter):
    """
    fill_params replaces the reference's pre-mask radius (1.05 r0 + 1 pixels) by the limb's own reach where that is
    tighter, and gives frames without a pre-mask one. Discs of many sizes, off-centre, rotated, with altitude
    offsets and with `optimize_speed=False`: the NaN mask is the oracle's (which tests every pixel the reference
    does) and the values sit inside the bars both ways. (Not bit-identical to each other: with the pre-mask off a
    wave at the limb can hold a candidate it did not hold before, and a wave with a lane in the limb band walks the
    reference's sequence of epochs instead of the closed form - profiles/EXPERIMENTS.md, round 3 - for all of its lanes.)
    """
    rng = np.random.default_rng(20261004)
    engine.set_geometry(jupiter)
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'EMISSION']  # (the longitude's bar needs the latitude: 1 / cos(lat))
    for k in range(24):
        nx, ny = int(rng.integers(90, 700)), int(rng.integers(90, 700))
        r0 = float(rng.uniform(8.0, 0.8 * max(nx, ny)))
        x0, y0 = float(rng.uniform(-0.2, 1.2) * nx), float(rng.uniform(-0.2, 1.2) * ny)
        rot = float(rng.uniform(0, 360))
        alt = float(rng.choice([0.0, 0.0, 500.0, 4000.0, -300.0]))
        d = oracle.make_disc(x0, y0, r0, rot, nx, ny)
        d.rotation_rad = float(np.deg2rad(rot))
        got = {}
        for speed in (True, False):
            engine.set_disc(x0, y0, r0, float(np.deg2rad(rot)), nx, ny, speed)
            got[speed] = engine.backplanes_img(names, alt=alt)
            d.optimize_speed = int(speed)
            ref = oracle.backplanes_img(jupiter, d, names, alt=alt)
            for n in names:
                assert np.array_equal(np.isnan(got[speed][n]), np.isnan(ref[n])), (k, speed, n, nx, ny, r0, x0, y0, alt)
            _compare(got[speed], ref, names, jupiter, r0=r0, flat=False)
        fin = np.isfinite(got[True]['EMISSION'])
        assert np.allclose(got[True]['EMISSION'][fin], got[False]['EMISSION'][fin], rtol=0, atol=1e-7), k


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['jupiter', 'saturn', 'triaxial'])
def test_a_pixel_does_not_depend_on_the_pixels_that_share_its_wave(engine, jupiter, saturn, which, general=0):
    """
    The frame kernel takes wave-uniform shortcuts (the closed-form light time for waves clear of the limb, the
    angle forms of vsep_fast / lat_of_normal): each is either the same operations on the same operands or chosen
    per lane, so what a pixel gets must not depend on which 63 pixels share its wave. The same frame shifted by
    17 and by 40 columns (every wave then holds other pixels, the limb crosses other waves): bit-identical planes
    on the common pixels. (The spheroid kernel: it takes the view angles from x - x0, which the shift leaves
    unchanged to the bit. The general kernel follows the reference's affine map with its constant term, whose
    rounding moves with x0 - a different INPUT, 1e-18 rad, that flips the epoch quantum of a few pixels.)
    """
    import copy

    g = {'jupiter': jupiter, 'saturn': saturn, 'triaxial': jupiter}[which]
    if which == 'triaxial':  # the TRI variant of the frame kernel: every pixel walks the reference's sequence
        g = copy.deepcopy(g)
        g.radii[1] = g.radii[0] * 0.97
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC', 'LAT-CENTRIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH',
             'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'] + (['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE'] if which == 'saturn' else [])  # fmt: skip
    from planetmapper_amd import _lib

    nx, ny = 1100, 900
    x0, y0, r0, rot = 531.25, 466.5, 402.75, float(np.deg2rad(23.0))
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, general)  # (whatever an earlier test left)
    engine.set_geometry(g)
    engine.set_disc(x0, y0, r0, rot, nx, ny, True)
    base = engine.backplanes_img(names)
    assert 0.3 < np.isfinite(base['EMISSION']).mean() < 0.8
    for shift in (17, 40):
        engine.set_disc(x0 + shift, y0, r0, rot, nx + shift, ny, True)
        moved = engine.backplanes_img(names)
        for n in names:
            a, b = base[n], moved[n][:, shift:]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (which, shift, n, int((~same).sum()), float(np.nanmax(np.abs(a - b))))
    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)


@pytest.mark.gpu
def test_debug_knobs_of_the_environment_need_pm_debug_env(jupiter):
    """
    A stray PM_LT_MODE / PM_FORCE_GENERAL / PM_FUSE_PLANES in a user's shell must not select another algorithm: the
    library reads these A/B knobs only with PM_DEBUG_ENV=1 beside them (pm_debug_env, pm_host.hip.h); each has a
    pm_set_option() form.
    """
    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine

    knobs = {'PM_LT_MODE': '2', 'PM_FORCE_GENERAL': '1', 'PM_FUSE_PLANES': '1', 'PM_HOSTPIPE_TRACE': '1', 'PM_SM_BATCH_PLANES': '2'}
    saved = {k: os.environ.get(k) for k in list(knobs) + ['PM_DEBUG_ENV']}

    def options():
        eng = Engine(0)
        try:
            return tuple(eng.get_option(o) for o in (_lib.PM_OPT_LT_MODE, _lib.PM_OPT_GENERAL_KERNEL, _lib.PM_OPT_FUSE_PLANES,
                                                     _lib.PM_OPT_TRACE, _lib.PM_OPT_SM_BATCH_PLANES))
        finally:
            eng.close()

    try:
        os.environ.pop('PM_DEBUG_ENV', None)
        os.environ.update(knobs)
        assert options() == (0, 0, 0, 0, 0)  # ignored
        os.environ['PM_DEBUG_ENV'] = '1'
        assert options() == (2, 1, 1, 1, 2)  # honoured
        os.environ['PM_DEBUG_ENV'] = '0'
        assert options() == (0, 0, 0, 0, 0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # the option forms validate their values
    eng = Engine(0)
    try:
        with pytest.raises(ValueError):
            eng.set_option(_lib.PM_OPT_LT_MODE, 3)
        with pytest.raises(ValueError):
            eng.set_option(_lib.PM_OPT_SM_BATCH_PLANES, -1)
        eng.set_option(_lib.PM_OPT_TRACE, 0)
    finally:
        eng.close()


@pytest.mark.gpu
def test_result_planes_of_the_shim_are_recycled_on_the_real_engine(jupiter):
    """
    tests/test_api_host.py's recycling test against the real engine: the planes of a cold getter live in page-locked arrays of
    the engine's pool; after `set_disc_params` they are written again - unless the caller still holds one, which then keeps its
    values while the new plane goes elsewhere.
    """
    from planetmapper_amd import BodyXY
    from planetmapper_amd.engine import Engine

    eng = Engine(0)
    try:
        body = BodyXY('Jupiter', geometry=jupiter, nx=700, ny=600, engine=eng)
        body.set_disc_params(350, 300, 250, 10)
        lon = body.get_lon_img()
        first = lon.ctypes.data
        assert not lon.flags.writeable and eng._plane_bytes == 2 * 700 * 600 * 8  # lon + lat, both from the pool
        del lon
        body.set_disc_params(352, 300, 250, 10)
        assert sum(len(v) for v in eng._plane_pool.values()) == 2
        lon2 = body.get_lon_img()
        lat2 = body.get_lat_img()
        assert {lon2.ctypes.data, lat2.ctypes.data} >= {first} and sum(len(v) for v in eng._plane_pool.values()) == 0
        kept = lon2[:, 300:310]
        values = kept.copy()
        del lon2
        body.set_disc_params(380, 300, 250, 10)
        lon3 = body.get_lon_img()
        assert lon3.ctypes.data != kept.ctypes.data - 300 * 8 and np.array_equal(kept, values, equal_nan=True)
        ref = eng.backplanes_img(['LON-GRAPHIC', 'LAT-GRAPHIC'])
        assert np.array_equal(lon3, ref['LON-GRAPHIC'], equal_nan=True)
    finally:
        eng.close()


@pytest.mark.gpu
def test_stage_record_of_a_host_fed_cube_call(engine, jupiter):
    """PM_OPT_LAST_STAGE_NS: the sequential stages add up to the call; device-side sums only with PM_OPT_TRACE bit 4"""
    import time

    import torch

    from planetmapper_amd import _lib

    sz, planes = 512, 40
    engine.set_geometry(jupiter)
    engine.set_disc(255.5, 255.5, 200.0, 0.1, sz, sz, True)
    lon, lat = np.meshgrid(np.arange(1.0, 360, 2.0)[::-1], np.arange(-89.0, 90, 2.0))
    xm, ym = engine.xy_map(lon, lat)
    dx, dy = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    cube = engine.pinned_empty((planes, sz, sz))
    cube[...] = np.random.default_rng(2).standard_normal(cube.shape)
    out = torch.empty((planes,) + xm.shape, dtype=torch.float64, device='cuda')
    for trace in (0, 4):
        engine.set_option(_lib.PM_OPT_TRACE, trace)
        try:
            for _ in range(3):
                t = time.perf_counter()
                engine.map_cube_host_to_device(cube, dx, dy, xm.shape[0], xm.shape[1], out)
                engine.synchronize()
                wall = (time.perf_counter() - t) * 1e3
            st = engine.last_stages_ms()
        finally:
            engine.set_option(_lib.PM_OPT_TRACE, 0)
        parts = st['tables'] + st['plan'] + st['issue'] + st['drain'] + st['finish']
        assert 0 < st['total'] <= wall * 1.05 and abs(parts - st['total']) <= 0.05 * st['total'] + 0.02, (st, wall)
        assert st['first_fill'] <= st['issue'] + 1e-6 and st['collect'] <= st['issue'] + 1e-6
        if trace:
            assert st['kernels_device'] > 0 and (st['dma_device'] > 0 or engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) in (1, 2))
        else:
            assert st['dma_device'] == 0 and (st['kernels_device'] == 0 or engine.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) == 4)
    assert np.array_equal(out.cpu().numpy(), engine.map_cube(np.array(cube), xm, ym, 'linear', True), equal_nan=True)


@pytest.mark.gpu
def test_bench_interpolations_section(jupiter):
    """bench.py's `interpolations` record (N = 1): every map_img mode on both maps, times positive and in the order the
    work implies (a fine map costs more than the 1 deg map; one plane alone costs more per plane than a batch)"""
    import bench

    rec = bench.interpolations_section(0, jupiter, planes=6, sz=256)
    assert set(rec) == {'workload', 'us_per_plane'} and len(rec['us_per_plane']) == 2
    coarse, fine = (rec['us_per_plane'][k] for k in sorted(rec['us_per_plane'], key=lambda k: float(k.split()[0]), reverse=True))
    assert set(coarse) == set(fine) == {'nearest', 'linear', 'quadratic', 'cubic', 'cubic, one plane', '5', 'smooth'}
    assert all(v > 0 for v in coarse.values()) and all(v > 0 for v in fine.values())
    assert fine['linear'] > coarse['linear'] and fine['smooth'] > coarse['smooth']
    assert coarse['cubic, one plane'] > coarse['cubic']
