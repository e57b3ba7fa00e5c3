"""
Row f3 (spline / 'smooth' / smoothing-spline map_img, body_xy.py:1651-1853) AT CUBE SCALE, `-m gpu`: the tiled
banded solves, the lazy plane medians and the batched smoothing-spline fits are tuned on 1024^2 planes by the dozen,
so that is where they are held against the oracle here - 1024 x 1024 and 1000 x 1031 (no power-of-two pitch, a last tile
that is not full) x 42 planes whose clean states interleave: finite; sparse NaN (every NaN has a finite neighbour);
NaN blocks whose inner pixels have NO finite neighbour (only those planes take the lazy-median redo); +-inf; all NaN.
NaN masks identical, values <= 1e-9 of scale (f32 input: the same bar - both sides convert the same f32 samples).
The oracle (serial per plane) runs on the box's cores through a thread pool: ctypes drops the GIL.
"""

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATES = ('finite', 'sparse_nan', 'nan_blocks', 'inf', 'all_nan', 'block_and_inf', 'finite')


@pytest.fixture(scope='module')
def engine():
    from planetmapper_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def oracle():
    from oracle import oracle as o

    return o


def oracle_map_cube_mt(oracle, cube, xm, ym, interp, prop, **kw):
    """oracle.map_cube over plane slices in parallel (planes are independent in every interpolation)"""
    n = cube.shape[0]
    workers = max(1, min(n, len(os.sched_getaffinity(0)), 16))
    bounds = np.linspace(0, n, workers + 1).astype(int)
    spans = [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    with ThreadPoolExecutor(len(spans)) as pool:
        parts = list(pool.map(lambda ab: oracle.map_cube(cube[ab[0] : ab[1]], xm, ym, interp, prop, **kw), spans))
    return np.concatenate(parts, axis=0)


def make_cube(n_planes, ny, nx, seed, dtype=np.float64, noise=1.0):
    """planes in every clean state, interleaved (plane p is in state STATES[p % 7])"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ny, 0:nx]
    cube = np.empty((n_planes, ny, nx), dtype=np.float64)
    states = []
    for p in range(n_planes):
        st = STATES[p % len(STATES)]
        states.append(st)
        pl = np.sin(xx / (7.0 + p)) * np.cos(yy / (11.0 + 0.5 * p)) * (3 + p) + noise * rng.standard_normal((ny, nx))
        if st == 'sparse_nan':
            # isolated NaN pixels on a lattice with jitter < 1: no two of them touch, every one has finite neighbours
            pl[3::7, 2::5] = np.nan
        elif st == 'nan_blocks':
            for _ in range(6):
                i, j = int(rng.integers(0, ny - 40)), int(rng.integers(0, nx - 40))
                pl[i : i + int(rng.integers(3, 40)), j : j + int(rng.integers(3, 40))] = np.nan
            pl[0:5, 0:4] = np.nan  # a corner block
        elif st == 'inf':
            m = rng.random((ny, nx))
            pl[m < 0.002] = np.inf
            pl[m > 0.998] = -np.inf
        elif st == 'all_nan':
            pl[:] = np.nan
        elif st == 'block_and_inf':
            pl[ny // 2 - 2 : ny // 2 + 3, nx // 3 : nx // 3 + 5] = np.nan  # 5 x 5: the inner 3 x 3 need the median
            pl[ny // 2, nx // 3 + 2] = np.inf
            pl[10, 10] = -np.inf
        cube[p] = pl
    return cube.astype(dtype), states


def setup_maps(engine, oracle, g, ny, nx, deg=1.0):
    x0, y0 = (nx - 1) / 2, (ny - 1) / 2
    r0 = 0.45 * min(nx, ny)
    engine.set_geometry(g)
    engine.set_disc(x0, y0, r0, 0.3, nx, ny, True)
    d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
    d.rotation_rad = 0.3
    lon, lat = oracle.rectangular_grid(g, deg)
    return oracle.xy_map(g, d, lon, lat)


def map_resident(engine, cube, xm, ym, interp, prop, **kw):
    """the device-resident route (what the kernels were tuned on)"""
    import torch

    dc = torch.from_numpy(cube).cuda()
    dx, dy = torch.from_numpy(np.ascontiguousarray(xm)).cuda(), torch.from_numpy(np.ascontiguousarray(ym)).cuda()
    out = torch.empty((cube.shape[0],) + xm.shape, dtype=torch.float64, device='cuda')
    engine.map_cube_device(dc, cube.dtype, cube.shape[0], dx, dy, xm.shape[0], xm.shape[1], out, interp, prop, **kw)
    engine.synchronize()
    return out.cpu().numpy()


def assert_close(a, b, label, bar=1e-9, scale_floor=1.0):
    assert np.array_equal(np.isnan(a), np.isnan(b)), label
    assert not np.isinf(a).any(), label
    fin = np.isfinite(b)
    assert fin.sum() > 1000 * a.shape[0] // 2, label
    worst = float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(scale_floor, np.abs(b[fin]))))
    assert worst <= bar, (label, worst)
    return worst


@pytest.mark.parametrize('ny,nx', [(1024, 1024), (1000, 1031)])
def test_interpolating_splines_and_smooth_at_cube_scale(engine, oracle, jupiter, ny, nx):
    n_planes = 42
    cube, states = make_cube(n_planes, ny, nx, seed=ny + nx)
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx)
    worst = {}
    for interp in ('quadratic', 'cubic', (1, 3), 5, 'smooth'):
        if interp == 'smooth':
            engine.set_smooth_options(5, 10_000)
        for prop in (True, False):
            a = map_resident(engine, cube, xm, ym, interp, prop)
            b = oracle_map_cube_mt(oracle, cube, xm, ym, (interp, interp) if isinstance(interp, int) else interp, prop)
            for p in range(n_planes):  # per plane: the failure names the clean state
                w = assert_close(a[p : p + 1], b[p : p + 1], (interp, prop, p, states[p])) if states[p] != 'all_nan' else 0.0
                worst[str(interp)] = max(worst.get(str(interp), 0.0), w)
            assert np.isnan(a[[p for p in range(n_planes) if states[p] == 'all_nan']]).all(), (interp, prop)
    print('\n[cube-scale splines] worst |HIP - oracle| / scale per interpolation:', {k: f'{v:.2e}' for k, v in worst.items()})
    # the host route (planes through the staging pipeline) gives the resident route's bits
    for interp in ('cubic', 'smooth'):
        assert np.array_equal(engine.map_cube(cube[:9], xm, ym, interp, True), map_resident(engine, cube[:9], xm, ym, interp, True),
                              equal_nan=True), interp
    # a plane's result does not depend on the planes mapped with it (the lazy-median redo touches flagged planes only)
    for p in (0, 2, 5, 3):
        assert np.array_equal(map_resident(engine, cube[p : p + 1], xm, ym, 'cubic', False)[0],
                              map_resident(engine, cube, xm, ym, 'cubic', False)[p], equal_nan=True), p


def test_segmented_spline_solves_give_the_line_solves_results(engine, oracle, jupiter):
    """
    Few, large planes: the banded solves cut every line into segments whose substitutions start `warm` samples early
    (k_spline_seg_*, PM_OPT_SPLINE_SEGMENT). The factors' recursion forgets geometrically, so a segment's steps are the
    serial substitution's operations on - to 4e-24 - its operands, and both forms run the same tile code: the results are
    held here to be the one-lane-per-line form's BITS, on every degree, clean state, dtype and on sizes whose last tile /
    last segment are not full.
    """
    from planetmapper_amd import _lib

    def both(cube, xm, ym, interp, prop, seg):
        engine.set_option(_lib.PM_OPT_SPLINE_SEGMENT, -1)
        a = map_resident(engine, cube, xm, ym, interp, prop)
        assert engine.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT) == 0
        engine.set_option(_lib.PM_OPT_SPLINE_SEGMENT, seg)
        b = map_resident(engine, cube, xm, ym, interp, prop)
        assert engine.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT) > 0
        return a, b

    total = 0
    try:
        for ny, nx, n_planes, segs in ((1000, 1031, 7, (64, 112, 256)), (517, 300, 7, (64, 128)), (1024, 1024, 3, (64, 512, 2048))):
            cube, states = make_cube(n_planes, ny, nx, seed=ny * 3 + nx)
            xm, ym = setup_maps(engine, oracle, jupiter, ny, nx)
            for interp in ('quadratic', 'cubic', (1, 3), (4, 2), 5):
                for prop, seg in zip((True, False, True), segs):
                    a, b = both(cube, xm, ym, interp, prop, seg)
                    assert np.array_equal(np.isnan(a), np.isnan(b)), (ny, nx, interp, seg)
                    fin = np.isfinite(a)
                    assert fin.sum() > 1000 * n_planes // 2
                    assert np.array_equal(a, b, equal_nan=True), (ny, nx, interp, seg, float(np.nanmax(np.abs(a - b))))
                    total += int(fin.sum())
            # other sample types take the same route (the forward pass of axis 0 reads the cube in its own dtype)
            c32 = cube.astype(np.float32)
            a, b = both(c32, xm, ym, 'cubic', False, segs[0])
            assert np.array_equal(a, b, equal_nan=True)
            ci = (np.nan_to_num(cube[:2], nan=0.0, posinf=0.0, neginf=0.0) * 10).clip(-30000, 30000).astype(np.int16)
            a, b = both(ci, xm, ym, 'cubic', True, segs[0])
            assert np.array_equal(a, b, equal_nan=True)
        print(f'\n[segmented solves] bit-identical to the line solves on {total} mapped samples')
        # the library's own choice: few planes of a large image are segmented, a batch that fills the chip is not
        engine.set_option(_lib.PM_OPT_SPLINE_SEGMENT, 0)
        cube, _ = make_cube(1, 1024, 1024, seed=9)
        xm, ym = setup_maps(engine, oracle, jupiter, 1024, 1024)
        map_resident(engine, cube, xm, ym, 'cubic', True)
        assert engine.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT) >= 64
        small, _ = make_cube(2, 200, 200, seed=9)
        xs, ys = setup_maps(engine, oracle, jupiter, 200, 200)
        map_resident(engine, small, xs, ys, 'cubic', True)
        assert engine.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT) == 0
    finally:
        engine.set_option(_lib.PM_OPT_SPLINE_SEGMENT, 0)


@pytest.mark.parametrize('n_planes,ny,nx', [(1, 4096, 4096), (2, 2048, 3000)])
def test_interpolating_splines_on_a_few_large_planes(engine, oracle, jupiter, n_planes, ny, nx):
    """the headline frame size as ONE plane (and two planes of 2048 x 3000): the segmented solves against the oracle"""
    from planetmapper_amd import _lib

    cube, states = make_cube(n_planes, ny, nx, seed=ny + nx + 1)
    cube[0, 100:140, 200:260] = np.nan  # a block whose inner pixels need the plane's nanmedian when cleaned
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx)
    worst = {}
    for interp, prop in (('cubic', True), ('cubic', False), ('quadratic', False), (5, True)):
        a = map_resident(engine, cube, xm, ym, interp, prop)
        assert engine.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT) >= 64
        b = oracle_map_cube_mt(oracle, cube, xm, ym, (interp, interp) if isinstance(interp, int) else interp, prop)
        worst[f'{interp}/{prop}'] = assert_close(a, b, (interp, prop))
    print('\n[few large planes] worst |HIP - oracle| / scale:', {k: f'{v:.2e}' for k, v in worst.items()})


def test_smooth_interpolation_4x4_form_gives_the_gap_aware_form_bits(engine, oracle, jupiter):
    """
    k_reproject_smooth builds a cell's four fine-grid nodes from the 4 x 4 finite pixels around it where it can (six sets of
    PCHIP coefficients) and from the gap-aware search otherwise (twenty): both are the same arithmetic, so the output must
    not depend on which one a cell took. `PM_OPT_GENERAL_KERNEL` sends every cell through the gap-aware form. Oversampling 1
    puts every second fine node ON a pixel (the 4 x 4 form returns the sample there), 5 and 3 put them inside the cells.
    """
    from planetmapper_amd import _lib

    ny, nx = 500, 523
    cube, states = make_cube(14, ny, nx, seed=31)
    cube32 = cube.astype(np.float32)
    # integer planes: the window's rows are fetched as ONE load at an address aligned to the element only (2 bytes, 1 byte)
    ci = np.nan_to_num(cube[:5], nan=0.0, posinf=0.0, neginf=0.0) * 10
    ints = [ci.clip(np.iinfo(dt).min, np.iinfo(dt).max).astype(dt) for dt in (np.int16, np.uint16, np.uint8, np.int32)]
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=0.5)
    try:
        for oversample, max_size in ((5, 10_000), (1, 10_000), (3, 10_000), (5, 700), (5, 300)):
            engine.set_smooth_options(oversample, max_size)
            for c in [cube, cube32] + (ints if oversample == 5 else []):
                for prop in (True, False):
                    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
                    a = map_resident(engine, c, xm, ym, 'smooth', prop)
                    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 1)
                    b = map_resident(engine, c, xm, ym, 'smooth', prop)
                    assert np.array_equal(a, b, equal_nan=True), (oversample, max_size, c.dtype, prop,
                                                                  float(np.nanmax(np.abs(a - b))))
                    assert np.isfinite(a).sum() > 1000 * 4
            if max_size == 10_000 and oversample == 5:  # ... and the integer planes against the oracle
                for c in ints[:2]:
                    engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
                    assert_close(map_resident(engine, c, xm, ym, 'smooth', True), oracle_map_cube_mt(oracle, c, xm, ym, 'smooth', True),
                                 ('smooth', c.dtype))
    finally:
        engine.set_option(_lib.PM_OPT_GENERAL_KERNEL, 0)
        engine.set_smooth_options(5, 10_000)


def test_splines_f32_and_integer_planes_at_cube_scale(engine, oracle, jupiter):
    ny = nx = 1024
    cube64, states = make_cube(14, ny, nx, seed=77)
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx)
    c32 = cube64.astype(np.float32)
    for interp in ('quadratic', 'cubic', (1, 3), 5, 'smooth'):
        for prop in (True, False):
            a = map_resident(engine, c32, xm, ym, interp, prop)
            b = oracle_map_cube_mt(oracle, c32, xm, ym, (interp, interp) if isinstance(interp, int) else interp, prop)
            assert_close(a, b, ('f32', interp, prop))
    ci = np.nan_to_num(cube64[:7], nan=0.0, posinf=0.0, neginf=0.0)
    for dt in (np.int16, np.uint16, np.uint8, np.int32):
        c = (ci * 10).clip(np.iinfo(dt).min, np.iinfo(dt).max).astype(dt)
        a = map_resident(engine, c, xm, ym, 'cubic', True)
        assert_close(a, oracle_map_cube_mt(oracle, c, xm, ym, 'cubic', True), dt)


def test_smoothing_splines_on_a_batch_of_planes(engine, oracle, jupiter):
    """
    `spline_smoothing > 0` on 12 planes of 512^2 in mixed clean states (and 500 x 523): the knot search of every plane
    advances in lock-step on the GPU; every plane against the oracle (knots / coefficients = scipy's) to 1e-7 of the
    data scale, as in test_smoothing_splines_vs_oracle_kats_and_golden.
    """
    for ny, nx in ((512, 512), (500, 523)):
        cube, states = make_cube(12, ny, nx, seed=5 + nx)
        xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=2.0)
        npx = ny * nx
        for interp, s in (('cubic', 1.0 * npx), ('linear', 0.95 * npx), ((2, 3), 1.1 * npx), (5, 40.0 * npx)):
            for prop in (True, False):
                a = map_resident(engine, cube, xm, ym, interp, prop, spline_smoothing=s)
                b = oracle_map_cube_mt(oracle, cube, xm, ym, (interp, interp) if isinstance(interp, int) else interp, prop,
                                       spline_smoothing=s)
                assert np.array_equal(np.isnan(a), np.isnan(b)), (interp, s, prop)
                for p in range(cube.shape[0]):
                    fin = np.isfinite(b[p])
                    if states[p] == 'all_nan':
                        assert not fin.any()
                        continue
                    assert fin.sum() > 1000
                    # (planes with +-inf pixels: the cleaned image carries the plane median there, data scale as the rest)
                    assert np.max(np.abs(a[p][fin] - b[p][fin])) <= 1e-7 * max(1.0, np.abs(b[p][fin]).max()), (interp, s, prop, p, states[p])
    # a small s on nearly noise-free data: a long knot search (hundreds of knots per axis) before the fit reaches s.
    # (s far below the NOISE level is not a test case: FITPACK itself runs into numerically singular knot sets there
    #  and scipy returns coefficients of 1e99)
    cube, states = make_cube(8, 256, 256, seed=3, noise=0.004)
    xm, ym = setup_maps(engine, oracle, jupiter, 256, 256, deg=2.0)
    a = map_resident(engine, cube, xm, ym, 'cubic', True, spline_smoothing=1.0)
    b = oracle_map_cube_mt(oracle, cube, xm, ym, 'cubic', True, spline_smoothing=1.0)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    fin = np.isfinite(b)
    assert np.max(np.abs(a[fin] - b[fin])) <= 1e-7 * np.abs(b[fin]).max()


def test_a_smoothing_fit_in_a_batch_equals_the_plane_fitted_alone(engine, oracle, jupiter):
    """bit for bit: a plane's knot search must not depend on the planes that share its launches"""
    ny, nx = 500, 523
    cube, states = make_cube(12, ny, nx, seed=5 + nx)
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=2.0)
    for interp, s in (('cubic', 1.0 * ny * nx), ((2, 3), 1.1 * ny * nx)):
        together = map_resident(engine, cube, xm, ym, interp, True, spline_smoothing=s)
        assert np.array_equal(together, map_resident(engine, cube, xm, ym, interp, True, spline_smoothing=s), equal_nan=True)
        for p in (0, 2, 3, 4, 5, 11):
            one = map_resident(engine, cube[p : p + 1], xm, ym, interp, True, spline_smoothing=s)[0]
            assert np.array_equal(one, together[p], equal_nan=True), (interp, p, states[p])


def smoothing_fuzz(engine, oracle, jupiter, seed, n_cases=8):
    """
    Random small frames, degrees (each axis 1-5), smoothing factors at and above the noise level, planes in mixed clean
    states, both NaN policies: every plane of a batch against the oracle (1e-7 of the data scale, NaN masks identical)
    and one plane of the batch against itself fitted alone (bit for bit). (`s` far below the noise is not drawn: FITPACK
    itself runs into singular knot sets there.) tests/soak_fuzz.py --only smoothing runs it over fresh seeds.
    The one way a plane may miss the bar (1 fit in ~9 600 of the round-5 soak, seed 111241): fpknot choosing between
    intervals whose residual shares are equal in exact arithmetic - a choice scipy makes by the last bit of a residual sum
    (tests/test_smoothing_knife_edge.py: a one-ulp change of one pixel flips it). The library reports such searches
    (PM_OPT_LAST_SM_KNIFE_EDGES); a plane beyond the bar is accepted only in a call that reported one, only within 5 % of
    scale (both outcomes are smoothing splines of the same data and s), and is returned to the caller, who counts them.
    The other: a fit beyond what semi-normal equations + refinement resolve (20-25 sample axes of degree 4-5 with p ~ 1e8 or
    a rank-deficient least-squares phase: 3 fits in 191 000 of the round-6 soak, 4e-6 .. 9e-5 of scale) - reported as
    well (PM_OPT_LAST_SM_ILL_CONDITIONED), accepted only where reported and only within 1e-3.
    """
    from planetmapper_amd import _lib

    knife_edges = []
    smoothing_fuzz.calls = getattr(smoothing_fuzz, 'calls', 0)      # (running totals for tests/soak_fuzz.py: how often the
    smoothing_fuzz.flagged = getattr(smoothing_fuzz, 'flagged', 0)  #  library reports a knife edge at all)
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        ny, nx = int(rng.integers(16, 90)), int(rng.integers(16, 90))
        ky, kx = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        n_planes = int(rng.integers(2, 7))
        yy, xx = np.mgrid[0:ny, 0:nx]
        sigma = float(rng.choice([0.3, 1.0, 4.0]))
        cube = np.empty((n_planes, ny, nx))
        for p in range(n_planes):
            cube[p] = np.sin(xx / rng.uniform(3, 15)) * np.cos(yy / rng.uniform(3, 15)) * rng.uniform(1, 20) + sigma * rng.standard_normal((ny, nx))
        flavour = int(rng.integers(0, 4))
        if flavour == 1:
            cube[0][rng.random((ny, nx)) < 0.03] = np.nan
        elif flavour == 2 and n_planes > 1:
            cube[1][ny // 3 : ny // 3 + 4, nx // 4 : nx // 4 + 5] = np.nan
            cube[0][rng.random((ny, nx)) < 0.01] = np.inf
        elif flavour == 3:
            cube[n_planes - 1][:] = np.nan
        x0, y0 = float(rng.uniform(0.35, 0.65) * nx), float(rng.uniform(0.35, 0.65) * ny)
        r0 = float(rng.uniform(0.25, 0.6) * min(nx, ny))
        rot = float(rng.uniform(0, 6.28))
        engine.set_geometry(jupiter)
        engine.set_disc(x0, y0, r0, rot, nx, ny, True)
        lon, lat = oracle.rectangular_grid(jupiter, float(rng.choice([5.0, 9.0, 15.0])))
        xm, ym = engine.xy_map(lon, lat)
        if np.isfinite(xm).sum() < 20:
            continue
        s = float(rng.uniform(0.8, 3.0)) * ny * nx * sigma * sigma
        label = (seed, case, ny, nx, (ky, kx), s)
        for prop in (True, False):
            a = engine.map_cube(cube, xm, ym, (ky, kx), prop, spline_smoothing=s)
            flagged = engine.get_option(_lib.PM_OPT_LAST_SM_KNIFE_EDGES)
            ill = engine.get_option(_lib.PM_OPT_LAST_SM_ILL_CONDITIONED)
            smoothing_fuzz.calls += n_planes
            smoothing_fuzz.flagged += flagged
            smoothing_fuzz.ill = getattr(smoothing_fuzz, 'ill', 0) + ill
            b = oracle.map_cube(cube, xm, ym, (ky, kx), prop, spline_smoothing=s)
            assert np.array_equal(np.isnan(a), np.isnan(b)), label
            for p in range(n_planes):
                fin = np.isfinite(b[p])
                if fin.any():
                    scale = max(1.0, float(np.abs(cube[p][np.isfinite(cube[p])]).max()))
                    dev = float(np.max(np.abs(a[p][fin] - b[p][fin]))) / scale
                    if dev > 1e-7 and flagged > 0 and dev <= 0.05:
                        knife_edges.append(label + (p, prop, dev))
                        continue
                    if dev > 1e-7 and ill > 0 and dev <= 1e-3:
                        knife_edges.append(label + (p, prop, dev, 'ill-conditioned fit reported'))
                        continue
                    assert dev <= 1e-7, label + (p, prop, dev, f'knife edges reported: {flagged}, ill-conditioned fits: {ill}')
        p = int(rng.integers(0, n_planes))
        assert np.array_equal(engine.map_cube(cube[p : p + 1], xm, ym, (ky, kx), True, spline_smoothing=s)[0],
                              engine.map_cube(cube, xm, ym, (ky, kx), True, spline_smoothing=s)[p], equal_nan=True), label + (p,)
    return knife_edges


@pytest.mark.parametrize('seed', [20261004, 5])
def test_random_smoothing_spline_fuzz(engine, oracle, jupiter, seed):
    assert smoothing_fuzz(engine, oracle, jupiter, seed) == []  # (fixed seeds: no plane of these is at a knife edge)


def test_ill_conditioned_smoothing_fits_are_reported(engine, oracle, jupiter):
    """
    The three planes of the round-6 soak (191 000 fits) that miss the 1e-7 bar without a knife edge: 20-25 sample axes of
    degree 4-5 whose fits are ill-conditioned in themselves (scipy's own answer on them moves by 2e-5 .. 2.5e-4 of scale under a
    one-ulp change of one pixel: tests/test_smoothing_knife_edge.py). The library must SAY so (PM_OPT_LAST_SM_ILL_CONDITIONED: the refinement step's own measure) - the fuzz then
    accepts them within 1e-3 of scale and returns them; every other plane of those cubes meets the bar as everywhere.
    """
    for seed, n_cases, planes in ((201558, 1, {0}), (201514, 1, {0}), (201498, 5, {2})):
        tolerated = smoothing_fuzz(engine, oracle, jupiter, seed, n_cases=n_cases)
        assert tolerated and all(t[-1] == 'ill-conditioned fit reported' and 1e-7 < t[-2] <= 1e-3 for t in tolerated), (seed, tolerated)
        assert {t[6] for t in tolerated} == planes and {t[1] for t in tolerated} == {n_cases - 1}, (seed, tolerated)


def test_smoothing_knife_edge_is_reported_and_lands_on_one_of_scipys_two_answers(engine, oracle, jupiter):
    """
    The plane of tests/golden/smoothing_knife_edge.npz (round-5 soak, seed 111241 case 3): scipy puts a knot at row 61, and at
    row 51 once ONE pixel of the plane moves by ONE ulp (tests/test_smoothing_knife_edge.py) - two smoothing splines 7e-3 of
    scale apart, chosen between by the last bit of a residual sum. The device must (a) say so, and (b) return one of the two
    to the 1e-7 of every other smoothing fit; planes whose search meets no such tie report nothing.
    """
    import os

    from scipy.interpolate import RectBivariateSpline

    from planetmapper_amd import _lib

    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'smoothing_knife_edge.npz'))
    plane, (ky, kx), s = fx['plane'], (int(fx['degrees'][0]), int(fx['degrees'][1])), float(fx['s'])
    ny, nx = plane.shape
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=5.0)
    fin = np.isfinite(xm)
    a = engine.map_cube(plane[None], xm, ym, (ky, kx), True, spline_smoothing=s)[0]
    assert engine.get_option(_lib.PM_OPT_LAST_SM_KNIFE_EDGES) == 1
    scale = float(np.abs(plane).max())
    answers = []
    for flip in [None] + [tuple(f) for f in fx['flips'][:1]]:
        z = plane.copy()
        if flip is not None:
            z[flip[0], flip[1]] = np.nextafter(z[flip[0], flip[1]], np.inf * flip[2])
        sp = RectBivariateSpline(np.arange(ny), np.arange(nx), z, kx=ky, ky=kx, s=s)
        answers.append(sp.ev(ym[fin], xm[fin]))
    devs = [float(np.max(np.abs(a[fin] - b))) / scale for b in answers]
    apart = float(np.max(np.abs(answers[0] - answers[1]))) / scale
    print(f'\n[knife edge] scipy / scipy after a one-ulp change of one pixel: {apart:.1e} of scale apart; device - each: {devs[0]:.1e}, {devs[1]:.1e}')
    assert apart > 1e-3
    assert min(devs) <= 1e-7, devs
    # the drop-in surface says so too (a RuntimeWarning from map_img), and stays quiet otherwise
    import warnings

    from planetmapper_amd import BodyXY

    body = BodyXY('jupiter', geometry=jupiter, nx=nx, ny=ny)
    body.set_disc_params((nx - 1) / 2, (ny - 1) / 2, 0.45 * min(nx, ny), np.rad2deg(0.3))
    with pytest.warns(RuntimeWarning, match='1 plane.s. whose knot search chose between intervals tied to rounding'):
        body.map_img(plane, interpolation=(ky, kx), spline_smoothing=s, degree_interval=5)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        body.map_img(plane, interpolation=(ky, kx), spline_smoothing=40 * s, degree_interval=5)
    # a search without such a tie reports none (and meets the oracle as everywhere)
    cube, _ = make_cube(3, 300, 260, seed=9)
    xm2, ym2 = setup_maps(engine, oracle, jupiter, 300, 260, deg=3.0)
    a2 = engine.map_cube(cube[:1], xm2, ym2, 'cubic', True, spline_smoothing=300.0 * 260)
    assert engine.get_option(_lib.PM_OPT_LAST_SM_KNIFE_EDGES) == 0
    assert_close(a2, oracle.map_cube(cube[:1], xm2, ym2, 'cubic', True, spline_smoothing=300.0 * 260), 'no tie', bar=1e-7, scale_floor=float(np.abs(cube[0]).max()))


def test_smoothing_splines_on_axes_too_long_for_lds(engine, oracle, jupiter):
    """
    An axis of 3300 samples: its row tables (3300 x 7 x 8 B) do not fit the 160 KB of LDS, its knot arrays not the 64 KB
    `k_smb_decide` asks for - the sweeps, the factor pipeline and the knot insertion then read the tables where the
    tables kernel left them (the `lds = 0` variants), while the short axis (520) keeps the LDS ones. Degrees (5, 3) and
    (2, 4), both orientations, against the oracle as everywhere else (1e-7 of the data scale).
    """
    for ny, nx in ((3300, 520), (500, 3290)):
        cube, states = make_cube(3, ny, nx, seed=ny)
        xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=3.0)
        for interp, s in (((5, 3), 1.0 * ny * nx), ((2, 4), 1.05 * ny * nx)):
            a = map_resident(engine, cube, xm, ym, interp, True, spline_smoothing=s)
            b = oracle_map_cube_mt(oracle, cube, xm, ym, interp, True, spline_smoothing=s)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (ny, nx, interp)
            for p in range(3):
                fin = np.isfinite(b[p])
                assert fin.sum() > 100
                assert np.max(np.abs(a[p][fin] - b[p][fin])) <= 1e-7 * max(1.0, np.abs(b[p][fin]).max()), (ny, nx, interp, p, states[p])
        # and the interpolating splines at that size (tiled banded solves with long lines)
        a = map_resident(engine, cube, xm, ym, 'cubic', False)
        assert_close(a, oracle_map_cube_mt(oracle, cube, xm, ym, 'cubic', False), (ny, nx, 'cubic'))


def test_smoothing_batches_capped_by_the_option_give_the_same_planes(engine, oracle, jupiter):
    """PM_OPT_SM_BATCH_PLANES caps the planes fitted together: a cube then takes several batches - the same bits as one"""
    from planetmapper_amd import _lib

    ny, nx = 300, 260
    cube, states = make_cube(7, ny, nx, seed=9)
    xm, ym = setup_maps(engine, oracle, jupiter, ny, nx, deg=3.0)
    s = 1.0 * ny * nx
    whole = map_resident(engine, cube, xm, ym, 'cubic', True, spline_smoothing=s)
    try:
        for cap in (1, 3):
            engine.set_option(_lib.PM_OPT_SM_BATCH_PLANES, cap)
            assert engine.get_option(_lib.PM_OPT_SM_BATCH_PLANES) == cap
            assert np.array_equal(map_resident(engine, cube, xm, ym, 'cubic', True, spline_smoothing=s), whole, equal_nan=True), cap
            assert np.array_equal(engine.map_cube(cube, xm, ym, 'cubic', True, spline_smoothing=s), whole, equal_nan=True), cap
    finally:
        engine.set_option(_lib.PM_OPT_SM_BATCH_PLANES, 0)
