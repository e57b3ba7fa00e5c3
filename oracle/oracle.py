"""
ctypes wrapper around the CPU oracle (oracle/pm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py. Nothing under planetmapper_amd/ imports this module.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from planetmapper_amd.geometry import PMDisc, PMGeometry

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libpm_oracle.so')
NUM_PLANES = 26

PLANE_NAMES = [
    'LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC', 'LAT-CENTRIC', 'RA', 'DEC',
    'PIXEL-X', 'PIXEL-Y', 'KM-X', 'KM-Y', 'ANGULAR-X', 'ANGULAR-Y',
    'PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH', 'LOCAL-SOLAR-TIME',
    'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER',
    'LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC',
    'RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE',
]  # fmt: skip
PLANE_INDEX = {n: i for i, n in enumerate(PLANE_NAMES)}

DTYPES = {
    np.dtype('float64'): 0,
    np.dtype('float32'): 1,
    np.dtype('int16'): 2,
    np.dtype('int32'): 3,
    np.dtype('uint8'): 4,
    np.dtype('uint16'): 5,
}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile) if needed; return the .so path."""
    src = os.path.join(_HERE, 'pm_oracle.c')
    hdr = os.path.join(_HERE, '..', 'include', 'planetmapper_hip.h')
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH)
        for p in (src, hdr)
    )
    if force or stale:
        subprocess.run(['make', '-C', _HERE, '-B', 'all'], check=True,
                       capture_output=True)
    return _LIB_PATH


_lib = None


def use_variant(name: str | None) -> None:
    """
    Switch the module to another build of the same source (`'fma'` ->
    libpm_oracle_fma.so, None -> the strict build). Only tests/test_noise_floor.py
    uses this, to measure the rounding-noise floor of the formulation.
    """
    global _lib, _LIB_PATH
    _LIB_PATH = os.path.join(_HERE, 'libpm_oracle.so' if name is None else f'libpm_oracle_{name}.so')
    _lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or _LIB_PATH.endswith('libpm_oracle.so'):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        dpp = ctypes.POINTER(ctypes.POINTER(ctypes.c_double))
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.pmo_backplanes_img.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.POINTER(PMDisc), ctypes.c_double,
            ctypes.c_uint64, dpp,
        ]
        _lib.pmo_backplanes_map.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.POINTER(PMDisc), ctypes.c_double,
            ctypes.c_uint64, dp, dp, ctypes.c_int, ctypes.c_int, dpp,
        ]
        _lib.pmo_map_cube.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp,
        ]
        _lib.pmo_rectangular_grid.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.c_double, ctypes.c_int, ctypes.c_int, dp, dp,
        ]
        _lib.pmo_radec_query.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.c_double, ctypes.c_int, dp, dp, ctypes.c_int, dp,
        ]
        _lib.pmo_map_cube_spline.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp,
        ]
        _lib.pmo_map_cube_smooth.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp,
        ]
        _lib.pmo_pchip.argtypes = [dp, dp, ctypes.c_int, dp, ctypes.c_int, dp]
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.pmo_regrid_smooth.argtypes = [
            dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, ip, dp, ip, dp,
        ]
        _lib.pmo_map_cube_spline_smooth.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, dp,
        ]
        _lib.pmo_clean_nans.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
        _lib.pmo_transform.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.POINTER(PMDisc), ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
            dp, dp, ctypes.c_double, ctypes.c_int, dp, dp,
        ]
        assert _lib.pmo_sizeof_geometry() == ctypes.sizeof(PMGeometry)
        assert _lib.pmo_sizeof_disc() == ctypes.sizeof(PMDisc)
    return _lib


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def make_disc(x0, y0, r0, rotation_deg, nx, ny, optimize_speed=True) -> PMDisc:
    d = PMDisc()
    d.x0, d.y0, d.r0 = float(x0), float(y0), float(r0)
    # BodyXY.set_rotation -> _set_rotation_radians: body_xy.py:867-890
    d.rotation_rad = float(np.deg2rad(rotation_deg) % (2 * np.pi))
    d.nx, d.ny = int(nx), int(ny)
    d.optimize_speed = 1 if optimize_speed else 0
    return d


def mask_of(names) -> int:
    m = 0
    for n in names:
        m |= 1 << (PLANE_INDEX[n] if isinstance(n, str) else int(n))
    return m


def backplanes_img(g: PMGeometry, d: PMDisc, names, alt: float = 0.0) -> dict[str, np.ndarray]:
    names = list(names)
    outs = {n: np.empty((d.ny, d.nx), dtype=np.float64) for n in names}
    ptrs = (ctypes.POINTER(ctypes.c_double) * NUM_PLANES)()
    for n, a in outs.items():
        ptrs[PLANE_INDEX[n]] = _dptr(a)
    rc = lib().pmo_backplanes_img(ctypes.byref(g), ctypes.byref(d), float(alt), mask_of(names), ptrs)
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return outs


_quad = None


def quad_lib() -> ctypes.CDLL:
    """libpm_oracle_quad.so: pm_oracle.c compiled in IEEE binary128 (oracle/pm_oracle_quad.c)."""
    global _quad
    if _quad is None:
        path = os.path.join(_HERE, 'libpm_oracle_quad.so')
        src = [os.path.join(_HERE, f) for f in ('pm_oracle_quad.c', 'pm_oracle.c')]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in src):
            subprocess.run(['make', '-C', _HERE, 'libpm_oracle_quad.so'], check=True, capture_output=True)
        _quad = ctypes.CDLL(path)
        _quad.pmoq_backplanes_img_rows.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.c_int, ctypes.POINTER(PMDisc), ctypes.c_double, ctypes.c_uint64,
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)),
        ]  # fmt: skip
        dp = ctypes.POINTER(ctypes.c_double)
        _quad.pmoq_backplanes_map.argtypes = [
            ctypes.POINTER(PMGeometry), ctypes.c_int, ctypes.POINTER(PMDisc), ctypes.c_double, ctypes.c_uint64, dp, dp,
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(dp),
        ]  # fmt: skip
    return _quad


def backplanes_map_quad(g: PMGeometry, d: PMDisc, names, lon_deg, lat_deg, alt: float = 0.0) -> dict[str, np.ndarray]:
    """Map-space planes of a lon/lat grid evaluated in binary128 and rounded to binary64 (the truth)."""
    names = list(names)
    lon = np.ascontiguousarray(lon_deg, dtype=np.float64)
    lat = np.ascontiguousarray(lat_deg, dtype=np.float64)
    n0, n1 = lon.shape
    outs = {n: np.empty((n0, n1), dtype=np.float64) for n in names}
    ptrs = (ctypes.POINTER(ctypes.c_double) * NUM_PLANES)()
    for n, a in outs.items():
        ptrs[PLANE_INDEX[n]] = _dptr(a)
    rc = quad_lib().pmoq_backplanes_map(ctypes.byref(g), ctypes.sizeof(PMGeometry), ctypes.byref(d), float(alt),
                                        mask_of(names), _dptr(lon), _dptr(lat), n0, n1, ptrs)  # fmt: skip
    if rc != 0:
        raise ValueError(f'quad oracle error {rc}')
    return outs


def backplanes_img_rows_quad(g: PMGeometry, d: PMDisc, names, row_begin: int, n_rows: int, alt: float = 0.0,
                             threads: int | None = None) -> dict[str, np.ndarray]:
    """
    Image rows [row_begin, row_begin + n_rows) of the named planes evaluated in binary128 and rounded
    to binary64: the exact value of the oracle's formulation on the same binary64 inputs (the truth the
    parity bars are measured against, tests/test_truth_f128.py). ~60x slower than the binary64 oracle.
    """
    names = list(names)
    q = quad_lib()
    if threads:
        q.pmoq_set_num_threads(int(threads))
    outs = {n: np.empty((int(n_rows), d.nx), dtype=np.float64) for n in names}
    ptrs = (ctypes.POINTER(ctypes.c_double) * NUM_PLANES)()
    for n, a in outs.items():
        ptrs[PLANE_INDEX[n]] = _dptr(a)
    rc = q.pmoq_backplanes_img_rows(ctypes.byref(g), ctypes.sizeof(PMGeometry), ctypes.byref(d), float(alt),
                                    mask_of(names), int(row_begin), int(n_rows), ptrs)  # fmt: skip
    if rc != 0:
        raise ValueError(f'quad oracle error {rc}')
    return outs


def backplanes_map(g, d, names, lon_deg, lat_deg, alt: float = 0.0) -> dict[str, np.ndarray]:
    names = list(names)
    lon = np.ascontiguousarray(lon_deg, dtype=np.float64)
    lat = np.ascontiguousarray(lat_deg, dtype=np.float64)
    n0, n1 = lon.shape
    outs = {n: np.empty((n0, n1), dtype=np.float64) for n in names}
    ptrs = (ctypes.POINTER(ctypes.c_double) * NUM_PLANES)()
    for n, a in outs.items():
        ptrs[PLANE_INDEX[n]] = _dptr(a)
    rc = lib().pmo_backplanes_map(
        ctypes.byref(g), ctypes.byref(d), float(alt), mask_of(names), _dptr(lon), _dptr(lat),
        n0, n1, ptrs,
    )
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return outs


def xy_map(g, d, lon_deg, lat_deg, alt: float = 0.0):
    o = backplanes_map(g, d, ['PIXEL-X', 'PIXEL-Y'], lon_deg, lat_deg, alt)
    return o['PIXEL-X'], o['PIXEL-Y']


def rectangular_grid(g: PMGeometry, degree_interval: float):
    """lon/lat grids of BodyXY.generate_map_coordinates('rectangular') body_xy.py:2899."""
    n1 = len(np.arange(degree_interval / 2, 360, degree_interval))
    n0 = len(np.arange(-90 + degree_interval / 2, 90, degree_interval))
    lon = np.empty((n0, n1))
    lat = np.empty((n0, n1))
    lib().pmo_rectangular_grid(ctypes.byref(g), float(degree_interval), n0, n1, _dptr(lon), _dptr(lat))
    return lon, lat


def regrid_smooth(z, kx: int, ky: int, s: float):
    """FITPACK regrid smoothing spline of z on (arange, arange): (tx, ty, c) like
    scipy.interpolate.RectBivariateSpline(..., s=s).get_knots() / get_coeffs()"""
    z = np.ascontiguousarray(z, dtype=np.float64)
    mx, my = z.shape
    tx, ty = np.empty(mx + kx + 1), np.empty(my + ky + 1)
    c = np.empty((mx + 1) * (my + 1))
    nx, ny = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().pmo_regrid_smooth(_dptr(z), mx, my, kx, ky, float(s), _dptr(tx), ctypes.byref(nx), _dptr(ty),
                                 ctypes.byref(ny), _dptr(c))
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    tx, ty = tx[: nx.value].copy(), ty[: ny.value].copy()
    return tx, ty, c[: (nx.value - kx - 1) * (ny.value - ky - 1)].reshape(nx.value - kx - 1, ny.value - ky - 1).copy()


def map_cube(cube: np.ndarray, x_map, y_map, interpolation='linear', propagate_nan=True,
             smooth_oversample_by=5, smooth_max_oversampled_img_size=10_000, spline_smoothing=0.0):
    cube = np.ascontiguousarray(cube)
    if cube.ndim == 2:
        cube = cube[None]
    p, ny, nx = cube.shape
    xm = np.ascontiguousarray(x_map, dtype=np.float64)
    ym = np.ascontiguousarray(y_map, dtype=np.float64)
    n0, n1 = xm.shape
    out = np.empty((p, n0, n1), dtype=np.float64)
    if interpolation == 'smooth':
        rc = lib().pmo_map_cube_smooth(
            cube.ctypes.data_as(ctypes.c_void_p), DTYPES[cube.dtype], p, ny, nx, _dptr(xm), _dptr(ym), n0, n1,
            int(smooth_oversample_by), int(smooth_max_oversampled_img_size), 1 if propagate_nan else 0, _dptr(out),
        )
        if rc != 0:
            raise ValueError(f'oracle error {rc}')
        return out
    spline = {'quadratic': (2, 2), 'cubic': (3, 3), 2: (2, 2), 3: (3, 3)}.get(interpolation)
    if isinstance(interpolation, tuple):
        spline = interpolation
    if spline_smoothing and interpolation != 'nearest':
        if spline is None:
            spline = (1, 1)
        rc = lib().pmo_map_cube_spline_smooth(
            cube.ctypes.data_as(ctypes.c_void_p), DTYPES[cube.dtype], p, ny, nx, _dptr(xm), _dptr(ym), n0, n1,
            int(spline[0]), int(spline[1]), float(spline_smoothing), 1 if propagate_nan else 0, _dptr(out),
        )
        if rc != 0:
            raise ValueError(f'oracle error {rc}')
        return out
    if spline is not None and spline != (1, 1):
        rc = lib().pmo_map_cube_spline(
            cube.ctypes.data_as(ctypes.c_void_p), DTYPES[cube.dtype], p, ny, nx, _dptr(xm), _dptr(ym), n0, n1,
            int(spline[0]), int(spline[1]), 1 if propagate_nan else 0, _dptr(out),
        )
        if rc != 0:
            raise ValueError(f'oracle error {rc}')
        return out
    interp = {'nearest': 0, 'linear': 1, 1: 1, (1, 1): 1}[interpolation]
    rc = lib().pmo_map_cube(
        cube.ctypes.data_as(ctypes.c_void_p), DTYPES[cube.dtype], p, ny, nx, _dptr(xm), _dptr(ym),
        n0, n1, interp, 1 if propagate_nan else 0, _dptr(out),
    )
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return out


def clean_nans(img) -> np.ndarray:
    """BodyXY._replace_nans_with_interpolated_values (body_xy.py:1871-1904)"""
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.empty_like(img)
    rc = lib().pmo_clean_nans(_dptr(img), img.shape[0], img.shape[1], _dptr(out))
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return out


def pchip(x, y, xq) -> np.ndarray:
    """scipy.interpolate.PchipInterpolator(x, y, extrapolate=False)(xq) as restated in the oracle"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    xq = np.ascontiguousarray(xq, dtype=np.float64)
    out = np.empty_like(xq)
    rc = lib().pmo_pchip(_dptr(x), _dptr(y), len(x), _dptr(xq), len(xq), _dptr(out))
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return out


def radec_query(g, ra_deg, dec_deg, alt: float = 0.0, ring_only_visible: bool = True) -> np.ndarray:
    """Rows of (lon, lat, ring radius, ring lon, ring dist, limb lon, limb lat, limb dist)."""
    ra = np.ascontiguousarray(np.atleast_1d(ra_deg), dtype=np.float64)
    dec = np.ascontiguousarray(np.atleast_1d(dec_deg), dtype=np.float64)
    out = np.empty((len(ra), 8), dtype=np.float64)
    lib().pmo_radec_query(ctypes.byref(g), float(alt), len(ra), _dptr(ra), _dptr(dec),
                          1 if ring_only_visible else 0, _dptr(out))
    return out


def set_num_threads(n: int) -> int:
    """OpenMP threads of the oracle's row loops; returns the count in effect."""
    return int(lib().pmo_set_num_threads(int(n)))


COORDS = {'xy': 0, 'radec': 1, 'angular': 2, 'km': 3, 'lonlat': 4}


def transform(g, d, src: str, dst: str, a, b, alt: float = 0.0, not_visible_nan=False, planetocentric=False):
    """Array-valued coordinate transform (see pmo_transform)."""
    a, b = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    shape = a.shape
    a = np.ascontiguousarray(a).ravel()
    b = np.ascontiguousarray(b).ravel()
    oa = np.empty_like(a)
    ob = np.empty_like(b)
    flags = (1 if not_visible_nan else 0) | (2 if planetocentric else 0)
    rc = lib().pmo_transform(ctypes.byref(g), ctypes.byref(d), COORDS[src], COORDS[dst], a.size, _dptr(a), _dptr(b),
                             float(alt), flags, _dptr(oa), _dptr(ob))
    if rc != 0:
        raise ValueError(f'oracle error {rc}')
    return oa.reshape(shape), ob.reshape(shape)
