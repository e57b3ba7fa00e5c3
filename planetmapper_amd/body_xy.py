"""
`BodyXY`: the reference's image-plane / backplane API (`planetmapper/body_xy.py`) on top
of the HIP engine.

The method names, argument meanings, array shapes/dtypes, copy-vs-read-only-view rules,
cache invalidation rules and error behaviour follow the reference class so that code
written against `planetmapper.BodyXY` for the backplane / mapping path runs unchanged:

    body = BodyXY(scenario='jupiter_hst_2005', sz=500)
    body.set_disc_params(x0=250, y0=250, r0=200)
    lon = body.get_lon_img()                      # read-only (ny, nx) float64 view
    emi = body.get_backplane_img('EMISSION')      # fresh copy
    img_map = body.map_img(img, degree_interval=1)

Only the per-pixel numerics differ: every pixel loop of the reference is one fused
kernel launch here. What is NOT provided (plotting, GUI, wireframes, SPICE kernel
management, pyproj projections) is out of scope for this path.
"""

from __future__ import annotations

import datetime
import math
from typing import Any, Callable, Iterable, NamedTuple

import numpy as np

from . import _lib
from .engine import Engine
from .geometry import PMGeometry
from .scenarios import load_scenario


class Backplane(NamedTuple):
    """`planetmapper.Backplane` (body_xy.py:79-107)."""

    name: str
    description: str
    get_img: Callable[[], np.ndarray]
    get_map: Callable[..., np.ndarray]


class BackplaneNotFoundError(Exception):
    """body_xy.py:4359"""


class NotFoundError(Exception):
    """Stand-in for spiceypy's NotFoundError (raised when `not_found_nan=False`)."""


MAP_KWARG_KEYS = (
    'projection', 'degree_interval', 'lon', 'lat', 'size', 'lon_coords', 'lat_coords',
    'projection_x_coords', 'projection_y_coords', 'xlim', 'ylim', 'alt',
)  # MapKwargs body_xy.py:51-69  # fmt: skip


def _readonly(a: np.ndarray) -> np.ndarray:
    """Read-only view (`_as_readonly_view` base.py:115-121)."""
    v = a.view()
    v.flags.writeable = False
    return v


def _freeze(v: Any) -> Any:
    """Hashable stand-in for a kwarg value (`_replace_np_arrr_args_with_tuples` base.py:41)."""
    if isinstance(v, np.ndarray):
        return ('ndarray', v.shape, v.dtype.str, v.tobytes())
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    return v


_ENGINES: dict[int, Engine] = {}


def _shared_engine(device: int) -> Engine:
    """One HIP context per device per process, shared by every BodyXY on that device."""
    eng = _ENGINES.get(device)
    if eng is None or eng._ctx is None:
        eng = Engine(device)
        _ENGINES[device] = eng
    return eng


class BodyXY:
    """
    Image-plane view of a body observed at one epoch (reference `BodyXY`,
    body_xy.py:114).

    Args:
        target, utc, observer: kept for signature compatibility; used only for
            `repr`/metadata when `geometry`/`scenario` supplies the numbers.
        nx, ny / sz: image size in pixels (body_xy.py:186-199).
        geometry: a filled `PMGeometry` block (from `GeometryBuilder` or from a
            spiceypy-backed reference `Body`).
        scenario: name of a packaged scenario (`planetmapper_amd.scenarios`).
        kernels: directory searched recursively for `*.bsp` / `*.tpc` / `*.tls` kernels like
            the reference's `kernel_path` (`planetmapper_amd.kernels`; SPK types 2/3 only).
        optimize_speed: enables the radius pre-mask exactly like the reference
            (`SpiceBase(optimize_speed=True)`, base.py:229; body_xy.py:3201-3218).
        device: GPU index.
    """

    def __init__(
        self,
        target: str | None = None,
        utc: Any = None,
        observer: str | int = 'EARTH',
        nx: int = 0,
        ny: int = 0,
        *,
        sz: int | None = None,
        geometry: PMGeometry | None = None,
        scenario: str | None = None,
        kernels: str | None = None,
        optimize_speed: bool = True,
        device: int = 0,
        engine: Engine | None = None,
    ) -> None:
        if sz is not None:
            if nx != 0 or ny != 0:
                raise ValueError('`sz` cannot be used if `nx` and/or `ny` are nonzero')
            nx = sz
            ny = sz
        if geometry is None:
            if kernels is not None:
                from .kernels import geometry_from_kernels

                geometry = geometry_from_kernels(target, utc, observer, kernels)
            elif scenario is not None:
                geometry = load_scenario(scenario)
            else:
                raise ValueError(
                    'BodyXY needs `geometry=` (a PMGeometry block), `kernels=` (a directory of '
                    '*.bsp/*.tpc/*.tls files) or `scenario=` (see INTEGRATION.md)'
                )
        self._geometry = geometry.copy()
        self.target = None if target is None else str(target).strip().upper()
        self.observer = observer.strip().upper() if isinstance(observer, str) else observer
        # base.py:818-829: `utc` is normalised to a UTC string made from the epoch itself
        from .timeconv import et2utc

        try:
            self.dtm = et2utc(geometry.et).replace(tzinfo=datetime.timezone.utc)
            self.utc = self.dtm.strftime('%Y-%m-%dT%H:%M:%S.%f')
        except (ValueError, OverflowError):
            self.dtm = None
            self.utc = utc
        self._optimize_speed = bool(optimize_speed)
        self._engine = engine if engine is not None else _shared_engine(device)

        g = self._geometry
        # Body attributes used on this path (body.py:501-614, base.py:795-839)
        self.et = g.et
        self.radii = np.array(g.radii[:])
        self.r_eq = float(g.radii[0])
        self.r_polar = float(g.radii[2])
        self.flattening = (self.r_eq - self.r_polar) / self.r_eq
        self.target_light_time = g.lt_c
        self.target_distance = g.lt_c * g.clight
        self.target_diameter_arcsec = g.diameter_arcsec
        self.km_per_arcsec = g.km_per_arcsec
        self.positive_longitude_direction = 'W' if g.west_positive else 'E'
        self.subpoint_distance = g.sub_dist
        self._alt_adjustment = 0.0
        # metadata of Body (body.py:501-606) that ends up in saved FITS headers
        from .geometry import GeometryBuilder
        from .kernels import body_id

        d = GeometryBuilder.describe(g)
        self.target_ra = d['target_ra']
        self.target_dec = d['target_dec']
        self.subpoint_lon = d['subpoint_lon']
        self.subpoint_lat = d['subpoint_lat']
        self.subsol_lon = d['subsol_lon']
        self.subsol_lat = d['subsol_lat']
        try:
            self.target_body_id = body_id(self.target) if self.target is not None else None
        except (KeyError, ValueError):
            self.target_body_id = None
        self.target_frame = None if self.target is None else 'IAU_' + self.target
        self.observer_frame = 'J2000'
        self.illumination_source = 'SUN'
        self.aberration_correction = 'CN'
        self.subpoint_method = 'INTERCEPT/ELLIPSOID'
        self.surface_method = 'ELLIPSOID'

        self._cache: dict = {}  # cleared when the disc parameters change (base.py:58-88)
        self._stable_cache: dict = {}  # map-space results (base.py:91-112)
        self._nx = int(nx)
        self._ny = int(ny)
        self._x0 = 0.0
        self._y0 = 0.0
        self._r0 = 10.0
        self._rotation_radians = 0.0
        self._disc_method = 'default'
        self.backplanes: dict[str, Backplane] = {}
        self._register_default_backplanes()
        self.reset_disc_params()

    def north_pole_angle(self) -> float:
        """
        Angle of the north pole on the sky, degrees in (-180, 180], clockwise from +Dec
        (body.py:2985-3005): from the angular coordinates of the pole (lon 0, lat 90, at the
        current altitude adjustment) and of the target centre.
        """
        if self._alt_adjustment == 0.0:
            return float(np.rad2deg(self._geometry.np_angle_rad))
        np_x, np_y = self._transform('lonlat', 'angular', 0.0, 90.0, alt=self._alt_adjustment)
        t_x, t_y = self._transform('radec', 'angular', self.target_ra, self.target_dec)
        theta = float(np.rad2deg(-np.arctan2(t_x - np_x, np_y - t_y))) % 360.0
        return theta - 360.0 if theta > 180 else theta

    @property
    def geometry(self) -> PMGeometry:
        """A copy of the geometry block this body was built from (reusable for other instances)."""
        return self._geometry.copy()

    def __repr__(self) -> str:
        return f'BodyXY({self.target!r}, {self.utc!r}, observer={self.observer!r}, nx={self._nx}, ny={self._ny})'

    # ------------------------------------------------------------------ caches
    def _clear_cache(self) -> None:
        # image planes nobody outside this object refers to go back to the engine's pool of result arrays
        # (Engine.plane_buffer): the next cold getter writes into memory whose pages exist - and are page-locked
        recycle = getattr(self._engine, 'recycle_plane', None)
        buffers = self.__dict__.get('_img_buffers')
        if recycle is not None and buffers:
            for key in list(buffers):
                self._cache.pop(key, None)  # (the read-only view goes first: what is left is the array itself)
                arr = buffers.pop(key)
                recycle(arr)
                del arr
        # device-resident results (`device=True` getters): later accesses of the handles raise; their memory goes back to
        # the engine once no DLPack consumer holds it
        for key, val in self._cache.items():
            if isinstance(key, tuple) and key and key[0] in ('img_dev', 'map_dev', 'mapped_dev'):
                for arr in (val.values() if isinstance(val, dict) else (val if isinstance(val, tuple) else (val,))):
                    arr.invalidate()
        self._cache.clear()

    def _invalidate_disc_parameters(self) -> None:  # body_xy.py:696-698
        self._clear_cache()

    # ------------------------------------------------------------------ disc parameters
    def set_disc_params(self, x0=None, y0=None, r0=None, rotation=None) -> None:
        """body_xy.py:700-729"""
        if x0 is not None:
            self.set_x0(x0)
        if y0 is not None:
            self.set_y0(y0)
        if r0 is not None:
            self.set_r0(r0)
        if rotation is not None:
            self.set_rotation(rotation)

    def adjust_disc_params(self, dx=0, dy=0, dr=0, drotation=0) -> None:
        self.set_x0(self.get_x0() + dx)
        self.set_y0(self.get_y0() + dy)
        self.set_r0(self.get_r0() + dr)
        self.set_rotation(self.get_rotation() + drotation)

    def get_disc_params(self) -> tuple[float, float, float, float]:
        return self.get_x0(), self.get_y0(), self.get_r0(), self.get_rotation()

    def reset_disc_params(self) -> str:
        """body_xy.py:772-789"""
        self.set_rotation(0.0)
        if self._test_if_img_size_valid():
            self.centre_disc()
        else:
            self.set_disc_params(x0=0, y0=0, r0=10)
            self.set_disc_method('zero')
        return self.get_disc_method()

    def centre_disc(self) -> None:
        """body_xy.py:791-803"""
        self.set_x0((self._nx - 1) / 2)
        self.set_y0((self._ny - 1) / 2)
        self.set_r0(0.9 * (min(self.get_x0(), self.get_y0())))
        self.set_disc_method('centre_disc')

    def set_x0(self, x0: float) -> None:
        if not math.isfinite(x0):
            raise ValueError('x0 must be finite')
        self._x0 = float(x0)
        self._invalidate_disc_parameters()

    def get_x0(self) -> float:
        return self._x0

    def set_y0(self, y0: float) -> None:
        if not math.isfinite(y0):
            raise ValueError('y0 must be finite')
        self._y0 = float(y0)
        self._invalidate_disc_parameters()

    def get_y0(self) -> float:
        return self._y0

    def set_r0(self, r0: float) -> None:
        if not math.isfinite(r0):
            raise ValueError('r0 must be finite')
        if not r0 > 0:
            raise ValueError('r0 must be greater than zero')
        self._r0 = float(r0)
        self._invalidate_disc_parameters()

    def get_r0(self) -> float:
        return self._r0

    def set_rotation(self, rotation: float) -> None:
        """Degrees; stored in radians modulo 2 pi (body_xy.py:867-890)."""
        if not math.isfinite(rotation):
            raise ValueError('rotation must be finite')
        self._rotation_radians = float(np.deg2rad(rotation) % (2 * np.pi))
        self._invalidate_disc_parameters()

    def get_rotation(self) -> float:
        return float(np.rad2deg(self._rotation_radians))

    def set_plate_scale_arcsec(self, arcsec_per_px: float) -> None:
        self.set_r0(self.target_diameter_arcsec / (2 * arcsec_per_px))

    def set_plate_scale_km(self, km_per_px: float) -> None:
        self.set_plate_scale_arcsec(km_per_px / self.km_per_arcsec)

    def get_plate_scale_arcsec(self) -> float:
        return self.target_diameter_arcsec / (2 * self.get_r0())

    def get_plate_scale_km(self) -> float:
        return self.get_plate_scale_arcsec() * self.km_per_arcsec

    def set_img_size(self, nx: int | None = None, ny: int | None = None) -> None:
        """body_xy.py:941-961"""
        nx = self._nx if nx is None else int(nx)
        ny = self._ny if ny is None else int(ny)
        if nx < 0 or ny < 0:
            raise ValueError('nx and ny must be non-negative')
        self._nx = nx
        self._ny = ny
        self._clear_cache()

    def get_img_size(self) -> tuple[int, int]:
        return (self._nx, self._ny)

    def rotate_north_to_top(self) -> None:
        """body_xy.py:892-902"""
        self.set_rotation(-self.north_pole_angle())
        self.set_disc_method('rotate_north_to_top')

    def _reframe(self, size, zoom: float, shift: float) -> None:
        """New frame size with the disc kept on the same piece of sky: centre -> centre * zoom + shift."""
        self.set_img_size(*size)
        if zoom != 1.0:
            self.set_r0(self.get_r0() * zoom)
        for get, put in ((self.get_x0, self.set_x0), (self.get_y0, self.set_y0)):
            put(get() * zoom + shift)

    def scale_img_size(self, factor: float, *, allow_rounding: bool = False) -> None:
        """body_xy.py:973-1023: multiply nx, ny by `factor`, keeping the disc where it is"""
        if factor <= 0:
            raise ValueError('Scaling factor must be greater than zero')
        old = self.get_img_size()
        exact = tuple(n * factor for n in old)
        whole = tuple(math.ceil(v) for v in exact)
        if whole != exact and not allow_rounding:
            raise ValueError(
                f'Image size {old} cannot be exactly scaled by {factor} to an integer number of pixels: new size '
                f'would be {exact}. Use `allow_rounding=True` to allow rounding of the image size.'
            )
        # pixel edges sit at -0.5 and n - 0.5: a zoom about the lower left EDGE moves a centre c to
        # c * factor + (factor - 1) / 2
        self._reframe(whole, factor, (factor - 1) / 2)

    def add_img_border(self, border: int) -> None:
        """body_xy.py:1025-1058: grow (or crop, if negative) the frame by `border` pixels per side"""
        pad = int(border)
        self._reframe(tuple(n + 2 * pad for n in self.get_img_size()), 1.0, pad)

    def add_arcsec_offset(self, dra_arcsec: float = 0, ddec_arcsec: float = 0) -> None:
        """body_xy.py:1088-1103: shift (x0, y0) by an offset given in RA/Dec arcseconds"""
        origin = self.xy2radec(0, 0)
        moved = self.radec2xy(origin[0] + dra_arcsec / 3600, origin[1] + ddec_arcsec / 3600)
        self.adjust_disc_params(dx=moved[0], dy=moved[1])

    def _get_img_limits(self, func):
        """body_xy.py:1106-1120: extremes of `func` over the four outer pixel corners"""
        edges_x, edges_y = (-0.5, self._nx - 0.5), (-0.5, self._ny - 0.5)
        u, v = zip(*(func(x, y) for x in edges_x for y in edges_y))
        return (min(u), max(u)), (min(v), max(v))

    def get_img_limits_radec(self):
        (ra_lo, ra_hi), dec = self._get_img_limits(self.xy2radec)
        return (ra_hi, ra_lo), dec  # RA increases to the left

    def get_img_limits_km(self):
        return self._get_img_limits(self.xy2km)

    def get_img_limits_angular(self):
        return self._get_img_limits(self.xy2angular)

    def get_img_limits_xy(self):
        return self._get_img_limits(lambda x, y: (x, y))

    def set_disc_method(self, method: str) -> None:
        self._disc_method = method

    def get_disc_method(self) -> str:
        return self._disc_method

    def _test_if_img_size_valid(self) -> bool:
        return (self._nx > 0) and (self._ny > 0)

    # ------------------------------------------------------------------ engine plumbing
    def _bind(self) -> Engine:
        """Point the (shared) context at this body's geometry and disc."""
        eng = self._engine
        eng.set_geometry(self._geometry)
        eng.set_disc(
            self._x0, self._y0, self._r0, self._rotation_radians, self._nx, self._ny, self._optimize_speed
        )
        return eng

    def _img_planes(self, names: Iterable[str]) -> dict[str, np.ndarray]:
        """
        Cached image-space planes; every missing one is produced by a single fused
        launch. Cache key includes the altitude adjustment like
        `_cache_clearable_alt_dependent_result` (body.py:255-272).
        """
        names = list(names)
        alt = self._alt_adjustment
        missing = [n for n in names if ('img', n, alt) not in self._cache]
        if missing:
            if not self._test_if_img_size_valid():
                # BodyXY._make_empty_img body_xy.py:3166-3168
                raise ValueError('nx and ny must be positive to create a backplane image')
            eng = self._bind()
            if hasattr(eng, 'plane_buffer'):
                out = eng.backplanes_img(missing, alt=alt, recycled=True)
                self.__dict__.setdefault('_img_buffers', {}).update({('img', n, alt): a for n, a in out.items()})
            else:
                out = eng.backplanes_img(missing, alt=alt)
            for n, a in out.items():
                self._cache[('img', n, alt)] = _readonly(a)
            del out, a
        return {n: self._cache[('img', n, alt)] for n in names}

    def _img_planes_device(self, names: Iterable[str]) -> dict:
        """
        The device form of `_img_planes`: the family's planes as `DeviceArray`s (planetmapper_amd/device_array.py) in the
        engine's HBM, one launch for the missing ones, nothing crosses PCIe. Cached like the host planes (same keys with
        'img_dev'), invalidated with them by `_clear_cache`.
        """
        names = list(names)
        alt = self._alt_adjustment
        missing = [n for n in names if ('img_dev', n, alt) not in self._cache]
        if missing:
            if not self._test_if_img_size_valid():
                raise ValueError('nx and ny must be positive to create a backplane image')
            eng = self._bind()
            if not hasattr(eng, 'device_array'):
                raise _lib.UnsupportedError('this engine keeps no results on a device')
            out = {n: eng.device_array((self._ny, self._nx)) for n in missing}
            eng.backplanes_img_device(out, alt=alt)
            for n, a in out.items():
                self._cache[('img_dev', n, alt)] = a
        return {n: self._cache[('img_dev', n, alt)] for n in names}

    def prefetch_backplane_imgs(self, names: Iterable[str] | None = None, *, alt: float = 0.0) -> None:
        """
        Compute several backplane images with ONE kernel launch and leave them in the
        cache (what `Observation.save_observation` needs, observation.py:1269-1279).
        Not in the reference API; purely an optimisation hook.
        """
        if names is None:
            names = [n for n in self.backplanes if n in _lib.PLANE_INDEX]
        names = [self.standardise_backplane_name(n) for n in names]
        with _AltitudeContext(self, alt):
            self._img_planes([n for n in names if n in _lib.PLANE_INDEX])

    # ------------------------------------------------------------------ coordinate transforms
    def _transform(self, src, dst, a, b, *, alt=0.0, not_visible_nan=False, planetocentric=False, not_found_nan=True):
        """
        Array-valued transform with the reference's broadcasting rule
        (`SpiceBase._maybe_transform_as_arrays`, base.py:719-757): two floats in -> a tuple of
        floats, otherwise the inputs are broadcast together and arrays are returned.
        """
        scalar = np.ndim(a) == 0 and np.ndim(b) == 0
        oa, ob = self._bind().transform(
            src, dst, a, b, alt=alt, not_visible_nan=not_visible_nan, planetocentric=planetocentric
        )
        if not not_found_nan:
            bad = np.isnan(oa) & np.isfinite(np.broadcast_to(a, oa.shape)) & np.isfinite(np.broadcast_to(b, oa.shape))
            if np.any(bad):
                raise NotFoundError('ray does not intercept the target body')
        if scalar:
            return float(oa), float(ob)
        return oa, ob

    # BodyXY: body_xy.py:385-690
    def xy2radec(self, x, y):
        return self._transform('xy', 'radec', x, y)

    def radec2xy(self, ra, dec):
        return self._transform('radec', 'xy', ra, dec)

    def xy2lonlat(self, x, y, *, not_found_nan=True, alt=0.0, planetocentric=False):
        return self._transform('xy', 'lonlat', x, y, alt=alt, planetocentric=planetocentric, not_found_nan=not_found_nan)

    def lonlat2xy(self, lon, lat, *, alt=0.0, not_visible_nan=True, planetocentric=False):
        return self._transform('lonlat', 'xy', lon, lat, alt=alt, not_visible_nan=not_visible_nan, planetocentric=planetocentric)

    def xy2km(self, x, y):
        return self._transform('xy', 'km', x, y)

    def km2xy(self, km_x, km_y):
        return self._transform('km', 'xy', km_x, km_y)

    def xy2angular(self, x, y):
        return self._transform('xy', 'angular', x, y)

    def angular2xy(self, angular_x, angular_y):
        return self._transform('angular', 'xy', angular_x, angular_y)

    # Body: body.py:1083-1217, 1375-1900 (default angular origin only)
    def lonlat2radec(self, lon, lat, *, alt=0.0, not_visible_nan=True, planetocentric=False):
        return self._transform('lonlat', 'radec', lon, lat, alt=alt, not_visible_nan=not_visible_nan, planetocentric=planetocentric)

    def radec2lonlat(self, ra, dec, *, not_found_nan=True, alt=0.0, planetocentric=False):
        return self._transform('radec', 'lonlat', ra, dec, alt=alt, planetocentric=planetocentric, not_found_nan=not_found_nan)

    def radec2angular(self, ra, dec):
        return self._transform('radec', 'angular', ra, dec)

    def angular2radec(self, angular_x, angular_y):
        return self._transform('angular', 'radec', angular_x, angular_y)

    def angular2lonlat(self, angular_x, angular_y, *, not_found_nan=True, alt=0.0, planetocentric=False):
        return self._transform('angular', 'lonlat', angular_x, angular_y, alt=alt, planetocentric=planetocentric, not_found_nan=not_found_nan)

    def lonlat2angular(self, lon, lat, *, alt=0.0, not_visible_nan=True, planetocentric=False):
        return self._transform('lonlat', 'angular', lon, lat, alt=alt, not_visible_nan=not_visible_nan, planetocentric=planetocentric)

    def km2radec(self, km_x, km_y):
        return self._transform('km', 'radec', km_x, km_y)

    def radec2km(self, ra, dec):
        return self._transform('radec', 'km', ra, dec)

    def km2lonlat(self, km_x, km_y, *, not_found_nan=True, alt=0.0, planetocentric=False):
        return self._transform('km', 'lonlat', km_x, km_y, alt=alt, planetocentric=planetocentric, not_found_nan=not_found_nan)

    def lonlat2km(self, lon, lat, *, alt=0.0, not_visible_nan=True, planetocentric=False):
        return self._transform('lonlat', 'km', lon, lat, alt=alt, not_visible_nan=not_visible_nan, planetocentric=planetocentric)

    def km2angular(self, km_x, km_y):
        return self._transform('km', 'angular', km_x, km_y)

    def angular2km(self, angular_x, angular_y):
        return self._transform('angular', 'km', angular_x, angular_y)

    # ------------------------------------------------------------------ point functions
    # The per-point siblings of the backplanes (Body.*_from_lonlat, body.py:2152-2415, 2549-2616,
    # 2830-2913): the same map-space kernel evaluated at caller-supplied longitude / latitude
    # points, with the reference's float-or-array broadcasting rule.
    def _point_planes(self, names, lon, lat, *, alt=0.0, planetocentric=False) -> list:
        if alt != 0.0:
            # the reference puts these points `alt` km above the unadjusted ellipsoid (pgrrec),
            # the map kernel works on the altitude-adjusted ellipsoid of the backplanes
            raise _lib.UnsupportedError('point functions are provided for alt = 0 only')
        scalar = np.ndim(lon) == 0 and np.ndim(lat) == 0
        if planetocentric:
            lon, lat = self.centric2graphic_lonlat(lon, lat)
        lon_b, lat_b = np.broadcast_arrays(np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64))
        shape = lon_b.shape
        if lon_b.size == 0:
            return [np.empty(shape) for _ in names]
        lon2 = np.ascontiguousarray(lon_b.reshape(1, -1))
        lat2 = np.ascontiguousarray(lat_b.reshape(1, -1))
        out = self._bind().backplanes_map(list(names), lon2, lat2, alt=0.0)
        res = [out[n].reshape(shape) for n in names]
        return [float(r) for r in res] if scalar else res

    def illumination_angles_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):
        """(phase, incidence, emission) in degrees. body.py:2295-2332"""
        ph, inc, em = self._point_planes(['PHASE', 'INCIDENCE', 'EMISSION'], lon, lat, alt=alt, planetocentric=planetocentric)
        return ph, inc, em

    def azimuth_angle_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):
        """body.py:2334-2374"""
        return self._point_planes(['AZIMUTH'], lon, lat, alt=alt, planetocentric=planetocentric)[0]

    def distance_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):
        """Observer -> point distance in km. body.py:2885-2903"""
        return self._point_planes(['DISTANCE'], lon, lat, alt=alt, planetocentric=planetocentric)[0]

    def radial_velocity_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):
        """Line-of-sight velocity in km/s. body.py:2858-2883"""
        return self._point_planes(['RADIAL-VELOCITY'], lon, lat, alt=alt, planetocentric=planetocentric)[0]

    def local_solar_time_from_lon(self, lon):
        """Local solar time in local hours (et2lst, whole seconds). body.py:2376-2398"""
        return self._point_planes(['LOCAL-SOLAR-TIME'], lon, np.zeros_like(np.asarray(lon, dtype=np.float64)))[0]

    def local_solar_time_string_from_lon(self, lon) -> str:
        """'HH:MM:SS', '' for a non-finite longitude. body.py:2400-2415"""
        lst = float(self.local_solar_time_from_lon(float(lon)))
        if not math.isfinite(lst):
            return ''
        sec = int(round(lst * 3600.0))
        return f'{sec // 3600:02d}:{sec // 60 % 60:02d}:{sec % 60:02d}'

    def test_if_lonlat_illuminated(self, lon, lat, *, alt=0.0, planetocentric=False):
        """illumf's `lit` flag: incidence < 90 deg; False for invalid points. body.py:2549-2577"""
        inc = self._point_planes(['INCIDENCE'], lon, lat, alt=alt, planetocentric=planetocentric)[0]
        lit = np.asarray(inc) < 90.0
        return bool(lit) if lit.ndim == 0 else lit

    def test_if_lonlat_visible(self, lon, lat, *, alt=0.0, planetocentric=False):
        """
        Whether a point on (or `alt` km above) the body can be seen by the observer
        (body.py:2152-2180): the `not_visible_nan` rule of the lon/lat transforms.
        """
        ra, _ = self.lonlat2radec(lon, lat, alt=alt, not_visible_nan=True, planetocentric=planetocentric)
        vis = np.isfinite(np.asarray(ra))
        return bool(vis) if vis.ndim == 0 else vis

    def ring_plane_coordinates(self, ra, dec, only_visible: bool = True):
        """
        (ring_radius [km], ring_longitude [deg], ring_distance [km]) of the point of the target's
        equatorial plane seen at RA/Dec; NaN where that point is hidden by the body when
        `only_visible`. body.py:2577-2658
        """
        scalar = np.ndim(ra) == 0 and np.ndim(dec) == 0
        q = self._bind().radec_query(ra, dec, ring_only_visible=only_visible)
        return (float(q[2]), float(q[3]), float(q[4])) if scalar else (q[2], q[3], q[4])

    def limb_coordinates_from_radec(self, ra, dec, *, alt=0.0, planetocentric=False):
        """
        (lon, lat, dist): the point of the limb nearest to the RA/Dec ray and the distance of
        the ray above it in km (negative on the disc). body.py:2040-2110
        """
        scalar = np.ndim(ra) == 0 and np.ndim(dec) == 0
        q = self._bind().radec_query(ra, dec, alt=alt)
        lon, lat, dist = q[5], q[6], q[7]
        if planetocentric:
            lon, lat = self.graphic2centric_lonlat(lon, lat)
        return (float(lon), float(lat), float(dist)) if scalar else (lon, lat, dist)

    # Wireframe coordinate producers that are plain applications of the lon/lat transforms
    # (the limb / terminator ellipses need CSPICE's limbpt / edterm and are not provided).
    def ring_radec(self, radius: float, npts: int = 360, only_visible: bool = True):
        """RA/Dec of a ring of `radius` km in the equatorial plane. body.py:2660-2692"""
        lons = np.linspace(0, 360, npts)
        return self.lonlat2radec(lons, np.zeros(npts), alt=radius - self.r_eq, not_visible_nan=only_visible)

    def ring_xy(self, radius: float, **kwargs):
        """body_xy.py:1238-1249"""
        return self.radec2xy(*self.ring_radec(radius, **kwargs))

    def visible_lon_grid_radec(self, lons, npts: int = 60, *, lat_limit: float = 90.0, alt: float = 0.0,
                               planetocentric: bool = False) -> list:
        """Visible parts of lines of constant longitude. body.py:2725-2779"""
        lats = np.linspace(-lat_limit, lat_limit, npts)
        out = []
        for lon in lons:
            lo, la = np.full(npts, float(lon)), lats
            if planetocentric:
                lo, la = self.centric2graphic_lonlat(lo, la)
            out.append(self.lonlat2radec(lo, la, alt=alt, not_visible_nan=True))
        return out

    def visible_lat_grid_radec(self, lats, npts: int = 120, *, lat_limit: float = 90.0, alt: float = 0.0,
                               planetocentric: bool = False) -> list:
        """Visible parts of lines of constant latitude. body.py:2781-2835"""
        lons = np.linspace(0, 360, npts)
        out = []
        for lat in lats:
            if abs(lat) > lat_limit:
                continue
            lo, la = lons, np.full(npts, float(lat))
            if planetocentric:
                lo, la = self.centric2graphic_lonlat(lo, la)
            out.append(self.lonlat2radec(lo, la, alt=alt, not_visible_nan=True))
        return out

    def visible_lonlat_grid_radec(self, interval: float = 30, **kwargs) -> list:
        """Longitude then latitude gridlines every `interval` degrees. body.py:2695-2723"""
        return self.visible_lon_grid_radec(np.arange(0, 360, interval), **kwargs) + self.visible_lat_grid_radec(
            np.arange(-90, 90, interval), **kwargs
        )

    def visible_lonlat_grid_xy(self, *args, **kwargs) -> list:
        """body_xy.py:1220-1236"""
        return [self.radec2xy(*rd) for rd in self.visible_lonlat_grid_radec(*args, **kwargs)]

    def graphic2centric_lonlat(self, lon, lat, *, alt=0.0):
        """
        Planetographic -> planetocentric (east-positive longitude in (-180, 180]) of the point
        `alt` km above the surface: pgrrec_c then reclat_c (body.py:2915-2943), closed forms.
        """
        scalar = np.ndim(lon) == 0 and np.ndim(lat) == 0
        lon_b, lat_b = np.broadcast_arrays(np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64))
        a, c = float(self.radii[0]), float(self.radii[2])
        with np.errstate(invalid='ignore'):
            ok = np.isfinite(lon_b) & np.isfinite(lat_b)
            le = np.deg2rad(np.where(ok, lon_b, 0.0)) * (-1.0 if self.positive_longitude_direction == 'W' else 1.0)
            phi = np.deg2rad(np.where(ok, lat_b, 0.0))
            e2 = 1.0 - (c / a) ** 2
            n = a / np.sqrt(1.0 - e2 * np.sin(phi) ** 2)  # georec_c
            rho = (n + alt) * np.cos(phi)
            x, y, z = rho * np.cos(le), rho * np.sin(le), (n * (1.0 - e2) + alt) * np.sin(phi)
            lon_c = np.rad2deg(np.arctan2(y, x))
            lat_c = np.rad2deg(np.arctan2(z, np.hypot(x, y)))
        lon_c = np.where(ok, lon_c, np.nan)
        lat_c = np.where(ok, lat_c, np.nan)
        return (float(lon_c), float(lat_c)) if scalar else (lon_c, lat_c)

    def centric2graphic_lonlat(self, lon_centric, lat_centric, *, alt=0.0):
        """
        Planetocentric -> planetographic: the surface point in that direction (latsrf_c), then
        recpgr_c (body.py:2945-2982). Spheroids, alt = 0.
        """
        if alt != 0.0 or self.radii[0] != self.radii[1]:
            raise _lib.UnsupportedError('centric2graphic_lonlat is provided for spheroids at alt = 0')
        scalar = np.ndim(lon_centric) == 0 and np.ndim(lat_centric) == 0
        lon_b, lat_b = np.broadcast_arrays(
            np.asarray(lon_centric, dtype=np.float64), np.asarray(lat_centric, dtype=np.float64)
        )
        a, c = float(self.radii[0]), float(self.radii[2])
        with np.errstate(invalid='ignore'):
            ok = np.isfinite(lon_b) & np.isfinite(lat_b)
            lam = np.deg2rad(np.where(ok, lon_b, 0.0))
            th = np.deg2rad(np.where(ok, lat_b, 0.0))
            # a surface point in direction th has z / rho = tan th; its normal has
            # tan(graphic latitude) = (a / c)^2 z / rho
            lat_g = np.rad2deg(np.arctan2((a / c) ** 2 * np.sin(th), np.cos(th)))
            le = np.arctan2(np.sin(lam) * np.cos(th), np.cos(lam) * np.cos(th))  # recpgr: atan2(y, x)
            polar = np.cos(th) == 0.0
            lon_g = np.rad2deg(-le if self.positive_longitude_direction == 'W' else le)
            lon_g = np.where(lon_g < 0.0, lon_g + 360.0, lon_g)
            lon_g = np.where(polar, 0.0, lon_g)
        lon_g = np.where(ok, lon_g, np.nan) + 0.0
        lat_g = np.where(ok, lat_g, np.nan)
        return (float(lon_g), float(lat_g)) if scalar else (lon_g, lat_g)

    # ------------------------------------------------------------------ map coordinates
    def generate_map_coordinates(
        self,
        projection: str = 'rectangular',
        *,
        degree_interval: float = 1,
        lon: float = 0,
        lat: float = 0,
        size: int = 100,
        lon_coords=None,
        lat_coords=None,
        projection_x_coords=None,
        projection_y_coords=None,
        xlim=None,
        ylim=None,
        alt: float = 0.0,
    ):
        """
        body_xy.py:2755-3012: `'rectangular'`, `'manual'`, `'orthographic'`, `'azimuthal'` and
        `'azimuthal equal area'` (closed forms of the PROJ transforms, `projections.py`);
        custom proj strings need pyproj and are not supported.
        Returns `(lons, lats, xx, yy, transformer, info)` with `transformer = None`.
        """
        if projection == 'rectangular':
            lons = np.arange(degree_interval / 2, 360, degree_interval)
            if self.positive_longitude_direction == 'W':
                lons = lons[::-1]
            lats = np.arange(-90 + degree_interval / 2, 90, degree_interval)
            lons, lats = np.meshgrid(lons, lats)
            xx, yy = lons, lats
            info: dict[str, Any] = dict(projection=projection, degree_interval=degree_interval)
        elif projection == 'manual':
            if lon_coords is None or lat_coords is None:
                raise ValueError('lon_coords and lat_coords must be provided for manual projection')
            lons = np.asarray(lon_coords, dtype=float)
            lats = np.asarray(lat_coords, dtype=float)
            if lons.ndim != lats.ndim:
                raise ValueError('lon_coords and lat_coords must have the same number of dimensions')
            if lons.ndim == 1:
                lons, lats = np.meshgrid(lons, lats)
            if lons.ndim != 2:
                raise ValueError('lon_coords and lat_coords must be 1D or 2D arrays')
            if lons.shape != lats.shape:
                raise ValueError('lon_coords and lat_coords must have the same shape')
            lons = np.array(lons, dtype=float)
            lats = np.array(lats, dtype=float)
            xx, yy = lons, lats
            info = dict(projection=projection)
        elif projection in ('orthographic', 'azimuthal', 'azimuthal equal area'):
            # closed forms of the PROJ inverse the reference uses (body_xy.py:2930-2968)
            from . import projections

            west = self.positive_longitude_direction == 'W'
            if projection == 'orthographic':
                lons, lats, xx, yy = projections.orthographic(self.r_eq, self.r_polar, lon, lat, size, west)
            elif projection == 'azimuthal':
                lons, lats, xx, yy = projections.azimuthal(lon, lat, size, west)
            else:
                lons, lats, xx, yy = projections.azimuthal_equal_area(lon, lat, size, west)
            info = dict(projection=projection, lon=lon, lat=lat, size=size)
        else:
            raise _lib.UnsupportedError(
                f'custom proj string {projection!r} needs pyproj in the reference and is not part of this '
                "path; pass its lon/lat grids with projection='manual'"
            )
        info['xlim'] = xlim
        info['ylim'] = ylim
        if xlim is not None:
            keep = (xx[0] >= min(xlim)) & (xx[0] <= max(xlim))
            xx, yy, lons, lats = xx[:, keep], yy[:, keep], lons[:, keep], lats[:, keep]
        if ylim is not None:
            keep = (yy[:, 0] >= min(ylim)) & (yy[:, 0] <= max(ylim))
            xx, yy, lons, lats = xx[keep, :], yy[keep, :], lons[keep, :], lats[keep, :]
        lons = np.array(lons, dtype=float)
        lats = np.array(lats, dtype=float)
        lons[~np.isfinite(lons)] = np.nan
        lats[~np.isfinite(lats)] = np.nan
        if alt != 0.0:
            info['alt'] = alt
        return _readonly(lons), _readonly(lats), _readonly(np.array(xx)), _readonly(np.array(yy)), None, info

    @staticmethod
    def _split_map_kwargs(map_kwargs: dict) -> tuple[dict, float]:
        for k in map_kwargs:
            if k not in MAP_KWARG_KEYS:
                raise TypeError(f'unexpected map keyword argument {k!r}')
        alt = float(map_kwargs.get('alt', 0.0))
        return map_kwargs, alt

    def _map_key(self, map_kwargs: dict) -> tuple:
        return tuple(sorted((k, _freeze(v)) for k, v in map_kwargs.items()))

    def _get_lonlat_grids(self, **map_kwargs) -> tuple[np.ndarray, np.ndarray]:
        """
        The map grid as two contiguous (n0, n1) arrays, longitudes modulo 360, non-finite -> NaN: the form the engine
        takes. (Cut out of the reference's interleaved (n0, n1, 2) array they cost two strided copies per call - 15 ms
        for a 0.1 deg grid, three times what the engine then needs for the x / y maps.)
        """
        key = ('lonlat_grids', self._map_key(map_kwargs))
        if key not in self._stable_cache:
            lons, lats, *_ = self.generate_map_coordinates(**map_kwargs)
            lons = np.ascontiguousarray(lons % 360, dtype=np.float64)
            lats = np.array(lats, dtype=np.float64, order='C')  # (a copy: meshgrid's outputs may be the caller's)
            lons[~np.isfinite(lons)] = np.nan
            lats[~np.isfinite(lats)] = np.nan
            self._stable_cache[key] = (_readonly(lons), _readonly(lats))
        return self._stable_cache[key]

    def _get_lonlat_map(self, **map_kwargs) -> np.ndarray:
        """body_xy.py:3290-3300: (n0, n1, 2) with longitudes modulo 360."""
        key = ('lonlat_map', self._map_key(map_kwargs))
        if key not in self._stable_cache:
            self._stable_cache[key] = _readonly(np.stack(self._get_lonlat_grids(**map_kwargs), axis=-1))
        return self._stable_cache[key]

    def _map_planes(self, names: Iterable[str], map_kwargs: dict) -> dict[str, np.ndarray]:
        """
        Cached map-space planes. Disc-independent planes live in the stable cache; x/y
        maps depend on the disc parameters (body_xy.py:3478) and live in the clearable one.
        """
        names = list(names)
        map_kwargs, alt = self._split_map_kwargs(dict(map_kwargs))
        mkey = self._map_key(map_kwargs)

        def slot(n: str):
            disc_dep = n in ('PIXEL-X', 'PIXEL-Y')
            cache = self._cache if disc_dep else self._stable_cache
            return cache, ('map', n, mkey)

        missing = [n for n in names if slot(n)[1] not in slot(n)[0]]
        if missing:
            lon, lat = self._get_lonlat_grids(**map_kwargs)
            out = self._bind().backplanes_map(missing, lon, lat, alt=alt)
            for n, a in out.items():
                cache, key = slot(n)
                cache[key] = _readonly(a)
        return {n: slot(n)[0][slot(n)[1]] for n in names}

    def _lonlat_grids_device(self, map_kwargs: dict) -> tuple:
        """the map grid in HBM (two `DeviceArray`s; it does not depend on the disc: kept across `_clear_cache`)"""
        key = ('lonlat_dev', self._map_key(map_kwargs))
        grids = self._stable_cache.get(key)
        if grids is None or not grids[0].valid:
            eng = self._bind()
            if not hasattr(eng, 'device_array'):
                raise _lib.UnsupportedError('this engine keeps no results on a device')
            lon, lat = self._get_lonlat_grids(**map_kwargs)
            grids = (eng.device_array(lon.shape), eng.device_array(lat.shape))
            eng.h2d(grids[0].ptr, lon)
            eng.h2d(grids[1].ptr, lat)
            self._stable_cache[key] = grids
        return grids

    def _map_planes_device(self, names: Iterable[str], map_kwargs: dict) -> dict:
        """
        The device form of `_map_planes`: map-space planes as `DeviceArray`s, computed from the grid's device copy into HBM,
        cached like the host planes (x / y maps with the disc, everything else for good).
        """
        names = list(names)
        map_kwargs, alt = self._split_map_kwargs(dict(map_kwargs))
        mkey = self._map_key(map_kwargs)

        def slot(n: str):
            return (self._cache if n in ('PIXEL-X', 'PIXEL-Y') else self._stable_cache), ('map_dev', n, mkey)

        missing = [n for n in names if slot(n)[1] not in slot(n)[0] or not slot(n)[0][slot(n)[1]].valid]
        if missing:
            lon_d, lat_d = self._lonlat_grids_device(map_kwargs)
            eng = self._bind()
            out = {n: eng.device_array(lon_d.shape) for n in missing}
            eng.backplanes_map_device(out, lon_d, lat_d, lon_d.shape[0], lon_d.shape[1], alt=alt)
            for n, a in out.items():
                cache, key = slot(n)
                cache[key] = a
        return {n: slot(n)[0][slot(n)[1]] for n in names}

    def get_x_map(self, *, device: bool = False, **map_kwargs):
        if device:
            return self._map_planes_device(['PIXEL-X', 'PIXEL-Y'], map_kwargs)['PIXEL-X']
        return self._map_planes(['PIXEL-X', 'PIXEL-Y'], map_kwargs)['PIXEL-X']

    def get_y_map(self, *, device: bool = False, **map_kwargs):
        if device:
            return self._map_planes_device(['PIXEL-X', 'PIXEL-Y'], map_kwargs)['PIXEL-Y']
        return self._map_planes(['PIXEL-X', 'PIXEL-Y'], map_kwargs)['PIXEL-Y']

    # ------------------------------------------------------------------ reprojection
    def map_img(
        self,
        img: np.ndarray,
        *,
        interpolation: str | int | tuple[int, int] = 'linear',
        spline_smoothing: float = 0,
        propagate_nan: bool = True,
        warn_nan: bool = False,
        smooth_oversample_by: int = 5,
        smooth_max_oversampled_img_size: int = 10_000,
        **map_kwargs,
    ) -> np.ndarray:
        """
        body_xy.py:1414-1631: project an image (ny, nx) - or a cube (P, ny, nx) - onto
        the map grid. `'nearest'`, `'linear'`, `'quadratic'`, `'cubic'`, integer degrees and
        `(k_rows, k_cols)` tuples (RectBivariateSpline with s=0) and `'smooth'` (PCHIP
        oversampling + bilinear, body_xy.py:1704-1853) run on the GPU, and so do the FITPACK
        smoothing splines of `spline_smoothing > 0`.
        """
        img = np.asarray(img)
        from .engine import interpolation_code

        interpolation_code(interpolation)  # ValueError for unknown methods (body_xy.py:1630)
        if spline_smoothing < 0:
            raise ValueError('s should be s >= 0.0')  # scipy's message (RectBivariateSpline)
        interp = interpolation
        single = img.ndim == 2
        if img.ndim not in (2, 3) or img.shape[-2:] != (self._ny, self._nx):
            raise ValueError(
                f'The input `img` shape {img.shape!r} is inconsistent with the body\'s image size '
                f'(ny={self._ny}, nx={self._nx})'
            )
        x_map = self.get_x_map(**map_kwargs)
        y_map = self.get_y_map(**map_kwargs)
        if warn_nan and interp != 'nearest' and not np.all(np.isfinite(img)):
            print('Warning, image contains NaN values which will be corrected')
        if interp == 'smooth':
            # (no visible cell at all; the strided sample settles the usual case without a pass over a fine map)
            if np.isnan(x_map[::7, ::7]).all() and np.all(np.isnan(x_map)) and not np.all(np.isnan(img)):
                # the reference fails here too: `original[0]` of an empty trimmed axis (body_xy.py:1741)
                raise IndexError('index 0 is out of bounds for axis 0 with size 0')
            out = self._bind().map_cube(
                img, x_map, y_map, interp, propagate_nan,
                smooth_oversample_by=smooth_oversample_by,
                smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
            )  # fmt: skip
        elif interp != 'nearest' and spline_smoothing != 0:
            eng = self._bind()
            out = eng.map_cube(img, x_map, y_map, interp, propagate_nan, spline_smoothing=spline_smoothing)
            self._warn_about_smoothing_fits(eng)
        else:
            out = self._bind().map_cube(img, x_map, y_map, interp, propagate_nan)
        return out[0] if single else out

    @staticmethod
    def _warn_about_smoothing_fits(eng) -> None:
        """
        Say so when a smoothing-spline fit (`spline_smoothing > 0`) is one scipy itself does not pin down: the library counts
        the planes whose knot search chose between tied intervals by rounding noise, and those that went through a fit its
        least-squares solve does not resolve (include/planetmapper_hip.h: PM_OPT_LAST_SM_KNIFE_EDGES / _ILL_CONDITIONED).
        """
        if not hasattr(eng, 'get_option'):
            return
        ties = eng.get_option(_lib.PM_OPT_LAST_SM_KNIFE_EDGES)
        ill = eng.get_option(_lib.PM_OPT_LAST_SM_ILL_CONDITIONED)
        if ties or ill:
            import warnings

            warnings.warn(
                f'smoothing spline: {ties} plane(s) whose knot search chose between intervals tied to rounding (scipy\'s own '
                f'choice there flips under a one-ulp change of the data; the alternatives are percent-level apart), {ill} plane(s) '
                'with an ill-conditioned fit (up to 1e-4 of the data scale from scipy\'s): both are valid smoothing splines of '
                'the data for this `spline_smoothing`, not necessarily the one scipy returns',
                RuntimeWarning, stacklevel=3)  # fmt: skip

    # ------------------------------------------------------------------ backplane registry
    @staticmethod
    def standardise_backplane_name(name: str) -> str:
        return name.strip().upper()

    def register_backplane(self, name: str, description: str, get_img, get_map) -> None:
        """body_xy.py:2512-2539"""
        name = self.standardise_backplane_name(name)
        if name in self.backplanes:
            raise ValueError(f'Backplane named {name!r} is already registered')
        self.backplanes[name] = Backplane(name=name, description=description, get_img=get_img, get_map=get_map)

    def backplane_summary_string(self) -> str:
        return '\n'.join(f'{bp.name}: {bp.description}' for bp in self.backplanes.values())

    def print_backplanes(self) -> None:
        print(self.backplane_summary_string())

    def get_backplane(self, name: str) -> Backplane:
        name = self.standardise_backplane_name(name)
        try:
            return self.backplanes[name]
        except KeyError as exc:
            raise BackplaneNotFoundError(
                '{n!r} not found. Currently registered backplanes are: {r}.'.format(
                    n=name, r=', '.join([repr(n) for n in self.backplanes.keys()])
                )
            ) from exc

    def get_backplane_img(self, name: str, *, alt: float = 0.0, device: bool = False):
        """
        Fresh copy of a backplane image (body_xy.py:2586-2630). `device=True` (not in the reference): the plane where it
        was computed - a read-only `DeviceArray` in the GPU's HBM (`__dlpack__`, `__cuda_array_interface__`; valid until
        the disc or the image size changes) instead of a numpy copy that crossed PCIe; default backplanes only.
        """
        with _AltitudeContext(self, alt):
            bp = self.backplanes[self.standardise_backplane_name(name)]
            if device:
                fn = getattr(bp.get_img, '__func__', None)
                if fn is None or not getattr(fn, '_pm_device', False):
                    raise _lib.UnsupportedError(f'backplane {bp.name!r} is a user-registered function of numpy arrays: no device form')
                return bp.get_img(device=True)
            return bp.get_img().copy()

    def get_backplane_map(self, name: str, *, device: bool = False, **map_kwargs):
        """Fresh copy of a backplane map (body_xy.py:2632-2664); `device=True`: the plane as a read-only `DeviceArray` in
        HBM (see `get_backplane_img`), default backplanes only."""
        bp = self.backplanes[self.standardise_backplane_name(name)]
        if device:
            fn = getattr(bp.get_map, '__func__', None)
            if fn is None or not getattr(fn, '_pm_device', False):
                raise _lib.UnsupportedError(f'backplane {bp.name!r} is a user-registered function of numpy arrays: no device form')
            return bp.get_map(device=True, **map_kwargs)
        return bp.get_map(**map_kwargs).copy()

    def _register_default_backplanes(self) -> None:
        """The 26 default backplanes, same names/order/descriptions as body_xy.py:4198-4356."""
        ew = self.positive_longitude_direction
        table = [
            ('LON-GRAPHIC', f'Planetographic longitude, positive {ew} [deg]', 'lon'),
            ('LAT-GRAPHIC', 'Planetographic latitude [deg]', 'lat'),
            ('LON-CENTRIC', 'Planetocentric longitude [deg]', 'lon_centric'),
            ('LAT-CENTRIC', 'Planetocentric latitude [deg]', 'lat_centric'),
            ('RA', 'Right ascension [deg]', 'ra'),
            ('DEC', 'Declination [deg]', 'dec'),
            ('PIXEL-X', 'Observation x pixel coordinate [pixels]', 'x'),
            ('PIXEL-Y', 'Observation y pixel coordinate [pixels]', 'y'),
            ('KM-X', 'East-West distance in target plane [km]', 'km_x'),
            ('KM-Y', 'North-South distance in target plane [km]', 'km_y'),
            ('ANGULAR-X', 'East-West distance in target plane [arcsec]', 'angular_x'),
            ('ANGULAR-Y', 'North-South distance in target plane [arcsec]', 'angular_y'),
            ('PHASE', 'Phase angle [deg]', 'phase_angle'),
            ('INCIDENCE', 'Incidence angle [deg]', 'incidence_angle'),
            ('EMISSION', 'Emission angle [deg]', 'emission_angle'),
            ('AZIMUTH', 'Azimuth angle [deg]', 'azimuth_angle'),
            ('LOCAL-SOLAR-TIME', 'Local solar time [local hours]', 'local_solar_time'),
            ('DISTANCE', 'Distance to observer [km]', 'distance'),
            ('RADIAL-VELOCITY', 'Radial velocity away from observer [km/s]', 'radial_velocity'),
            ('DOPPLER', 'Doppler factor, sqrt((1 + v/c)/(1 - v/c)) where v is radial velocity', 'doppler'),
            ('LIMB-DISTANCE', 'Distance above limb [km]', 'limb_distance'),
            ('LIMB-LON-GRAPHIC', 'Planetographic longitude of closest point on the limb [deg]', 'limb_lon'),
            ('LIMB-LAT-GRAPHIC', 'Planetographic latitude of closest point on the limb [deg]', 'limb_lat'),
            ('RING-RADIUS', 'Equatorial (ring) plane radius [km]', 'ring_plane_radius'),
            ('RING-LON-GRAPHIC', 'Equatorial (ring) plane planetographic longitude [deg]', 'ring_plane_longitude'),
            ('RING-DISTANCE', 'Equatorial (ring) plane distance to observer [km]', 'ring_plane_distance'),
        ]  # fmt: skip
        for name, desc, stem in table:
            self.register_backplane(
                name, desc, getattr(self, f'get_{stem}_img'), getattr(self, f'get_{stem}_map')
            )


# Families computed together by the reference (one cached pixel loop each); a request
# for one member computes - and caches - its whole family with one launch.
_FAMILIES = {
    'lon': ('LON-GRAPHIC', ('LON-GRAPHIC', 'LAT-GRAPHIC')),  # _get_lonlat_img :3281
    'lat': ('LAT-GRAPHIC', ('LON-GRAPHIC', 'LAT-GRAPHIC')),
    'lon_centric': ('LON-CENTRIC', ('LON-CENTRIC', 'LAT-CENTRIC')),  # :3346
    'lat_centric': ('LAT-CENTRIC', ('LON-CENTRIC', 'LAT-CENTRIC')),
    'ra': ('RA', ('RA', 'DEC')),  # _get_radec_img :3409
    'dec': ('DEC', ('RA', 'DEC')),
    'x': ('PIXEL-X', ('PIXEL-X', 'PIXEL-Y')),  # :3494
    'y': ('PIXEL-Y', ('PIXEL-X', 'PIXEL-Y')),
    'km_x': ('KM-X', ('KM-X', 'KM-Y')),  # _get_km_xy_img :3545
    'km_y': ('KM-Y', ('KM-X', 'KM-Y')),
    'angular_x': ('ANGULAR-X', ('ANGULAR-X', 'ANGULAR-Y')),  # :3610
    'angular_y': ('ANGULAR-Y', ('ANGULAR-X', 'ANGULAR-Y')),
    'phase_angle': ('PHASE', ('PHASE', 'INCIDENCE', 'EMISSION')),  # _get_illumination_gie_img :3658
    'incidence_angle': ('INCIDENCE', ('PHASE', 'INCIDENCE', 'EMISSION')),
    'emission_angle': ('EMISSION', ('PHASE', 'INCIDENCE', 'EMISSION')),
    'azimuth_angle': ('AZIMUTH', ('PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH')),  # :3742
    'local_solar_time': ('LOCAL-SOLAR-TIME', ('LON-GRAPHIC', 'LAT-GRAPHIC', 'LOCAL-SOLAR-TIME')),  # :3787
    'distance': ('DISTANCE', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER')),  # _get_state_imgs :3830
    'radial_velocity': ('RADIAL-VELOCITY', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER')),
    'doppler': ('DOPPLER', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER')),
    'limb_distance': ('LIMB-DISTANCE', ('LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC')),  # :3964
    'limb_lon': ('LIMB-LON-GRAPHIC', ('LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC')),
    'limb_lat': ('LIMB-LAT-GRAPHIC', ('LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC')),
    'ring_plane_radius': ('RING-RADIUS', ('RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE')),  # :4059
    'ring_plane_longitude': ('RING-LON-GRAPHIC', ('RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE')),
    'ring_plane_distance': ('RING-DISTANCE', ('RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE')),
}


def _make_getters(stem: str, plane: str, family: tuple[str, ...]):
    def get_img(self: BodyXY, *, device: bool = False):
        if device:
            return self._img_planes_device(family)[plane]
        return self._img_planes(family)[plane]

    get_img._pm_device = True

    def get_map(self: BodyXY, *, device: bool = False, **map_kwargs):
        if device:
            return self._map_planes_device(family, map_kwargs)[plane]
        return self._map_planes(family, map_kwargs)[plane]

    get_map._pm_device = True

    get_img.__name__ = f'get_{stem}_img'
    get_map.__name__ = f'get_{stem}_map'
    get_img.__doc__ = (
        f'Read-only (ny, nx) float64 array of the {plane} backplane (reference `BodyXY.get_{stem}_img`); '
        'NaN where undefined. `device=True`: the same plane as a read-only `DeviceArray` left in the GPU\'s HBM.'
    )
    get_map.__doc__ = (
        f'Read-only (n0, n1) float64 map of the {plane} backplane (reference `BodyXY.get_{stem}_map`). '
        '`device=True`: the same map as a read-only `DeviceArray` left in the GPU\'s HBM.'
    )
    return get_img, get_map


for _stem, (_plane, _family) in _FAMILIES.items():
    _gi, _gm = _make_getters(_stem, _plane, _family)
    setattr(BodyXY, f'get_{_stem}_img', _gi)
    if _stem not in ('x', 'y'):  # get_x_map / get_y_map are defined explicitly above
        setattr(BodyXY, f'get_{_stem}_map', _gm)
BodyXY.get_x_map._pm_device = True
BodyXY.get_y_map._pm_device = True


class _AltitudeContext:
    """
    `_AdjustedSurfaceAltitude` (body.py:172-229): temporarily add `alt` km to all three
    radii. Here it only switches the altitude passed to the kernels (and the cache key);
    like the reference it cannot be nested with two different non-zero altitudes.
    """

    def __init__(self, body: BodyXY, alt: float = 0.0) -> None:
        self.body = body
        self.do = alt != 0.0 and alt != body._alt_adjustment
        if self.do:
            self.alt = float(alt)
            if not math.isfinite(self.alt):
                raise ValueError('Cannot adjust surface altitude with non-finite alt value')
            if body._alt_adjustment != 0.0:
                raise ValueError('Cannot nest _AdjustedSurfaceAltitude context managers with alt != 0')

    def _set_radii(self, alt: float) -> None:
        b = self.body
        b.radii = np.array(b._geometry.radii[:]) + alt  # body.py:196-214: all three radii, then
        b.r_eq = float(b.radii[0])  # the derived attributes
        b.r_polar = float(b.radii[2])
        b.flattening = (b.r_eq - b.r_polar) / b.r_eq

    def __enter__(self) -> None:
        if self.do:
            self.body._alt_adjustment = self.alt
            self._set_radii(self.alt)

    def __exit__(self, *exc) -> None:
        if self.do:
            self.body._alt_adjustment = 0.0
            self._set_radii(0.0)
