"""
The reference-side binding: a `planetmapper.Body` (SPICE does the one-time geometry) in, the engine's geometry block out,
and the mixin that swaps `BodyXY`'s pixel loops for engine calls.

This is the route by which observers the SPICE-free provider (`geometry.py` / `ephem.py`: text PCK + Chebyshev SPK types
2 / 3) cannot evaluate - HST, JWST, any spacecraft SPK - reach the engine: everything is asked of SPICE through the
handful of spiceypy calls below, on the user's machine. `spiceypy` (and `planetmapper`) are imported lazily and only
there; `tests/test_reference_binding.py` runs the same code against a stand-in for spiceypy built on this repo's own
ephemeris / PCK readers (`tests/spice_standin.py`) and a duck-typed Body, and holds the block against
`GeometryBuilder`'s to the bars of the motion-model test (displacement < 1e-9 km over +-4 R/c, rotation increment
< 1e-13 rad).

What the block holds and where the reference computes it: `planetmapper/body.py:501-606` (`Body.__init__`),
`planetmapper/base.py:795-839` (`BodyBase.__init__`), include/planetmapper_hip.h (`pm_geometry`).

    from planetmapper_amd.reference_binding import geometry_from_body, HipBackplanes
    class BodyXY(HipBackplanes, planetmapper.BodyXY): ...
"""

from __future__ import annotations

import os

import numpy as np

from .geometry import PMGeometry

J2000 = 'J2000'


def _spiceypy():
    import spiceypy  # noqa: PLC0415 - on the user's machine only

    return spiceypy


def _chain(spice, body: int, et: float) -> list[tuple[int, int, int]]:
    """
    The segments CSPICE adds up for the SSB position of `body` at `et`: [(body, centre, SPK type), ...] down to the
    solar system barycentre (spksfs / spkuds). Without those two calls: the standard NAIF chain body -> system
    barycentre -> SSB, types unknown.
    """
    hops = []
    cur = int(body)
    try:
        while cur != 0 and len(hops) < 8:
            _, descr, _ = spice.spksfs(cur, et, 41)
            _, centre, _, spk_type, *_ = spice.spkuds(descr)
            hops.append((cur, int(centre), int(spk_type)))
            cur = int(centre)
        if cur == 0:
            return hops
    except (AttributeError, NotImplementedError):
        pass
    if body == 0:
        return []
    centre = int(body) // 100 if int(body) >= 100 else 0
    return [(int(body), centre, 0)] + ([(centre, 0, 0)] if centre else [])


def _series_derivatives(values: np.ndarray, offsets: np.ndarray, h: float) -> tuple[np.ndarray, np.ndarray]:
    """
    First and second derivative at offset 0 of the quartic through five samples, `values[k]` taken at `offsets[k]` (~ -2h,
    -h, 0, h, 2h: the ACTUAL differences of the epochs that were evaluated - epochs are doubles with a quantum of 3e-8 s,
    which a nominal step would turn into 1e-7 km/s of velocity).
    """
    x = offsets / h
    vand = np.vander(x, 5, increasing=True)
    coeff = np.linalg.solve(vand, values - values[2])  # (differences of neighbouring positions: exact)
    return coeff[1] / h, 2.0 * coeff[2] / (h * h)


def position_series_derivatives(spice, body: int, et: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """
    (p, dp/dt, d2p/dt2) of the SSB POSITION series of `body` at `et` - the series CSPICE re-evaluates at every light-time
    epoch (`sincpt`, `illumf`, `spkcpt`), which the kernels replace by p + v d + a d^2 / 2.

    Hop by hop along the chain of segments, so that what is differenced is of the hop's own size (a moon about its
    planet: 4e5 km, ulp 6e-11 km) and not an SSB vector (1e9 km, ulp 1e-7 km): a type 2 segment's state velocity IS the
    derivative of its position polynomial (taken as it is; the acceleration from differences of it); any other type (3: a
    velocity polynomial of its own; 5, 9, 13, ...) is differentiated numerically from positions, five points, a step chosen
    for the size of the vector (rounding ~ ulp / h, truncation ~ h^4 p^(5) / 30).
    """
    p = np.zeros(3)
    v = np.zeros(3)
    a = np.zeros(3)
    for hop, centre, spk_type in _chain(spice, body, et):
        state, _ = spice.spkgeo(hop, et, J2000, centre)
        state = np.asarray(state, dtype=float)
        p += state[:3]
        size = float(np.linalg.norm(state[:3]))
        if spk_type == 2:
            h = 64.0
            ts = np.array([et - 2 * h, et - h, et, et + h, et + 2 * h])
            vel = np.array([np.asarray(spice.spkgeo(hop, float(t), J2000, centre)[0], dtype=float)[3:] for t in ts])
            v += state[3:]
            a += _series_derivatives(vel, ts - et, h)[0]
        else:
            h = float(min(4096.0, max(4.0, 2.0 ** np.ceil(np.log2(7.5e9 * np.spacing(max(size, 1.0)))))))
            ts = np.array([et - 2 * h, et - h, et, et + h, et + 2 * h])
            pos = np.array([np.asarray(spice.spkgps(hop, float(t), J2000, centre)[0], dtype=float) for t in ts])
            dv, da = _series_derivatives(pos, ts - et, h)
            v += dv
            a += da
    return p, v, a


def state_velocity_derivative(spice, body: int, et: float) -> np.ndarray:
    """d/dt of the STATE velocity of `body` wrt the SSB (what `spkcpt`'s velocities follow), five-point differences"""
    h = 64.0
    ts = np.array([et - 2 * h, et - h, et, et + h, et + 2 * h])
    vel = np.array([np.asarray(spice.spkssb(body, float(t), J2000), dtype=float)[3:] for t in ts])
    return _series_derivatives(vel, ts - et, h)[0]


def body_z_rate(spice, frame: str, et: float) -> float:
    """
    Angular velocity of `frame` about its own +z axis, rad/s, from the derivative block of sxform('J2000', frame, et) -
    CSPICE differentiates the IAU model term by term (trigonometric terms of W, RA and Dec included: every moon of
    pck00010.tpc has them): for an inertial vector i = R' b of a body-fixed b, di/dt = w x i with [w]x = (dR/dt)' R.
    """
    xf = np.asarray(spice.sxform(J2000, frame, et), dtype=float)
    rot, drot = xf[:3, :3], xf[3:, :3]
    skew = drot.T @ rot
    w_inertial = np.array([skew[2, 1], skew[0, 2], skew[1, 0]])
    return float((rot @ w_inertial)[2])


def pole_drift(spice, frame: str, et: float) -> np.ndarray:
    """the frame's angular velocity less its component about the frame's own z axis (J2000 components): pm_geometry.WP"""
    xf = np.asarray(spice.sxform(J2000, frame, et), dtype=float)
    rot, drot = xf[:3, :3], xf[3:, :3]
    skew = drot.T @ rot
    w_inertial = np.array([skew[2, 1], skew[0, 2], skew[1, 0]])
    return w_inertial - float(rot[2] @ w_inertial) * rot[2]


def geometry_from_body(body, spice=None) -> PMGeometry:
    """
    The engine's geometry block of a reference `planetmapper.Body` (or anything that carries the attributes read here).
    `spice`: the module to ask (default: spiceypy with the kernels `body` has loaded).
    """
    spice = spice if spice is not None else _spiceypy()
    if getattr(body, 'aberration_correction', 'CN') != 'CN':
        raise NotImplementedError("the engine evaluates the reference's default aberration correction 'CN' only")
    if getattr(body, 'observer_frame', J2000) != J2000:
        raise NotImplementedError("the engine works in the reference's default observer frame 'J2000' only")
    g = PMGeometry()
    et, lt = float(body.et), float(body.target_light_time)  # base.py:828-839
    t0 = et - lt
    target = int(body.target_body_id)
    g.et, g.lt_c, g.clight = et, lt, float(spice.clight())
    g.radii[:] = [float(r) for r in body.radii]  # body.py:521
    g.T0[:] = [float(x) for x in body._target_obsvec]  # base.py:836
    # the target centre about t0: position series (what a re-evaluation at et - lt' follows) and what a STATE adds
    p_t0, v_series, a_series = position_series_derivatives(spice, target, t0)
    state = np.asarray(spice.spkssb(target, t0, J2000), dtype=float)
    g.VT[:] = v_series
    g.AT[:] = a_series
    g.DVT[:] = state[3:] - v_series
    g.DAT[:] = state_velocity_derivative(spice, target, t0) - a_series
    g.VO[:] = np.asarray(spice.spkssb(int(spice.bods2c(body.observer)), et, J2000), dtype=float)[3:]
    # the illumination source seen from the target centre at t0 (one-way light time)
    source = getattr(body, 'illumination_source', 'SUN')
    _, lts = spice.spkpos(source, t0, J2000, 'CN', body.target)
    ts0 = t0 - float(lts)
    p_s, v_s, a_s = position_series_derivatives(spice, int(spice.bods2c(source)), ts0)
    g.ts0 = ts0
    g.S0[:] = p_s - state[:3]
    g.VS[:] = v_s
    g.AS[:] = a_s
    # orientation: the frame at t0 and its rate about its own z axis (R(t0 + d) = Rz(wdot d) R0)
    g.R0[:] = np.asarray(spice.pxform(J2000, body.target_frame, t0), dtype=float).ravel()
    g.wdot = body_z_rate(spice, body.target_frame, t0)
    g.WP[:] = pole_drift(spice, body.target_frame, t0)
    # what Body.__init__ has already asked of SPICE (body.py:538-588)
    g.sub_sp[:] = [float(x) for x in body._subpoint_targvec]
    g.sub_ray[:] = [float(x) for x in body._subpoint_rayvec]
    g.sub_obsvec[:] = [float(x) for x in body._subpoint_obsvec]
    g.sub_et = float(body._subpoint_et)
    g.sub_dist = float(body.subpoint_distance)
    normal, const = spice.pl2nvc(body._ring_plane)  # body.py:585
    g.ring_n[:] = [float(x) for x in normal]
    g.ring_k = float(const)
    g.M[:] = np.asarray(body._get_obsvec2angular_matrix(), dtype=float).ravel()  # body.py:1317
    g.diameter_arcsec = float(body.target_diameter_arcsec)
    g.km_per_arcsec = float(body.km_per_arcsec)
    g.np_angle_rad = float(np.deg2rad(body.north_pole_angle()))  # body.py:2985
    sun_b, _ = spice.spkpos('SUN', t0, body.target_frame, 'LT+S', body.target)  # et2lst, body.py:2364
    g.lst_sun_lon = float(np.arctan2(sun_b[1], sun_b[0]))
    g.west_positive = int(body.positive_longitude_direction == 'W')  # body.py:526-535
    return g


def _freeze(v):
    """hashable stand-in for a keyword value of a mapping function (the reference turns arrays into nested tuples, base.py:41)"""
    if isinstance(v, np.ndarray):
        return ('__ndarray__', v.shape, v.tobytes())
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    return v


def _readonly(a: np.ndarray) -> np.ndarray:
    v = a.view()
    v.setflags(write=False)
    return v


class HipBackplanes:
    """
    Mixed into the reference's `BodyXY` (before it in the MRO): the pixel loops of `body_xy.py:3195-4190` and `map_img`
    become engine calls. Results are cached where the reference caches them - image-space producers in the Body's own
    `_cache` (emptied by its `_clear_cache()` whenever the disc changes, keyed with the altitude adjustment like
    `_cache_clearable_alt_dependent_result`, body.py:255-272), map-space producers in `_stable_cache` (base.py:91-112) - and
    handed out as read-only arrays (`_return_readonly_array`, base.py:115-138). The registry, `get_backplane_img / map`,
    the map-grid helpers (`_get_lonlat_map` with every projection), `_AdjustedSurfaceAltitude` and the argument handling
    stay the reference's. Names and return shapes follow the reference method each override replaces (line numbers in
    `_IMG_FAMILIES` / `_MAP_FAMILIES`).
    """

    _hip_spice = None  # a module to ask instead of spiceypy (tests)

    # ------------------------------------------------------------------ the engine, bound to this body's geometry and disc
    _hip_device = None  # the GPU of this process: None = LOCAL_RANK where the launcher left every card visible, else 0

    def _hip(self):
        """the process's ONE context on this rank's device (shared with every other body, like the native BodyXY's),
        pointed at this body's geometry block - which is all a body keeps - and at its disc"""
        from .body_xy import _shared_engine  # noqa: PLC0415
        from .engine import device_count  # noqa: PLC0415

        device = self._hip_device
        if device is None:
            local = int(os.environ.get('LOCAL_RANK', '0') or 0)
            device = local if 0 <= local < device_count() else 0
        eng = _shared_engine(int(device))
        g = self.__dict__.get('_hip_geometry')
        if g is None:
            g = self.__dict__['_hip_geometry'] = geometry_from_body(self, self._hip_spice)  # (et, target and observer of a Body never change)
        eng.set_geometry(g)
        eng.set_disc(self.get_x0(), self.get_y0(), self.get_r0(), self._get_rotation_radians(), self._nx, self._ny, self._optimize_speed)
        return eng

    def _hip_img_family(self, names: tuple) -> dict:
        """the planes of one image-space family: one launch, cached until the disc (or the altitude adjustment) changes"""
        alt = float(getattr(self, '_alt_adjustment', 0.0))
        key = ('_hip_img', names, alt)
        if key not in self._cache:
            out = self._hip().backplanes_img(list(names), alt=alt)
            self._cache[key] = {n: _readonly(a) for n, a in out.items()}
        return self._cache[key]

    def _hip_lonlat_grids(self, map_kwargs: dict) -> tuple:
        """the reference's own grid (`_get_lonlat_map`: every projection it knows) as two contiguous arrays, cut out of the
        interleaved (n0, n1, 2) array once per grid - the two strided copies are 15 ms on a 0.1 deg grid"""
        key = ('_hip_lonlat', frozenset((k, _freeze(v)) for k, v in map_kwargs.items() if k != 'alt'))
        if key not in self._stable_cache:
            ll = self._get_lonlat_map(**map_kwargs)
            self._stable_cache[key] = (_readonly(np.ascontiguousarray(ll[..., 0])), _readonly(np.ascontiguousarray(ll[..., 1])))
        return self._stable_cache[key]

    def _hip_map_family(self, names: tuple, map_kwargs: dict) -> dict:
        """the planes of one map-space family on the grid the reference builds for `map_kwargs` (stable: no disc dependence)"""
        key = ('_hip_map', names, frozenset((k, _freeze(v)) for k, v in map_kwargs.items()))
        xy = 'PIXEL-X' in names  # (pixel coordinates of map cells DO depend on the disc: the reference's _get_xy_map is clearable)
        cache = self._cache if xy else self._stable_cache
        if xy:
            key = key + (self.get_x0(), self.get_y0(), self.get_r0(), self._get_rotation_radians())
        if key not in cache:
            lon, lat = self._hip_lonlat_grids(map_kwargs)
            out = self._hip().backplanes_map(list(names), lon, lat, alt=float(map_kwargs.get('alt', 0.0)))
            cache[key] = {n: _readonly(a) for n, a in out.items()}
        return cache[key]

    # ------------------------------------------------------------------ reprojection and the point forms
    def map_img(self, img, *, interpolation='linear', propagate_nan=True, warn_nan=False, spline_smoothing=0, smooth_oversample_by=5,
                smooth_max_oversampled_img_size=10_000, **map_kwargs):  # body_xy.py:1414  # fmt: skip
        eng = self._hip()
        out = eng.map_cube(img, self.get_x_map(**map_kwargs), self.get_y_map(**map_kwargs), interpolation, propagate_nan,
                           smooth_oversample_by=smooth_oversample_by, smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
                           spline_smoothing=spline_smoothing)[0]  # fmt: skip
        if spline_smoothing and interpolation not in ('nearest', 'smooth'):
            from .body_xy import BodyXY as _NativeBodyXY  # noqa: PLC0415

            _NativeBodyXY._warn_about_smoothing_fits(eng)  # (a fit scipy itself does not pin down: said, as the native class does)
        return out

    def illumination_angles_from_lonlat(self, lon, lat, *, alt=0.0, planetocentric=False):  # body.py:2295
        """
        (phase, incidence, emission) in degrees: floats for a scalar point like the reference's, arrays of the inputs'
        broadcast shape for arrays (the reference takes scalars only). `alt` there is the height of the POINT above the
        unadjusted ellipsoid (lonlat2targvec -> pgrrec), not the altitude adjustment of the backplanes, and planetocentric
        input goes through the reference's own conversion: both stay the reference's scalar SPICE path.
        """
        if alt != 0.0 or planetocentric:
            return super().illumination_angles_from_lonlat(lon, lat, alt=alt, planetocentric=planetocentric)
        scalar = np.ndim(lon) == 0 and np.ndim(lat) == 0
        lon_b, lat_b = np.broadcast_arrays(np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64))
        if lon_b.size == 0:
            return tuple(np.empty(lon_b.shape) for _ in range(3))
        o = self._hip().backplanes_map(['PHASE', 'INCIDENCE', 'EMISSION'], np.ascontiguousarray(lon_b.reshape(1, -1)),
                                       np.ascontiguousarray(lat_b.reshape(1, -1)))
        res = tuple(o[n].reshape(lon_b.shape) for n in ('PHASE', 'INCIDENCE', 'EMISSION'))
        return tuple(float(r) for r in res) if scalar else res

    def ring_plane_coordinates(self, ra, dec, only_visible=True):  # body.py:2617
        """(ring_radius, ring_longitude, ring_distance): floats for a scalar sky point like the reference's, arrays for arrays"""
        q = self._hip().radec_query(ra, dec, ring_only_visible=only_visible)  # (8, ...): q[5:8] = limb_coordinates_from_radec body.py:2040
        if np.ndim(ra) == 0 and np.ndim(dec) == 0:
            return float(q[2]), float(q[3]), float(q[4])
        return q[2], q[3], q[4]

    def _get_backplane_imgs_for_saving(self, names):  # observation.py:1269-1279
        return self._hip().backplanes_img(names, alt=self._alt_adjustment)  # all 26 planes, 2 launches


# The reference's producers and what replaces them. (method, planes, how the planes are put together: 'stack' -> one
# (.., .., n) array like the reference's `_get_*_img`, 'tuple' -> a tuple of arrays, 'one' -> the plane itself.)
_IMG_FAMILIES = (
    ('_get_lonlat_img', ('LON-GRAPHIC', 'LAT-GRAPHIC'), 'stack'),  # body_xy.py:3284
    ('_get_lonlat_centric_img', ('LON-CENTRIC', 'LAT-CENTRIC'), 'stack'),  # :3349
    ('_get_radec_img', ('RA', 'DEC'), 'stack'),  # :3413
    ('_get_km_xy_img', ('KM-X', 'KM-Y'), 'stack'),  # :3547
    ('_get_illumination_gie_img', ('PHASE', 'INCIDENCE', 'EMISSION'), 'stack'),  # :3661
    ('get_azimuth_angle_img', ('PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH'), 'AZIMUTH'),  # :3744
    ('get_local_solar_time_img', ('LON-GRAPHIC', 'LAT-GRAPHIC', 'LOCAL-SOLAR-TIME'), 'LOCAL-SOLAR-TIME'),  # :3790 (an et2lst call per pixel)
    ('get_distance_img', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'DISTANCE'),  # :3870 (from _get_state_imgs :3832)
    ('get_radial_velocity_img', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'RADIAL-VELOCITY'),  # :3898
    ('get_doppler_img', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'DOPPLER'),  # :3938
    ('_get_limb_coordinate_imgs', ('LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC', 'LIMB-DISTANCE'), 'stack'),  # :3967 (lon, lat, dist)
    ('_get_ring_plane_coordinate_imgs', ('RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE'), 'tuple'),  # :4061
)
_MAP_FAMILIES = (
    ('_get_xy_map', ('PIXEL-X', 'PIXEL-Y'), 'stack'),  # :3482
    ('_get_lonlat_centric_map', ('LON-CENTRIC', 'LAT-CENTRIC'), 'stack'),  # :3359
    ('_get_radec_map', ('RA', 'DEC'), 'stack'),  # :3423
    ('_get_km_xy_map', ('KM-X', 'KM-Y'), 'stack'),  # :3557
    ('get_phase_angle_map', ('PHASE', 'INCIDENCE', 'EMISSION'), 'PHASE'),  # :3687 (columns of _get_illumf_map :3671)
    ('get_incidence_angle_map', ('PHASE', 'INCIDENCE', 'EMISSION'), 'INCIDENCE'),  # :3709
    ('get_emission_angle_map', ('PHASE', 'INCIDENCE', 'EMISSION'), 'EMISSION'),  # :3731
    ('get_azimuth_angle_map', ('PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH'), 'AZIMUTH'),  # :3767
    ('get_local_solar_time_map', ('LOCAL-SOLAR-TIME',), 'LOCAL-SOLAR-TIME'),  # :3812
    ('get_distance_map', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'DISTANCE'),  # :3883 (from _get_state_maps :3851)
    ('get_radial_velocity_map', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'RADIAL-VELOCITY'),  # :3919
    ('get_doppler_map', ('DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'), 'DOPPLER'),  # :3951
    ('_get_limb_coordinate_maps', ('LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC', 'LIMB-DISTANCE'), 'stack'),  # :3979
    ('_get_ring_plane_coordinate_maps', ('RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE'), 'tuple'),  # :4090
)


def _assemble(planes: dict, names: tuple, how: str):
    if how == 'stack':
        return _readonly(np.stack([planes[n] for n in names], axis=-1))
    if how == 'tuple':
        return tuple(planes[n] for n in names)
    return planes[how]


def _make_img_override(method: str, names: tuple, how: str):
    def override(self):
        key = ('_hip_out', method, float(getattr(self, '_alt_adjustment', 0.0)))
        if key not in self._cache:  # (the assembled form is cached too: `get_lon_img()[...] is get_lon_img()[...]`-style identity holds)
            self._cache[key] = _assemble(self._hip_img_family(names), names, how)
        return self._cache[key]

    override.__name__ = method
    override.__doc__ = f'planetmapper.BodyXY.{method}: {", ".join(names)} from one engine launch'
    return override


def _make_map_override(method: str, names: tuple, how: str):
    def override(self, **map_kwargs):
        return _assemble(self._hip_map_family(names, map_kwargs), names, how)

    override.__name__ = method
    override.__doc__ = f'planetmapper.BodyXY.{method}: {", ".join(names)} of the map grid from one engine launch'
    return override


for _method, _names, _how in _IMG_FAMILIES:
    setattr(HipBackplanes, _method, _make_img_override(_method, _names, _how))
for _method, _names, _how in _MAP_FAMILIES:
    setattr(HipBackplanes, _method, _make_map_override(_method, _names, _how))
