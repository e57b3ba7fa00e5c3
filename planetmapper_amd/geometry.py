"""
Host-side geometry block.

`pm_geometry` (include/planetmapper_hip.h) holds every constant the per-pixel kernels
need about one body at one epoch. The reference computes the same quantities once per
`Body` with SPICE calls (`planetmapper/base.py:795-839`, `planetmapper/body.py:501-606`);
`GeometryBuilder` computes them SPICE-free from an `ephem.Ephemeris` + `RotationModel`.
A deployment that has spiceypy can fill the same structure from a reference `Body`
instance instead (recipe in INTEGRATION.md); the kernels only ever see the block.
"""

from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from .ephem import CLIGHT, Ephemeris, RotationModel, rotate

_D3 = ctypes.c_double * 3
_D9 = ctypes.c_double * 9


class PMGeometry(ctypes.Structure):
    """ctypes mirror of `pm_geometry` (field order/size must match the header)."""

    _fields_ = [
        ('et', ctypes.c_double),
        ('lt_c', ctypes.c_double),
        ('clight', ctypes.c_double),
        ('radii', _D3),
        ('T0', _D3),
        ('VT', _D3),
        ('AT', _D3),
        ('VO', _D3),
        ('DVT', _D3),
        ('DAT', _D3),
        ('ts0', ctypes.c_double),
        ('S0', _D3),
        ('VS', _D3),
        ('AS', _D3),
        ('R0', _D9),
        ('wdot', ctypes.c_double),
        ('sub_sp', _D3),
        ('sub_ray', _D3),
        ('sub_obsvec', _D3),
        ('sub_et', ctypes.c_double),
        ('sub_dist', ctypes.c_double),
        ('ring_n', _D3),
        ('ring_k', ctypes.c_double),
        ('M', _D9),
        ('diameter_arcsec', ctypes.c_double),
        ('km_per_arcsec', ctypes.c_double),
        ('np_angle_rad', ctypes.c_double),
        ('lst_sun_lon', ctypes.c_double),
        ('WP', _D3),
        ('west_positive', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]

    _VEC_FIELDS = (
        'radii', 'T0', 'VT', 'AT', 'VO', 'DVT', 'DAT', 'S0', 'VS', 'AS', 'R0',
        'sub_sp', 'sub_ray', 'sub_obsvec', 'ring_n', 'M',
    )

    def to_dict(self) -> dict:
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            if name in self._VEC_FIELDS:
                out[name] = [float(x).hex() for x in v]
            elif isinstance(v, float):
                out[name] = float(v).hex()
            else:
                out[name] = int(v)
        return out

    @classmethod
    def from_dict(cls, d: dict) -> 'PMGeometry':
        g = cls()
        for name, _ in cls._fields_:
            v = d[name]
            if name in cls._VEC_FIELDS:
                arr = getattr(g, name)
                for i, x in enumerate(v):
                    arr[i] = float.fromhex(x) if isinstance(x, str) else float(x)
            elif isinstance(v, str):
                setattr(g, name, float.fromhex(v))
            else:
                setattr(g, name, v)
        return g

    def copy(self) -> 'PMGeometry':
        g = PMGeometry()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(self), ctypes.sizeof(PMGeometry))
        return g


class PMDisc(ctypes.Structure):
    """ctypes mirror of `pm_disc`."""

    _fields_ = [
        ('x0', ctypes.c_double),
        ('y0', ctypes.c_double),
        ('r0', ctypes.c_double),
        ('rotation_rad', ctypes.c_double),
        ('nx', ctypes.c_int32),
        ('ny', ctypes.c_int32),
        ('optimize_speed', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]


# ----------------------------------------------------------------------------------
# CSPICE-style scalar helpers (host only)
# ----------------------------------------------------------------------------------
def radrec(r: float, ra: float, dec: float) -> np.ndarray:
    return np.array(
        [r * math.cos(ra) * math.cos(dec), r * math.sin(ra) * math.cos(dec), r * math.sin(dec)]
    )


def recrad(v: Sequence[float]) -> tuple[float, float, float]:
    x, y, z = (float(c) for c in v)
    big = max(abs(x), abs(y), abs(z))
    if big == 0.0:
        return 0.0, 0.0, 0.0
    xs, ys, zs = x / big, y / big, z / big
    r = big * math.sqrt(xs * xs + ys * ys + zs * zs)
    lat = math.atan2(zs, math.sqrt(xs * xs + ys * ys))
    lon = 0.0 if (xs == 0.0 and ys == 0.0) else math.atan2(ys, xs)
    if lon < 0.0:
        lon += 2.0 * math.pi
    return r, lon, lat


def surfpt(p: np.ndarray, u: np.ndarray, radii: Sequence[float]) -> np.ndarray | None:
    """First intersection of ray (p, u) with the ellipsoid, or None (`surfpt_c`)."""
    ax = np.asarray(radii, dtype=float)
    X = u / ax
    Y = p / ax
    xx = float(X @ X)
    if xx == 0.0:
        return None
    yx = float(Y @ X)
    P = Y - (yx / xx) * X
    pmag = float(np.linalg.norm(P))
    ymag = float(np.linalg.norm(Y))
    ux = X / math.sqrt(xx)
    if ymag > 1.0:
        if pmag > 1.0 or yx > 0.0:
            return None
        sign = -1.0
    elif ymag == 1.0:
        return np.array(p, dtype=float)
    else:
        sign = 1.0
    s = sign * math.sqrt(max(0.0, 1.0 - pmag * pmag))
    return (P + s * ux) * ax


def stelab(pobj: np.ndarray, vobs: np.ndarray) -> np.ndarray:
    """Stellar aberration correction of `pobj` for observer velocity `vobs` (`stelab_c`)."""
    u = pobj / np.linalg.norm(pobj)
    vbyc = vobs / CLIGHT
    h = np.cross(u, vbyc)
    sinphi = float(np.linalg.norm(h))
    if sinphi == 0.0:
        return np.array(pobj, dtype=float)
    phi = math.asin(sinphi)
    # vrotv: rotate pobj about axis h by phi
    axis = h / sinphi
    p_par = axis * float(pobj @ axis)
    p_perp = pobj - p_par
    return p_par + math.cos(phi) * p_perp + math.sin(phi) * np.cross(axis, p_perp)


# ----------------------------------------------------------------------------------
@dataclass
class GeometryBuilder:
    """
    SPICE-free equivalent of `Body.__init__` for the quantities on the hot path.

    Args:
        ephemeris: Chebyshev ephemeris covering target, Sun and (optionally) observer.
        rotation: IAU rotation model + radii of the target.
        target_id: NAIF id of the target centre (e.g. 599).
        sun_id: NAIF id of the illumination source (10).
    """

    ephemeris: Ephemeris
    rotation: RotationModel
    target_id: int
    sun_id: int = 10

    # -- helpers -------------------------------------------------------------------
    def _ptarget(self, t: float) -> np.ndarray:
        return self.ephemeris.ssb_state(self.target_id, t)[0]

    def build(
        self,
        et: float,
        *,
        observer_id: int | None = None,
        observer_velocity: Sequence[float] | None = None,
        target_ra_dec_dist_lt: tuple[float, float, float, float] | None = None,
    ) -> PMGeometry:
        """
        Build the geometry block for epoch `et` (TDB seconds past J2000).

        Either give `observer_id` (a body in the ephemeris, e.g. 399) or, for observers
        whose ephemeris cannot be evaluated here (e.g. HST, SPK type 10), the apparent
        target `(ra_deg, dec_deg, distance_km, light_time_s)` as published by the
        reference in its FITS headers (`TARGET RA/DEC`, `DISTANCE`, `LIGHT-TIME`,
        `planetmapper/observation.py:1022-1060`) plus the observer SSB velocity.
        """
        eph, rot = self.ephemeris, self.rotation
        g = PMGeometry()
        g.et = et
        g.clight = CLIGHT
        radii = np.array(rot.radii, dtype=float)
        for i in range(3):
            g.radii[i] = radii[i]

        # --- target centre, spkezr(target, et, 'J2000', 'CN', observer) base.py:828-839
        if observer_id is not None:
            p_obs, v_obs, _ = eph.ssb_state(observer_id, et)
            lt = 0.0
            for _ in range(12):
                t0 = et - lt
                new_lt = float(np.linalg.norm(self._ptarget(t0) - p_obs)) / CLIGHT
                if new_lt == lt:
                    break
                lt = new_lt
            t0 = et - lt
            T0 = self._ptarget(t0) - p_obs
            if observer_velocity is not None:
                v_obs = np.asarray(observer_velocity, dtype=float)
        else:
            if target_ra_dec_dist_lt is None:
                raise ValueError('need observer_id or target_ra_dec_dist_lt')
            ra, dec, dist, lt = target_ra_dec_dist_lt
            t0 = et - lt
            T0 = radrec(dist, math.radians(ra), math.radians(dec))
            p_obs = self._ptarget(t0) - T0
            v_obs = (
                np.zeros(3)
                if observer_velocity is None
                else np.asarray(observer_velocity, dtype=float)
            )
        g.lt_c = lt
        # The kernels propagate the target as T0 + VT d + AT d^2 / 2 where the reference has CSPICE
        # re-evaluate its POSITION at et - lt: VT / AT are the time derivatives of the position series.
        # A state (spkezr, for the radial velocity) takes its velocity from the segment's own velocity
        # polynomial where there is one (SPK type 3): DVT is the difference (2e-6 km/s for 599 in jup120).
        p_t0, v_state, a_state = eph.ssb_state(self.target_id, t0)
        _, v_t0, a_t0 = eph.ssb_motion(self.target_id, t0)
        for i in range(3):
            g.T0[i] = T0[i]
            g.VT[i] = v_t0[i]
            g.AT[i] = a_t0[i]
            g.DVT[i] = v_state[i] - v_t0[i]
            g.DAT[i] = a_state[i] - a_t0[i]
            g.VO[i] = v_obs[i]

        # --- Sun as seen from the target centre at t0 (one-way light time)
        lts = 0.0
        for _ in range(12):
            ps = eph.ssb_state(self.sun_id, t0 - lts)[0]
            new = float(np.linalg.norm(ps - p_t0)) / CLIGHT
            if new == lts:
                break
            lts = new
        ts0 = t0 - lts
        p_s, v_s, a_s = eph.ssb_motion(self.sun_id, ts0)  # (positions only are ever asked of the Sun)
        g.ts0 = ts0
        for i in range(3):
            g.S0[i] = p_s[i] - p_t0[i]
            g.VS[i] = v_s[i]
            g.AS[i] = a_s[i]

        # --- orientation
        R0 = rot.matrix(t0)
        wdot = rot.body_z_rate(t0)  # (dW/dt + the pole's precession along the pole: what pxform at t0 + d amounts to)
        for i in range(9):
            g.R0[i] = R0.flat[i]
        g.wdot = wdot
        drift = rot.pole_drift(t0)  # (STATE planes only: see pm_geometry.WP)
        for i in range(3):
            g.WP[i] = drift[i]

        def rot_at(t: float) -> np.ndarray:
            ang = wdot * (t - t0)
            return rotate(ang, 3) @ R0

        def target_at(t: float) -> np.ndarray:
            d = t - t0
            return T0 + v_t0 * d + 0.5 * a_t0 * d * d

        # --- positive longitude direction body.py:526-535
        west = rot.prograde and self.target_id not in (10, 301, 399)
        g.west_positive = 1 if west else 0

        # --- sub-observer point: subpnt('INTERCEPT/ELLIPSOID', 'CN') body.py:538-555
        lt_s = lt
        sp = None
        obs_b = None
        for _ in range(12):
            te = et - lt_s
            Rk = rot_at(te)
            obs_b = -(Rk @ target_at(te))
            sp = surfpt(obs_b, -obs_b, radii)
            assert sp is not None
            new = float(np.linalg.norm(sp - obs_b)) / CLIGHT
            if abs(new - lt_s) <= 1e-17 * abs(et - new):
                lt_s = new
                break
            lt_s = new
        sub_et = et - lt_s
        Rk = rot_at(sub_et)
        obs_b = -(Rk @ target_at(sub_et))
        sp = surfpt(obs_b, -obs_b, radii)
        sub_ray = sp - obs_b
        sub_obsvec = Rk.T @ sub_ray  # _rayvec2obsvec body.py:950-962
        for i in range(3):
            g.sub_sp[i] = sp[i]
            g.sub_ray[i] = sub_ray[i]
            g.sub_obsvec[i] = sub_obsvec[i]
        g.sub_et = sub_et
        g.sub_dist = float(np.linalg.norm(sub_ray))

        # PM's own body-fixed -> observer transform, body.py:917-948
        def targvec2obsvec(tv: np.ndarray) -> np.ndarray:
            off = tv - sp
            dist_off = float(np.linalg.norm(sub_ray + off)) - g.sub_dist
            t = sub_et - dist_off / CLIGHT
            return sub_obsvec + rot_at(t).T @ off

        # --- target diameter body.py:574-577
        target_distance = lt * CLIGHT
        r_eq = float(radii[0])
        g.diameter_arcsec = float(
            2.0 * 60.0 * 60.0 * np.rad2deg(np.arcsin(r_eq / target_distance))
        )
        g.km_per_arcsec = (2.0 * r_eq) / g.diameter_arcsec

        # --- obsvec -> angular matrix body.py:1317-1343 (origin = target RA/Dec)
        _, tra, tdec = recrad(T0)
        tra_deg, tdec_deg = np.rad2deg(tra), np.rad2deg(tdec)  # BodyBase.target_ra/dec
        origin = radrec(1.0, float(np.deg2rad(tra_deg)), float(np.deg2rad(tdec_deg)))
        _, ra_angle, _ = recrad(origin)
        ra_matrix = rotate(ra_angle, 3)
        _, _, dec_angle = recrad(ra_matrix @ origin)
        dec_matrix = rotate(-dec_angle, 2)
        M = rotate(float(np.deg2rad(0.0)), 1) @ dec_matrix @ ra_matrix
        for i in range(9):
            g.M[i] = M.flat[i]

        def obsvec2angular(ov: np.ndarray) -> tuple[float, float]:
            _, x, y = recrad(M @ ov)
            x = (-np.rad2deg(x)) % 360.0
            if x > 180.0:
                x -= 360.0
            return float(x * 3600.0), float(np.rad2deg(y) * 3600.0)

        # --- ring (equatorial) plane, body.py:580-588
        f = (radii[0] - radii[2]) / radii[0]
        np_targvec = np.array([0.0, 0.0, radii[0] * (1.0 - f)])  # pgrrec(0, pi/2, 0)
        # pgrrec evaluates cos(pi/2) != 0 exactly; keep the same tiny x like georec
        clat = math.cos(math.pi / 2.0)
        slat = math.sin(math.pi / 2.0)
        a_eq = radii[0]
        b_pol = a_eq * (1.0 - f)
        big = max(abs(a_eq * clat), abs(b_pol * slat))
        xs, ys = a_eq * clat / big, b_pol * slat / big
        scale = 1.0 / (big * math.sqrt(xs * xs + ys * ys))
        np_targvec = np.array(
            [scale * a_eq * a_eq * clat, 0.0, scale * b_pol * b_pol * slat]
        )
        np_obsvec = targvec2obsvec(np_targvec)
        normal = np_obsvec - T0
        nhat = normal / np.linalg.norm(normal)
        k = float(nhat @ T0)
        if k < 0.0:  # nvp2pl_c keeps the constant non-negative
            k = -k
            nhat = -nhat
        for i in range(3):
            g.ring_n[i] = nhat[i]
        g.ring_k = k

        # --- north pole angle, body.py:2985-3009
        _, npra, npdec = recrad(np_obsvec)
        np_ra_deg, np_dec_deg = np.rad2deg(npra), np.rad2deg(npdec)
        np_x, np_y = obsvec2angular(
            radrec(1.0, float(np.deg2rad(np_ra_deg)), float(np.deg2rad(np_dec_deg)))
        )
        t_x, t_y = obsvec2angular(origin)
        theta = -np.arctan2(t_x - np_x, np_y - t_y)
        theta = np.rad2deg(theta) % 360.0
        if theta > 180:
            theta -= 360
        g.np_angle_rad = float(np.deg2rad(theta))

        # --- local solar time: Sun longitude, et2lst(et - lt, body, lon) body.py:2364-2374
        # spkez(10, t0, frame, 'LT+S', body): one-way light time + stellar aberration
        lt1 = 0.0
        for _ in range(3):
            ps = eph.ssb_state(self.sun_id, t0 - lt1)[0]
            lt1 = float(np.linalg.norm(ps - p_t0)) / CLIGHT
        ps = eph.ssb_state(self.sun_id, t0 - lt1)[0]
        sun_app = stelab(ps - p_t0, v_t0)
        sun_b = R0 @ sun_app
        g.lst_sun_lon = math.atan2(sun_b[1], sun_b[0])
        return g

    # -- scalars the reference publishes, for validating a block ---------------------
    @staticmethod
    def describe(g: PMGeometry) -> dict:
        """Derived scalars comparable with `Body` attributes / FITS header cards."""
        radii = np.array(g.radii[:])
        T0 = np.array(g.T0[:])
        _, ra, dec = recrad(T0)
        sp = np.array(g.sub_sp[:])
        a, c = radii[0], radii[2]
        f = (a - c) / a
        lon = math.atan2(sp[1], sp[0])
        if g.west_positive:
            lon = -lon
        lon %= 2 * math.pi
        lat = math.atan2(sp[2] / (1 - f) ** 2, math.hypot(sp[0], sp[1]))

        # sub-solar point, spice.subslr('INTERCEPT/ELLIPSOID', ..., 'CN') (body.py:553-560):
        # where the centre -> Sun line leaves the surface, body orientation and Sun direction
        # taken at et - (light time from the observer to that surface point)
        R0 = np.array(g.R0[:]).reshape(3, 3)
        VT, AT = np.array(g.VT[:]), np.array(g.AT[:])
        S0, VS, AS = np.array(g.S0[:]), np.array(g.VS[:]), np.array(g.AS[:])
        t0 = g.et - g.lt_c
        lt = g.lt_c
        ssp = np.zeros(3)
        for _ in range(4):
            d = (g.et - lt) - t0
            ang = g.wdot * d
            cz, sz = math.cos(ang), math.sin(ang)
            R = np.array([[cz, sz, 0.0], [-sz, cz, 0.0], [0.0, 0.0, 1.0]]) @ R0
            moved = VT * d + 0.5 * AT * d * d  # target centre since t0 (SSB)
            sun = S0 + VS * d + 0.5 * AS * d * d - moved  # Sun wrt target centre, same light-time lag
            u = R @ sun
            u /= np.linalg.norm(u)
            ssp = u / math.sqrt((u[0] / radii[0]) ** 2 + (u[1] / radii[1]) ** 2 + (u[2] / radii[2]) ** 2)
            lt = float(np.linalg.norm(T0 + moved + R.T @ ssp)) / g.clight
        slon = math.atan2(ssp[1], ssp[0])
        if g.west_positive:
            slon = -slon
        slon %= 2 * math.pi
        slat = math.atan2(ssp[2] / (1 - f) ** 2, math.hypot(ssp[0], ssp[1]))
        return {
            'subsol_lon': float(np.rad2deg(slon)),
            'subsol_lat': float(np.rad2deg(slat)),
            'target_ra': float(np.rad2deg(ra)),
            'target_dec': float(np.rad2deg(dec)),
            'target_distance': g.lt_c * g.clight,
            'target_light_time': g.lt_c,
            'target_diameter_arcsec': g.diameter_arcsec,
            'km_per_arcsec': g.km_per_arcsec,
            'subpoint_distance': g.sub_dist,
            'subpoint_lon': float(np.rad2deg(lon)),
            'subpoint_lat': float(np.rad2deg(lat)),
            'north_pole_angle': float(np.rad2deg(g.np_angle_rad)),
        }
