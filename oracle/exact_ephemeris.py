"""
TEST INFRASTRUCTURE ONLY (imported by tests/ alone, like everything under oracle/).

A second CPU restatement of the per-pixel path for a handful of planes that RE-EVALUATES the
ephemeris (Chebyshev SPK types 2 / 3) and the IAU orientation model at EVERY light-time epoch - what
CSPICE does for the reference inside `sincpt` (planetmapper/body.py:1008-1020), `illumf`
(:1925-1934), `spkcpt` (:2833-2842) and `pxfrm2` (:940, 955, 998) - instead of propagating the
geometry block's motion model T0 + VT d + AT d^2 / 2, Rz(wdot d) R0, S0 + VS ds + AS ds^2 / 2
(include/planetmapper_hip.h, oracle/pm_oracle.c:121-147). The HIP kernels, the C oracle and its
binary128 build all share that model; this module is what they are held against for geometries the
reference's golden files do not cover (Saturn, other epochs, other bodies): SURVEY.md Appendix A
with R(t), T(t), Sun(t) taken from the kernel data directly. Scalar Python loops: for ~1e3 pixels.

Also here: exact (mpmath) evaluation of Chebyshev displacements and of rotation increments, so that
the TRUNCATION error of the motion model can be told apart from the rounding of 1e9-km vectors.
"""

from __future__ import annotations

import math

import numpy as np

from planetmapper_amd.ephem import CLIGHT, JULIAN_CENTURY_S, SPD, Ephemeris, RotationModel, rotate
from planetmapper_amd.geometry import PMGeometry, recrad, surfpt

TWO_PI = 2.0 * math.pi


# ---------------------------------------------------------------------------- exact arithmetic (mpmath)
def displacement_exact(eph: Ephemeris, body: int, t_a: float, t_b: float):
    """P_body(t_b) - P_body(t_a) wrt the SSB, each Chebyshev series summed in 60-digit arithmetic."""
    import mpmath as mp

    mp.mp.dps = 60

    def pos(t):
        out = [mp.mpf(0)] * 3
        cur = body
        while cur != 0:
            seg = eph._find(cur, t)
            rec = seg._record(t)
            ncomp = 3 if seg.spk_type == 2 else 6
            n = (len(rec) - 2) // ncomp
            s = (mp.mpf(t) - mp.mpf(float(rec[0]))) / mp.mpf(float(rec[1]))
            tk = [mp.mpf(1), s]
            for k in range(2, n):
                tk.append(2 * s * tk[k - 1] - tk[k - 2])
            for i in range(3):
                c = rec[2 + i * n : 2 + (i + 1) * n]
                out[i] = out[i] + sum(mp.mpf(float(ck)) * tk[k] for k, ck in enumerate(c))
            cur = seg.center
        return out

    a, b = pos(t_a), pos(t_b)
    return np.array([float(b[i] - a[i]) for i in range(3)])


def rotation_exact(rot: RotationModel, t: float):
    """J2000 -> body-fixed matrix of the IAU model at t with the angles reduced in 60-digit arithmetic
    (a double W of 1.6e6 deg carries 4e-12 rad of rounding on its own), as an mpmath matrix."""
    import mpmath as mp

    mp.mp.dps = 60
    tc = mp.mpf(t) / mp.mpf(JULIAN_CENTURY_S)
    dd = mp.mpf(t) / mp.mpf(SPD)
    ra = sum(mp.mpf(c) * tc**k for k, c in enumerate(rot.pole_ra))
    dec = sum(mp.mpf(c) * tc**k for k, c in enumerate(rot.pole_dec))
    w = sum(mp.mpf(c) * dd**k for k, c in enumerate(rot.pm))
    for i in range(len(rot.nut_prec_angles) // 2):
        th = mp.radians(mp.mpf(rot.nut_prec_angles[2 * i]) + mp.mpf(rot.nut_prec_angles[2 * i + 1]) * tc)
        if i < len(rot.nut_prec_ra):
            ra += mp.mpf(rot.nut_prec_ra[i]) * mp.sin(th)
        if i < len(rot.nut_prec_dec):
            dec += mp.mpf(rot.nut_prec_dec[i]) * mp.cos(th)
        if i < len(rot.nut_prec_pm):
            w += mp.mpf(rot.nut_prec_pm[i]) * mp.sin(th)

    def r(angle, axis):
        c, s = mp.cos(angle), mp.sin(angle)
        if axis == 1:
            return mp.matrix([[1, 0, 0], [0, c, s], [0, -s, c]])
        return mp.matrix([[c, s, 0], [-s, c, 0], [0, 0, 1]])

    return r(mp.radians(w), 3) * r(mp.pi / 2 - mp.radians(dec), 1) * r(mp.pi / 2 + mp.radians(ra), 3)


def rotation_angle_between(A, B) -> float:
    """angle (rad) of the rotation A B^T for two mpmath / numpy rotation matrices"""
    import mpmath as mp

    A, B = mp.matrix(A.tolist() if hasattr(A, 'tolist') and not isinstance(A, mp.matrix) else A), mp.matrix(
        B.tolist() if hasattr(B, 'tolist') and not isinstance(B, mp.matrix) else B
    )
    D = A * B.T
    # small angles: from the antisymmetric part, not from the trace
    ax = mp.sqrt((D[2, 1] - D[1, 2]) ** 2 + (D[0, 2] - D[2, 0]) ** 2 + (D[1, 0] - D[0, 1]) ** 2) / 2
    return float(mp.asin(min(ax, mp.mpf(1))))


# ---------------------------------------------------------------------------- per-pixel path, direct evaluation
def vsep(u, v) -> float:
    """vsep_c"""
    nu, nv = np.linalg.norm(u), np.linalg.norm(v)
    if nu == 0.0 or nv == 0.0:
        return 0.0
    u, v = u / nu, v / nv
    d = float(u @ v)
    if d > 0.0:
        return 2.0 * math.asin(0.5 * np.linalg.norm(u - v))
    if d < 0.0:
        return math.pi - 2.0 * math.asin(0.5 * np.linalg.norm(u + v))
    return math.pi / 2.0


def geodetic(tv, a: float, f: float):
    """recpgr_c core for an arbitrary point: (east longitude, geodetic latitude, altitude), fixed-point
    iteration on the latitude (converges in a few steps away from the centre)"""
    x, y, z = (float(c) for c in tv)
    rho = math.hypot(x, y)
    lon = 0.0 if (x == 0.0 and y == 0.0) else math.atan2(y, x)
    e2 = f * (2.0 - f)
    lat = math.atan2(z, rho * (1.0 - e2))
    alt = 0.0
    for _ in range(60):
        n = a / math.sqrt(1.0 - e2 * math.sin(lat) ** 2)
        alt = rho / math.cos(lat) - n if abs(math.cos(lat)) > 1e-3 else z / math.sin(lat) - n * (1.0 - e2)
        new = math.atan2(z, rho * (1.0 - e2 * n / (n + alt)))
        if abs(new - lat) < 1e-17:
            lat = new
            break
        lat = new
    return lon, lat, alt


class ExactBody:
    """
    One target / observer / epoch with T(t), R(t), Sun(t) evaluated from the kernel data at every epoch.
    `g` supplies what does not depend on the motion model: et, radii, the obsvec -> angular matrix M,
    the apparent diameter, the longitude convention; and, for observers without an ephemeris (HST:
    apparent RA / Dec / distance from the reference's header), the observer's position through T0.
    """

    def __init__(self, eph: Ephemeris, rot: RotationModel, target_id: int, g: PMGeometry, observer_id: int | None = None,
                 sun_id: int = 10) -> None:
        self.eph, self.rot, self.target, self.sun = eph, rot, target_id, sun_id
        self.g = g
        self.et = g.et
        self.radii = np.array(g.radii[:])
        self.a = float(self.radii[0])
        self.f = (self.radii[0] - self.radii[2]) / self.radii[0]
        self.M = np.array(g.M[:]).reshape(3, 3)
        if observer_id is not None:
            self.p_obs = eph.ssb_state(observer_id, self.et)[0]
        else:
            self.p_obs = eph.ssb_state(target_id, g.et - g.lt_c)[0] - np.array(g.T0[:])
        self.west = bool(g.west_positive)
        self._subpoint()

    # -- state from the kernels -----------------------------------------------------
    def T(self, t: float) -> np.ndarray:
        return self.eph.ssb_state(self.target, t)[0] - self.p_obs

    def R(self, t: float) -> np.ndarray:
        return self.rot.matrix(t)

    # -- Body.__init__ sub-observer constants (subpnt INTERCEPT/ELLIPSOID CN, body.py:538-555)
    def _subpoint(self) -> None:
        lt = self.g.lt_c
        for _ in range(12):
            te = self.et - lt
            R = self.R(te)
            obs_b = -(R @ self.T(te))
            sp = surfpt(obs_b, -obs_b, self.radii)
            new = float(np.linalg.norm(sp - obs_b)) / CLIGHT
            done = abs(new - lt) <= 1e-17 * abs(self.et - new)
            lt = new
            if done:
                break
        self.sub_et = self.et - lt
        R = self.R(self.sub_et)
        obs_b = -(R @ self.T(self.sub_et))
        self.sub_sp = surfpt(obs_b, -obs_b, self.radii)
        self.sub_ray = self.sub_sp - obs_b
        self.sub_dist = float(np.linalg.norm(self.sub_ray))
        self.sub_obsvec = R.T @ self.sub_ray

    # -- PM's own transforms (body.py:917-948, 972-1006), pxfrm2 evaluated directly
    def obsvec2targvec(self, ov: np.ndarray) -> np.ndarray:
        off = ov - self.sub_obsvec
        d = float(np.linalg.norm(-self.sub_ray + off)) - self.sub_dist
        t = self.sub_et - d / CLIGHT
        return self.sub_sp + self.R(t) @ off

    def targvec2obsvec(self, tv: np.ndarray) -> np.ndarray:
        off = tv - self.sub_sp
        d = float(np.linalg.norm(self.sub_ray + off)) - self.sub_dist
        t = self.sub_et - d / CLIGHT
        return self.sub_obsvec + self.R(t).T @ off

    # -- pixel -> ray (body_xy.py:354-377, body.py:1363-1373)
    def ray(self, x: float, y: float, x0: float, y0: float, r0: float, rotation_rad: float) -> np.ndarray:
        s = self.g.diameter_arcsec / (2.0 * r0)
        th = -rotation_rad
        c, sn = math.cos(th), math.sin(th)
        a2 = np.array([[s * c, s * sn], [-s * sn, s * c]])
        ax, ay = a2 @ np.array([x, y]) - a2 @ np.array([x0, y0])
        ra, dec = -math.radians(ax / 3600.0), math.radians(ay / 3600.0)
        v = np.array([math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec)])
        return self.M.T @ v

    # -- sincpt_c ELLIPSOID CN (body.py:1008-1020): target and orientation re-evaluated at every pass
    def sincpt(self, ray: np.ndarray):
        lt = self.g.lt_c
        sp = None
        for _ in range(10):
            te = self.et - lt
            R = self.R(te)
            obs_b = -(R @ self.T(te))
            sp = surfpt(obs_b, R @ ray, self.radii)
            if sp is None:
                return None, None
            new = float(np.linalg.norm(sp - obs_b)) / CLIGHT
            done = abs(new - lt) <= 1e-17 * abs(self.et - new)
            lt = new
            if done:
                break
        return sp, lt

    def lonlat(self, sp: np.ndarray) -> tuple[float, float]:
        """recpgr_c on the surface (body.py:1022-1036), degrees"""
        lam = 0.0 if (sp[0] == 0.0 and sp[1] == 0.0) else math.atan2(sp[1], sp[0])
        lon = (-lam if self.west else lam) % TWO_PI
        lat = math.atan2(sp[2] / (1.0 - self.f) ** 2, math.hypot(sp[0], sp[1]))
        return math.degrees(lon), math.degrees(lat)

    # -- illumf_c (Sun, CN; body.py:1915-1935): both light times solved against the kernels
    def illum(self, sp: np.ndarray) -> tuple[float, float, float, float]:
        lt = float(np.linalg.norm(self.T(self.et - self.g.lt_c))) / CLIGHT
        pos = None
        for _ in range(10):
            te = self.et - lt
            pos = self.T(te) + self.R(te).T @ sp
            new = float(np.linalg.norm(pos)) / CLIGHT
            done = abs(new - lt) <= 1e-17 * abs(self.et - new)
            lt = new
            if done:
                break
        te = self.et - lt
        R = self.R(te)
        q = self.eph.ssb_state(self.target, te)[0] + R.T @ sp  # the surface point wrt the SSB
        lts = 0.0
        sun = None
        for _ in range(10):
            sun = self.eph.ssb_state(self.sun, te - lts)[0] - q
            new = float(np.linalg.norm(sun)) / CLIGHT
            done = abs(new - lts) <= 1e-17 * abs(te - new)
            lts = new
            if done:
                break
        sun_b = R @ sun
        obs_b = -(R @ pos)
        n = sp / self.radii**2
        return (math.degrees(vsep(sun_b, obs_b)), math.degrees(vsep(sun_b, n)), math.degrees(vsep(obs_b, n)), lt * CLIGHT)

    # -- ring plane (body.py:583-588, 2577-2615 with only_visible=False)
    def ring_plane(self):
        np_obs = self.targvec2obsvec(np.array([0.0, 0.0, self.radii[2]]))
        T0 = self.T(self.et - self.g.lt_c)
        n = np_obs - T0
        n = n / np.linalg.norm(n)
        k = float(n @ T0)
        return (-n, -k) if k < 0.0 else (n, k)

    def ring(self, ray: np.ndarray, plane) -> tuple[float, float, float]:
        n, k = plane
        # (the reference rebuilds the ray from RA / Dec in degrees, body_xy.py:3262-3271)
        _, ra, dec = recrad(ray)
        ra, dec = math.radians(math.degrees(ra)), math.radians(math.degrees(dec))
        ray = np.array([math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec)])
        den = float(n @ ray)
        if den == 0.0 or k / den <= 0.0:
            return math.nan, math.nan, math.nan
        ip = (k / den) * ray
        tv = self.obsvec2targvec(ip)
        lam, _, alt = geodetic(tv, self.a, self.f)
        lon = (-lam if self.west else lam) % TWO_PI
        return alt + self.a, math.degrees(lon), float(np.linalg.norm(ip))

    # -- the planes of a set of pixels
    def planes(self, pixels, x0: float, y0: float, r0: float, rotation_rad: float, rings: bool = False) -> dict[str, np.ndarray]:
        names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'DISTANCE']
        if rings:
            names += ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
        out = {n: np.full(len(pixels), np.nan) for n in names}
        plane = self.ring_plane() if rings else None
        for i, (x, y) in enumerate(pixels):
            ray = self.ray(float(x), float(y), x0, y0, r0, rotation_rad)
            sp, _ = self.sincpt(ray)
            dist = math.nan
            if sp is not None:
                out['LON-GRAPHIC'][i], out['LAT-GRAPHIC'][i] = self.lonlat(sp)
                ph, inc, emi, dist = self.illum(sp)
                out['PHASE'][i], out['INCIDENCE'][i], out['EMISSION'][i], out['DISTANCE'][i] = ph, inc, emi, dist
            if rings:
                rr, rl, rd = self.ring(ray, plane)
                if rd > dist:  # hidden behind the disc (NaN compares false), body_xy.py:4077-4080
                    rr = rl = rd = math.nan
                out['RING-RADIUS'][i], out['RING-LON-GRAPHIC'][i], out['RING-DISTANCE'][i] = rr, rl, rd
        return out


    # -- the map direction (body_xy.py:3227-3300, 3419-3491, 3667-3675)
    def pgrrec(self, lon_deg: float, lat_deg: float) -> np.ndarray:
        """pgrrec_c on the surface (body.py:903-910)"""
        lon, lat = math.radians(lon_deg), math.radians(lat_deg)
        lam = -lon if self.west else lon
        e2 = self.f * (2.0 - self.f)
        n = self.a / math.sqrt(1.0 - e2 * math.sin(lat) ** 2)
        return np.array([n * math.cos(lat) * math.cos(lam), n * math.cos(lat) * math.sin(lam), n * (1.0 - self.f) ** 2 * math.sin(lat)])

    def map_cells(self, lon_deg, lat_deg, x0: float, y0: float, r0: float, rotation_rad: float, nx: int, ny: int) -> dict[str, np.ndarray]:
        """
        For each (lon, lat): the map-space planes EMISSION / INCIDENCE / PHASE (defined everywhere), RA / DEC and
        the pixel coordinates PIXEL-X / PIXEL-Y (NaN where not visible / outside the frame): pgrrec -> illumf
        (kernel data re-evaluated per epoch) -> PM's _targvec2obsvec (pxfrm2 evaluated directly) -> recrad ->
        degrees -> radrec -> _obsvec2angular -> inverse affine -> in-frame test.
        """
        lon_deg, lat_deg = np.asarray(lon_deg, dtype=float), np.asarray(lat_deg, dtype=float)
        shape = lon_deg.shape
        names = ['PHASE', 'INCIDENCE', 'EMISSION', 'RA', 'DEC', 'PIXEL-X', 'PIXEL-Y']
        out = {n: np.full(lon_deg.size, np.nan) for n in names}
        s = self.g.diameter_arcsec / (2.0 * r0)
        th = -rotation_rad
        c, sn = math.cos(th), math.sin(th)
        a2 = np.array([[s * c, s * sn], [-s * sn, s * c]])
        a2i = np.linalg.inv(a2)
        off = a2 @ np.array([x0, y0])
        for i, (lo, la) in enumerate(zip(lon_deg.ravel(), lat_deg.ravel())):
            if not (math.isfinite(lo) and math.isfinite(la)):
                continue
            tv = self.pgrrec(lo % 360.0, la)
            ph, inc, emi, _ = self.illum(tv)
            out['PHASE'][i], out['INCIDENCE'][i], out['EMISSION'][i] = ph, inc, emi
            if not emi < 90.0:
                continue
            ov = self.targvec2obsvec(tv)
            _, ra, dec = recrad(ov)
            ra_d, dec_d = math.degrees(ra), math.degrees(dec)
            out['RA'][i], out['DEC'][i] = ra_d, dec_d
            ra, dec = math.radians(ra_d), math.radians(dec_d)
            u = np.array([math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec)])
            _, wr, wd = recrad(self.M @ u)
            ax = (-math.degrees(wr)) % 360.0
            if ax > 180.0:
                ax -= 360.0
            ax, ay = ax * 3600.0, math.degrees(wd) * 3600.0
            x, y = a2i @ (np.array([ax, ay]) + off)
            if -0.5 < x < nx - 0.5 and -0.5 < y < ny - 0.5:
                out['PIXEL-X'][i], out['PIXEL-Y'][i] = x, y
        return {n: v.reshape(shape) for n, v in out.items()}


def from_scenario(d: dict, g: PMGeometry) -> ExactBody:
    """`d`: a scenario / motion fixture (planetmapper_amd/data/*.json, tests/golden/motion_*.json)"""
    return ExactBody(Ephemeris.from_json(d['ephemeris']), RotationModel.from_json(d['pck']), d['target_id'], g,
                     observer_id=d.get('observer_id'))
