"""
A stand-in for the handful of spiceypy calls `planetmapper_amd/reference_binding.py` makes, built on this repo's own
kernel readers (`planetmapper_amd/ephem.py`: Chebyshev SPK types 2 / 3, IAU orientation from PCK constants). Test
infrastructure: it lets the reference-side binding RUN here, where neither spiceypy nor CSPICE exists, against the
same kernel data `GeometryBuilder` reads. Semantics follow the CSPICE routines of the same names (spkssb, spkgeo,
spkgps, spkpos with CN / LT+S, spksfs + spkuds, pxform, sxform with the derivative block tisbod builds, bodvar,
bods2c, pl2nvc, clight).
"""

from __future__ import annotations

import math

import numpy as np

from planetmapper_amd.ephem import CLIGHT, JULIAN_CENTURY_S, SPD, Ephemeris, RotationModel, rotate
from planetmapper_amd.geometry import stelab


def _drotate(angle: float, axis: int) -> np.ndarray:
    """d rotate(angle, axis) / d angle"""
    c, s = math.cos(angle), math.sin(angle)
    if axis == 1:
        return np.array([[0.0, 0.0, 0.0], [0.0, -s, c], [0.0, -c, -s]])
    if axis == 3:
        return np.array([[-s, c, 0.0], [-c, -s, 0.0], [0.0, 0.0, 0.0]])
    raise ValueError(axis)


class Plane:
    def __init__(self, normal, const):
        self.normal, self.const = np.asarray(normal, dtype=float), float(const)


class SpiceStandIn:
    def __init__(self, ephemeris: Ephemeris, rotations: dict[str, RotationModel], names: dict[str, int]):
        self.eph = ephemeris
        self.rotations = {k.upper(): v for k, v in rotations.items()}  # frame name ('IAU_JUPITER') -> model
        self.names = {k.upper(): v for k, v in names.items()}
        self.calls: list[str] = []

    # ---------------------------------------------------------------- constants, names
    def clight(self) -> float:
        return CLIGHT

    def bods2c(self, name) -> int:
        if isinstance(name, (int, np.integer)):
            return int(name)
        try:
            return int(name)
        except ValueError:
            return self.names[str(name).upper()]

    def bodvar(self, body: int, item: str, n: int):
        rot = next(r for r in self.rotations.values() if r.body_id == body)
        return np.array({'RADII': rot.radii, 'PM': rot.pm, 'POLE_RA': rot.pole_ra, 'POLE_DEC': rot.pole_dec}[item][:n], dtype=float)

    def pl2nvc(self, plane: Plane):
        return plane.normal, plane.const

    # ---------------------------------------------------------------- ephemeris
    def spksfs(self, body: int, et: float, idlen: int):
        self.calls.append('spksfs')
        return 0, self.eph._find(int(body), et), 'segment'

    def spkuds(self, descr):
        return descr.target, descr.center, descr.frame, descr.spk_type, descr.et_begin, descr.et_end, 0, 0

    def _rel_state(self, targ: int, et: float, obs: int) -> np.ndarray:
        pt, vt, _ = self.eph.ssb_state(targ, et) if targ != 0 else (np.zeros(3), np.zeros(3), None)
        po, vo, _ = self.eph.ssb_state(obs, et) if obs != 0 else (np.zeros(3), np.zeros(3), None)
        return np.concatenate([pt - po, vt - vo])

    def _hop_state(self, targ: int, et: float, obs: int):
        """targ relative to obs without going through the SSB when obs is the centre of targ's own segment (as CSPICE does:
        it adds up only the segments between the two bodies)"""
        seg = self.eph._find(int(targ), et)
        if seg.center == obs:
            p, v, _ = seg.state(et)
            return np.concatenate([p, v])
        return self._rel_state(targ, et, obs)

    def spkssb(self, body: int, et: float, ref: str) -> np.ndarray:
        assert ref == 'J2000'
        return self._rel_state(int(body), et, 0)

    def spkgeo(self, targ: int, et: float, ref: str, obs: int):
        assert ref == 'J2000'
        st = self._hop_state(int(targ), et, int(obs))
        return st, float(np.linalg.norm(st[:3])) / CLIGHT

    def spkgps(self, targ: int, et: float, ref: str, obs: int):
        st, lt = self.spkgeo(targ, et, ref, obs)
        return st[:3], lt

    def spkpos(self, targ, et: float, ref: str, abcorr: str, obs):
        """position of targ as seen from obs at et: 'CN' converged Newtonian light time, 'LT+S' one iteration + stellar
        aberration; ref: 'J2000' or a body-fixed frame evaluated at et (the observer's epoch: spkpos_c for a frame
        centred on the observer)"""
        t, o = self.bods2c(targ), self.bods2c(obs)
        po, vo, _ = self.eph.ssb_state(o, et)
        lt = 0.0
        for _ in range(3 if abcorr == 'CN' else 1):
            lt = float(np.linalg.norm(self.eph.ssb_state(t, et - lt)[0] - po)) / CLIGHT
        pos = self.eph.ssb_state(t, et - lt)[0] - po
        if abcorr == 'LT+S':
            pos = stelab(pos, vo)
        elif abcorr != 'CN':
            raise NotImplementedError(abcorr)
        if ref != 'J2000':
            pos = self.pxform('J2000', ref, et) @ pos
        return pos, lt

    # ---------------------------------------------------------------- orientation
    def pxform(self, frm: str, to: str, et: float) -> np.ndarray:
        assert frm == 'J2000'
        return self.rotations[to.upper()].matrix(et)

    def sxform(self, frm: str, to: str, et: float) -> np.ndarray:
        """6 x 6 state transformation [[R, 0], [dR/dt, R]]: the IAU model differentiated term by term (tisbod)"""
        assert frm == 'J2000'
        rot = self.rotations[to.upper()]
        ra, dec, w = rot.euler_rad(et)
        ra_dot, dec_dot = rot.pole_rates(et)
        w_dot = rot.spin_rate(et)
        a3, a1, b3 = w, math.pi / 2.0 - dec, math.pi / 2.0 + ra
        r3a, r1, r3b = rotate(a3, 3), rotate(a1, 1), rotate(b3, 3)
        drdt = w_dot * _drotate(a3, 3) @ r1 @ r3b - dec_dot * r3a @ _drotate(a1, 1) @ r3b + ra_dot * r3a @ r1 @ _drotate(b3, 3)
        rmat = r3a @ r1 @ r3b
        xf = np.zeros((6, 6))
        xf[:3, :3] = rmat
        xf[3:, 3:] = rmat
        xf[3:, :3] = drdt
        return xf


class DuckBody:
    """what `geometry_from_body` reads of a reference Body, filled from a GeometryBuilder block (the attributes Body.__init__
    has already asked of SPICE: base.py:828-839, body.py:521-588)"""

    aberration_correction = 'CN'
    observer_frame = 'J2000'
    illumination_source = 'SUN'

    def __init__(self, g, *, target: str, target_id: int, observer, frame: str):
        self.et, self.target_light_time = g.et, g.lt_c
        self.target, self.target_body_id, self.observer, self.target_frame = target, target_id, observer, frame
        self.radii = np.array(g.radii[:])
        self._target_obsvec = np.array(g.T0[:])
        self._subpoint_targvec, self._subpoint_rayvec = np.array(g.sub_sp[:]), np.array(g.sub_ray[:])
        self._subpoint_obsvec, self._subpoint_et, self.subpoint_distance = np.array(g.sub_obsvec[:]), g.sub_et, g.sub_dist
        self._ring_plane = Plane(g.ring_n[:], g.ring_k)
        self._m = np.array(g.M[:]).reshape(3, 3)
        self.target_diameter_arcsec, self.km_per_arcsec = g.diameter_arcsec, g.km_per_arcsec
        self._np_deg = float(np.rad2deg(g.np_angle_rad))
        self.positive_longitude_direction = 'W' if g.west_positive else 'E'

    def _get_obsvec2angular_matrix(self):
        return self._m

    def north_pole_angle(self):
        return self._np_deg
