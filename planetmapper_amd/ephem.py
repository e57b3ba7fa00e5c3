"""
SPICE-free host ephemeris + orientation provider.

The reference obtains every geometric constant from NAIF CSPICE through spiceypy
(`planetmapper/base.py:795-839`, `planetmapper/body.py:501-606`). CSPICE is a
third-party dependency that is not vendored by the reference, so this module restates
the small part of it that the *host side* of the hot path needs:

* text kernels (PCK, `pck00010.tpc` style): ``BODYnnn_RADII``, ``_POLE_RA``,
  ``_POLE_DEC``, ``_PM``, ``_NUT_PREC_*`` -> IAU rotation model J2000 -> body-fixed;
* binary DAF/SPK ephemerides, Chebyshev segment types 2 and 3 (planets, barycentres,
  Sun, Earth) -> position / velocity / acceleration relative to the SSB;
* SPK type 10 (two-line elements of an Earth satellite: the reference's canonical observer, HST,
  `tests/test_body.py:29-31`): CSPICE `spke10` restated - SGP4 for near-earth orbits from the two
  bracketing element sets, blended, TEME -> J2000 with the segment's own nutation angles.

Nothing here runs per pixel: it produces the numbers that `geometry.py` packs into
the geometry block consumed by the HIP kernels.

A `MiniEphemeris` is a JSON-serialisable subset of an SPK (only the Chebyshev records
that cover a time window) so test fixtures stay a few KB and travel to machines that
do not have the original kernels.
"""

from __future__ import annotations

import json
import math
import re
import struct
from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

CLIGHT = 299792.458  # km/s, spice.clight() (`planetmapper/base.py:514-522`)
SPD = 86400.0
JULIAN_CENTURY_S = SPD * 36525.0


# ----------------------------------------------------------------------------------
# Text kernels
# ----------------------------------------------------------------------------------
_NUM_RE = re.compile(r'^[+-]?(\d+\.?\d*|\.\d+)([EeDd][+-]?\d+)?$')


def _parse_scalar(tok: str):
    if tok.startswith("'"):
        return tok.strip("'")
    if tok.startswith('@'):
        return tok
    if _NUM_RE.match(tok):
        return float(tok.replace('D', 'E').replace('d', 'e'))
    return tok


def parse_text_kernel(text: str) -> dict[str, list]:
    """
    Parse the ``\\begindata`` sections of a NAIF text kernel into ``{NAME: [values]}``.
    Supports ``=`` and ``+=`` assignments, parenthesised vectors and Fortran ``D``
    exponents. Quoted strings are kept as strings.
    """
    pool: dict[str, list] = {}
    in_data = False
    buf: list[str] = []
    for line in text.splitlines():
        s = line.strip()
        if s.startswith('\\begindata'):
            in_data = True
            continue
        if s.startswith('\\begintext'):
            in_data = False
            continue
        if in_data:
            buf.append(line)
    data = '\n'.join(buf)
    # tokenise: quoted strings, parentheses, assignment operators, bare words
    tokens = re.findall(r"'(?:[^']|'')*'|\(|\)|\+=|=|[^\s,()=]+", data)
    i = 0
    while i < len(tokens):
        name = tokens[i]
        if i + 1 >= len(tokens) or tokens[i + 1] not in ('=', '+='):
            i += 1
            continue
        op = tokens[i + 1]
        i += 2
        values: list = []
        if i < len(tokens) and tokens[i] == '(':
            i += 1
            while i < len(tokens) and tokens[i] != ')':
                values.append(_parse_scalar(tokens[i]))
                i += 1
            i += 1
        elif i < len(tokens):
            values.append(_parse_scalar(tokens[i]))
            i += 1
        if op == '+=' and name in pool:
            pool[name].extend(values)
        else:
            pool[name] = values
    return pool


# ----------------------------------------------------------------------------------
# Rotation helpers (CSPICE frame-rotation convention, `rotate_c`)
# ----------------------------------------------------------------------------------
def rotate(angle: float, axis: int) -> np.ndarray:
    """
    CSPICE ``rotate_c``: matrix that rotates the *coordinate frame* by `angle` about
    `axis`; e.g. axis 3 -> [[c, s, 0], [-s, c, 0], [0, 0, 1]].
    """
    c = math.cos(angle)
    s = math.sin(angle)
    if axis == 1:
        return np.array([[1.0, 0.0, 0.0], [0.0, c, s], [0.0, -s, c]])
    if axis == 2:
        return np.array([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])
    if axis == 3:
        return np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    raise ValueError('axis must be 1, 2 or 3')


@dataclass
class RotationModel:
    """
    IAU orientation model of one body from PCK constants (``tisbod`` semantics):

        T = et / (86400 * 36525),  d = et / 86400
        RA  = RA0  + RA1 T  + RA2 T^2  + sum a_i sin(theta_i)
        DEC = DEC0 + DEC1 T + DEC2 T^2 + sum d_i cos(theta_i)
        W   = W0   + W1 d   + W2 d^2   + sum w_i sin(theta_i)
        theta_i = A_i + B_i T      (degrees, ``BODY<bary>_NUT_PREC_ANGLES``)
        R(t) = rot3(W) rot1(pi/2 - DEC) rot3(pi/2 + RA)     (J2000 -> body-fixed)
    """

    body_id: int
    pole_ra: Sequence[float]
    pole_dec: Sequence[float]
    pm: Sequence[float]
    nut_prec_ra: Sequence[float] = ()
    nut_prec_dec: Sequence[float] = ()
    nut_prec_pm: Sequence[float] = ()
    nut_prec_angles: Sequence[float] = ()  # flat (A0, B0, A1, B1, ...)
    radii: Sequence[float] = (1.0, 1.0, 1.0)

    @classmethod
    def from_pool(cls, pool: dict[str, list], body_id: int) -> 'RotationModel':
        def get(name: str, default=()):
            return tuple(pool.get(f'BODY{body_id}_{name}', default))

        if body_id >= 100:
            bary = body_id // 100
        else:
            bary = body_id
        angles = tuple(pool.get(f'BODY{bary}_NUT_PREC_ANGLES', ()))
        return cls(
            body_id=body_id,
            pole_ra=get('POLE_RA'),
            pole_dec=get('POLE_DEC'),
            pm=get('PM'),
            nut_prec_ra=get('NUT_PREC_RA'),
            nut_prec_dec=get('NUT_PREC_DEC'),
            nut_prec_pm=get('NUT_PREC_PM'),
            nut_prec_angles=angles,
            radii=get('RADII', (1.0, 1.0, 1.0)),
        )

    def to_json(self) -> dict:
        return {
            'body_id': self.body_id,
            'pole_ra': list(self.pole_ra),
            'pole_dec': list(self.pole_dec),
            'pm': list(self.pm),
            'nut_prec_ra': list(self.nut_prec_ra),
            'nut_prec_dec': list(self.nut_prec_dec),
            'nut_prec_pm': list(self.nut_prec_pm),
            'nut_prec_angles': list(self.nut_prec_angles),
            'radii': list(self.radii),
        }

    @classmethod
    def from_json(cls, d: dict) -> 'RotationModel':
        return cls(**d)

    @property
    def prograde(self) -> bool:
        # `planetmapper/body.py:526-528`
        return self.pm[1] >= 0

    def euler_deg(self, et: float) -> tuple[float, float, float]:
        """(RA, DEC, W) of the pole / prime meridian in degrees at TDB `et`."""
        t_cent = et / JULIAN_CENTURY_S
        d_days = et / SPD

        def poly(c, x):
            out = 0.0
            for k, ck in enumerate(c):
                out += ck * x**k
            return out

        ra = poly(self.pole_ra, t_cent)
        dec = poly(self.pole_dec, t_cent)
        w = poly(self.pm, d_days)
        na = len(self.nut_prec_angles) // 2
        for i in range(na):
            theta = math.radians(
                self.nut_prec_angles[2 * i] + self.nut_prec_angles[2 * i + 1] * t_cent
            )
            if i < len(self.nut_prec_ra):
                ra += self.nut_prec_ra[i] * math.sin(theta)
            if i < len(self.nut_prec_dec):
                dec += self.nut_prec_dec[i] * math.cos(theta)
            if i < len(self.nut_prec_pm):
                w += self.nut_prec_pm[i] * math.sin(theta)
        return ra, dec, w

    def euler_rad(self, et: float) -> tuple[float, float, float]:
        ra, dec, w = self.euler_deg(et)
        twopi = 2.0 * math.pi
        return (
            math.radians(ra) % twopi,
            math.radians(dec) % twopi,
            math.radians(w) % twopi,
        )

    def matrix(self, et: float) -> np.ndarray:
        """J2000 -> body-fixed rotation at `et` (``pxform('J2000', 'IAU_x', et)``)."""
        ra, dec, w = self.euler_rad(et)
        return (
            rotate(w, 3)
            @ rotate(math.pi / 2.0 - dec, 1)
            @ rotate(math.pi / 2.0 + ra, 3)
        )

    def spin_rate(self, et: float) -> float:
        """
        dW/dt in rad/s, the derivative of the model taken term by term (nutation terms included). (A
        central difference of W itself, 1.6e6 deg for Jupiter in 2005, carries the rounding of W divided
        by the step: 1.7e-13 rad/s - as much as the whole rotation budget over a light-time span.)
        """
        t_cent = et / JULIAN_CENTURY_S
        d_days = et / SPD
        rate = 0.0  # deg / s
        for k, ck in enumerate(self.pm):
            if k >= 1:
                rate += k * ck * d_days ** (k - 1) / SPD
        for i in range(min(len(self.nut_prec_pm), len(self.nut_prec_angles) // 2)):
            theta = math.radians(self.nut_prec_angles[2 * i] + self.nut_prec_angles[2 * i + 1] * t_cent)
            theta_dot = math.radians(self.nut_prec_angles[2 * i + 1]) / JULIAN_CENTURY_S  # rad / s
            rate += self.nut_prec_pm[i] * math.cos(theta) * theta_dot
        return math.radians(rate)

    def pole_rates(self, et: float) -> tuple[float, float]:
        """(d RA / dt, d DEC / dt) of the pole in rad/s, term by term."""
        t_cent = et / JULIAN_CENTURY_S
        ra = dec = 0.0  # deg / century
        for k, ck in enumerate(self.pole_ra):
            if k >= 1:
                ra += k * ck * t_cent ** (k - 1)
        for k, ck in enumerate(self.pole_dec):
            if k >= 1:
                dec += k * ck * t_cent ** (k - 1)
        for i in range(len(self.nut_prec_angles) // 2):
            theta = math.radians(self.nut_prec_angles[2 * i] + self.nut_prec_angles[2 * i + 1] * t_cent)
            rate = math.radians(self.nut_prec_angles[2 * i + 1])  # rad / century
            if i < len(self.nut_prec_ra):
                ra += self.nut_prec_ra[i] * math.cos(theta) * rate
            if i < len(self.nut_prec_dec):
                dec -= self.nut_prec_dec[i] * math.sin(theta) * rate
        return math.radians(ra) / JULIAN_CENTURY_S, math.radians(dec) / JULIAN_CENTURY_S

    def angular_velocity(self, et: float) -> np.ndarray:
        """
        Angular velocity of the body-fixed frame with respect to J2000, J2000 components, rad/s: with
        R = rot3(W) rot1(pi/2 - DEC) rot3(pi/2 + RA), d RA/dt about J2000 z, -d DEC/dt about the node
        (-sin RA, cos RA, 0) and dW/dt about the pole. (What the derivative block of sxform_c encodes.)
        """
        ra_dot, dec_dot = self.pole_rates(et)
        ra, _dec, _w = self.euler_rad(et)
        pole = self.matrix(et)[2]
        return (ra_dot * np.array([0.0, 0.0, 1.0]) - dec_dot * np.array([-math.sin(ra), math.cos(ra), 0.0])
                + self.spin_rate(et) * pole)  # fmt: skip

    def pole_drift(self, et: float) -> np.ndarray:
        """`angular_velocity` less its component along the pole (which is `body_z_rate`): the motion of the pole itself"""
        w = self.angular_velocity(et)
        pole = self.matrix(et)[2]
        return w - float(w @ pole) * pole

    def body_z_rate(self, et: float) -> float:
        """
        Angular velocity of the body-fixed frame about its own +z axis, rad/s: dW/dt plus the component of
        the pole's precession along the pole, d RA / dt * sin(DEC) (W is counted from the node of the
        equator, which moves with the pole: Saturn 2e-13 rad/s). R(t0 + d) = Rz(body_z_rate * d) R(t0) up
        to the TRANSVERSE motion of the pole (Jupiter 1e-14, Saturn 2e-14, Mars 5e-13 rad/s).
        """
        ra_dot, _ = self.pole_rates(et)
        dec = math.radians(self.euler_deg(et)[1])
        return self.spin_rate(et) + ra_dot * math.sin(dec)


# ----------------------------------------------------------------------------------
# Chebyshev ephemeris segments
# ----------------------------------------------------------------------------------
def _cheby_eval(coeffs: np.ndarray, s: float) -> tuple[float, float, float]:
    """Value, d/ds and d2/ds2 of sum c_k T_k(s) by the Chebyshev recurrences."""
    n = len(coeffs)
    t = np.empty(n)
    dt = np.empty(n)
    ddt = np.empty(n)
    t[0], dt[0], ddt[0] = 1.0, 0.0, 0.0
    if n > 1:
        t[1], dt[1], ddt[1] = s, 1.0, 0.0
    for k in range(2, n):
        t[k] = 2.0 * s * t[k - 1] - t[k - 2]
        dt[k] = 2.0 * t[k - 1] + 2.0 * s * dt[k - 1] - dt[k - 2]
        ddt[k] = 4.0 * dt[k - 1] + 2.0 * s * ddt[k - 1] - ddt[k - 2]
    return float(coeffs @ t), float(coeffs @ dt), float(coeffs @ ddt)


@dataclass
class ChebySegment:
    """One SPK segment of type 2 (position coefficients) or 3 (position+velocity)."""

    target: int
    center: int
    frame: int
    spk_type: int
    et_begin: float
    et_end: float
    init: float
    intlen: float
    first_record: int  # index of records[0] within the original segment
    records: np.ndarray  # (n_records, rsize)

    def _record(self, et: float) -> np.ndarray:
        idx = int(math.floor((et - self.init) / self.intlen)) - self.first_record
        # the final epoch of a segment belongs to the last record
        idx = min(idx, len(self.records) - 1)
        if idx < 0:
            raise ValueError(
                f'et={et} outside stored records of segment {self.target}/{self.center}'
            )
        return self.records[idx]

    def covers(self, et: float) -> bool:
        if not self.et_begin <= et <= self.et_end:
            return False
        idx = int(math.floor((et - self.init) / self.intlen)) - self.first_record
        idx = min(idx, len(self.records) - 1)
        return 0 <= idx

    def state(self, et: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(position km, velocity km/s, acceleration km/s^2) of target wrt center."""
        rec = self._record(et)
        mid, radius = rec[0], rec[1]
        ncomp = 3 if self.spk_type == 2 else 6
        n = (len(rec) - 2) // ncomp
        s = (et - mid) / radius
        pos = np.empty(3)
        vel = np.empty(3)
        acc = np.empty(3)
        for i in range(3):
            c = rec[2 + i * n : 2 + (i + 1) * n]
            p, dp, ddp = _cheby_eval(c, s)
            pos[i] = p
            vel[i] = dp / radius
            acc[i] = ddp / (radius * radius)
        if self.spk_type == 3:
            for i in range(3):
                c = rec[2 + (3 + i) * n : 2 + (4 + i) * n]
                v, dv, _ = _cheby_eval(c, s)
                vel[i] = v
                acc[i] = dv / radius
        return pos, vel, acc

    def motion(self, et: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """
        (position, d/dt position, d2/dt2 position): the derivatives of the POSITION series itself, also
        for type 3 - whose separate velocity polynomial (`state`) is a fit of its own, a few 1e-6 km/s
        away from this. What evaluating the position at neighbouring epochs amounts to.
        """
        rec = self._record(et)
        mid, radius = rec[0], rec[1]
        ncomp = 3 if self.spk_type == 2 else 6
        n = (len(rec) - 2) // ncomp
        s = (et - mid) / radius
        pos, vel, acc = np.empty(3), np.empty(3), np.empty(3)
        for i in range(3):
            p, dp, ddp = _cheby_eval(rec[2 + i * n : 2 + (i + 1) * n], s)
            pos[i], vel[i], acc[i] = p, dp / radius, ddp / (radius * radius)
        return pos, vel, acc

    def trimmed(self, et_lo: float, et_hi: float) -> 'ChebySegment':
        """Copy holding only the records needed for [et_lo, et_hi]."""
        i0 = int(math.floor((et_lo - self.init) / self.intlen)) - self.first_record
        i1 = int(math.floor((et_hi - self.init) / self.intlen)) - self.first_record
        i0 = max(i0, 0)
        i1 = min(i1, len(self.records) - 1)
        return ChebySegment(
            target=self.target,
            center=self.center,
            frame=self.frame,
            spk_type=self.spk_type,
            et_begin=max(self.et_begin, self.init + (i0 + self.first_record) * self.intlen),
            et_end=min(
                self.et_end, self.init + (i1 + 1 + self.first_record) * self.intlen
            ),
            init=self.init,
            intlen=self.intlen,
            first_record=self.first_record + i0,
            records=np.array(self.records[i0 : i1 + 1], dtype=np.float64),
        )

    def to_json(self) -> dict:
        return {
            'target': self.target,
            'center': self.center,
            'frame': self.frame,
            'spk_type': self.spk_type,
            'et_begin': self.et_begin,
            'et_end': self.et_end,
            'init': self.init,
            'intlen': self.intlen,
            'first_record': self.first_record,
            # hex keeps every bit of the float64 coefficients
            'records': [[float(v).hex() for v in rec] for rec in self.records],
        }

    @classmethod
    def from_json(cls, d: dict) -> 'ChebySegment':
        recs = np.array(
            [[float.fromhex(v) for v in rec] for rec in d['records']], dtype=np.float64
        )
        kw = {k: v for k, v in d.items() if k != 'records'}
        return cls(records=recs, **kw)


# ----------------------------------------------------------------------------------
# SPK type 10: two-line elements (CSPICE spke10 / spkr10, N0067)
# ----------------------------------------------------------------------------------
_ARCSEC = math.pi / 648000.0
_TWOPI = 2.0 * math.pi


class _Sgp4:
    """
    SGP4 for NEAR-EARTH orbits (period < 225 min) as `spke10` evaluates it since N0067 (Vallado, Crawford, Hujsak & Kelso
    2006, "Revisiting Spacetrack Report #3": CSPICE xxsgp4i / xxsgp4e), with the geophysical constants of the SEGMENT
    (J2, J3, J4, KE, QO, SO, ER, AE) in place of WGS-72's. Elements as the type 10 packet holds them: NDT20, NDD60, BSTAR,
    INCL, NODE0, ECC, OMEGA, M0 (rad), N0 (rad / min), EPOCH (TDB seconds past J2000). Deep-space element sets are refused.
    """

    def __init__(self, geophs: Sequence[float], el: Sequence[float]) -> None:
        j2, j3, j4, xke, qo, so, er, _ae = (float(v) for v in geophs[:8])
        _ndt20, _ndd60, bstar, inclo, nodeo, ecco, argpo, mo, no, epoch = (float(v) for v in el[:10])
        self.j2, self.xke, self.er, self.epoch, self.bstar = j2, xke, er, epoch, bstar
        self.inclo, self.nodeo, self.ecco, self.argpo, self.mo = inclo, nodeo, ecco, argpo, mo
        j3oj2 = j3 / j2
        ss = so / er + 1.0
        qzms2t = ((qo - so) / er) ** 4
        x2o3 = 2.0 / 3.0
        # initl: the mean motion and semi-major axis without the Kozai terms
        eccsq = ecco * ecco
        omeosq = 1.0 - eccsq
        rteosq = math.sqrt(omeosq)
        cosio = math.cos(inclo)
        cosio2 = cosio * cosio
        ak = (xke / no) ** x2o3
        d1 = 0.75 * j2 * (3.0 * cosio2 - 1.0) / (rteosq * omeosq)
        del_ = d1 / (ak * ak)
        adel = ak * (1.0 - del_ * del_ - del_ * (1.0 / 3.0 + 134.0 * del_ * del_ / 81.0))
        del_ = d1 / (adel * adel)
        no = no / (1.0 + del_)
        ao = (xke / no) ** x2o3
        sinio = math.sin(inclo)
        po = ao * omeosq
        con42 = 1.0 - 5.0 * cosio2
        con41 = -con42 - cosio2 - cosio2
        posq = po * po
        rp = ao * (1.0 - ecco)
        if _TWOPI / no >= 225.0:
            raise ValueError('SPK type 10: deep-space element set (period >= 225 min) - only the near-earth model is restated')
        self.no = no
        self.isimp = 1 if rp < (220.0 / er + 1.0) else 0
        sfour, qzms24 = ss, qzms2t
        perige = (rp - 1.0) * er
        if perige < 156.0:
            sfour = perige - 78.0
            if perige < 98.0:
                sfour = 20.0
            qzms24 = ((120.0 - sfour) / er) ** 4
            sfour = sfour / er + 1.0
        pinvsq = 1.0 / posq
        tsi = 1.0 / (ao - sfour)
        self.eta = eta = ao * ecco * tsi
        etasq = eta * eta
        eeta = ecco * eta
        psisq = abs(1.0 - etasq)
        coef = qzms24 * tsi**4
        coef1 = coef / psisq**3.5
        cc2 = coef1 * no * (ao * (1.0 + 1.5 * etasq + eeta * (4.0 + etasq))
                            + 0.375 * j2 * tsi / psisq * con41 * (8.0 + 3.0 * etasq * (8.0 + etasq)))  # fmt: skip
        self.cc1 = cc1 = bstar * cc2
        cc3 = -2.0 * coef * tsi * j3oj2 * no * sinio / ecco if ecco > 1.0e-4 else 0.0
        self.x1mth2 = x1mth2 = 1.0 - cosio2
        self.cc4 = 2.0 * no * coef1 * ao * omeosq * (
            eta * (2.0 + 0.5 * etasq) + ecco * (0.5 + 2.0 * etasq)
            - j2 * tsi / (ao * psisq) * (-3.0 * con41 * (1.0 - 2.0 * eeta + etasq * (1.5 - 0.5 * eeta))
                                         + 0.75 * x1mth2 * (2.0 * etasq - eeta * (1.0 + etasq)) * math.cos(2.0 * argpo)))  # fmt: skip
        self.cc5 = 2.0 * coef1 * ao * omeosq * (1.0 + 2.75 * (etasq + eeta) + eeta * etasq)
        cosio4 = cosio2 * cosio2
        temp1 = 1.5 * j2 * pinvsq * no
        temp2 = 0.5 * temp1 * j2 * pinvsq
        temp3 = -0.46875 * j4 * pinvsq * pinvsq * no
        self.mdot = no + 0.5 * temp1 * rteosq * con41 + 0.0625 * temp2 * rteosq * (13.0 - 78.0 * cosio2 + 137.0 * cosio4)
        self.argpdot = (-0.5 * temp1 * con42 + 0.0625 * temp2 * (7.0 - 114.0 * cosio2 + 395.0 * cosio4)
                        + temp3 * (3.0 - 36.0 * cosio2 + 49.0 * cosio4))  # fmt: skip
        xhdot1 = -temp1 * cosio
        self.nodedot = xhdot1 + (0.5 * temp2 * (4.0 - 19.0 * cosio2) + 2.0 * temp3 * (3.0 - 7.0 * cosio2)) * cosio
        self.omgcof = bstar * cc3 * math.cos(argpo)
        self.xmcof = -x2o3 * coef * bstar / eeta if ecco > 1.0e-4 else 0.0
        self.nodecf = 3.5 * omeosq * xhdot1 * cc1
        self.t2cof = 1.5 * cc1
        self.xlcof = -0.25 * j3oj2 * sinio * (3.0 + 5.0 * cosio) / ((1.0 + cosio) if abs(cosio + 1.0) > 1.5e-12 else 1.5e-12)
        self.aycof = -0.5 * j3oj2 * sinio
        self.delmo = (1.0 + eta * math.cos(mo)) ** 3
        self.sinmao = math.sin(mo)
        self.x7thm1 = 7.0 * cosio2 - 1.0
        self.con41 = con41
        if self.isimp != 1:
            cc1sq = cc1 * cc1
            self.d2 = d2 = 4.0 * ao * tsi * cc1sq
            temp = d2 * tsi * cc1 / 3.0
            self.d3 = d3 = (17.0 * ao + sfour) * temp
            self.d4 = d4 = 0.5 * temp * ao * tsi * (221.0 * ao + 31.0 * sfour) * cc1
            self.t3cof = d2 + 2.0 * cc1sq
            self.t4cof = 0.25 * (3.0 * d3 + cc1 * (12.0 * d2 + 10.0 * cc1sq))
            self.t5cof = 0.2 * (3.0 * d4 + 12.0 * cc1 * d3 + 6.0 * d2 * d2 + 15.0 * cc1sq * (2.0 * d2 + cc1sq))

    def state(self, et: float) -> tuple[np.ndarray, np.ndarray]:
        """position (km) and velocity (km/s) at `et` in TEME, the frame of the elements"""
        t = (et - self.epoch) / 60.0
        xke, j2 = self.xke, self.j2
        # secular gravity and atmospheric drag
        xmdf = self.mo + self.mdot * t
        argpdf = self.argpo + self.argpdot * t
        nodedf = self.nodeo + self.nodedot * t
        argpm, mm = argpdf, xmdf
        t2 = t * t
        nodem = nodedf + self.nodecf * t2
        tempa = 1.0 - self.cc1 * t
        tempe = self.bstar * self.cc4 * t
        templ = self.t2cof * t2
        if self.isimp != 1:
            delomg = self.omgcof * t
            delm = self.xmcof * ((1.0 + self.eta * math.cos(xmdf)) ** 3 - self.delmo)
            temp = delomg + delm
            mm = xmdf + temp
            argpm = argpdf - temp
            t3 = t2 * t
            t4 = t3 * t
            tempa = tempa - self.d2 * t2 - self.d3 * t3 - self.d4 * t4
            tempe = tempe + self.bstar * self.cc5 * (math.sin(mm) - self.sinmao)
            templ = templ + self.t3cof * t3 + t4 * (self.t4cof + t * self.t5cof)
        nm, em, inclm = self.no, self.ecco, self.inclo
        am = (xke / nm) ** (2.0 / 3.0) * tempa * tempa
        nm = xke / am**1.5
        em = max(em - tempe, 1.0e-6)
        mm = mm + self.no * templ
        xlm = mm + argpm + nodem
        nodem = math.fmod(nodem, _TWOPI)
        argpm = math.fmod(argpm, _TWOPI)
        xlm = math.fmod(xlm, _TWOPI)
        mm = math.fmod(xlm - argpm - nodem, _TWOPI)
        sinip, cosip = math.sin(inclm), math.cos(inclm)
        # long-period periodics, Kepler's equation
        axnl = em * math.cos(argpm)
        temp = 1.0 / (am * (1.0 - em * em))
        aynl = em * math.sin(argpm) + temp * self.aycof
        xl = mm + argpm + nodem + temp * self.xlcof * axnl
        u = math.fmod(xl - nodem, _TWOPI)
        eo1, tem5, ktr = u, 9999.9, 1
        sineo1 = coseo1 = 0.0
        while abs(tem5) >= 1.0e-12 and ktr <= 10:
            sineo1, coseo1 = math.sin(eo1), math.cos(eo1)
            tem5 = 1.0 - coseo1 * axnl - sineo1 * aynl
            tem5 = (u - aynl * coseo1 + axnl * sineo1 - eo1) / tem5
            if abs(tem5) >= 0.95:
                tem5 = math.copysign(0.95, tem5)
            eo1 += tem5
            ktr += 1
        # short-period periodics
        ecose = axnl * coseo1 + aynl * sineo1
        esine = axnl * sineo1 - aynl * coseo1
        el2 = axnl * axnl + aynl * aynl
        pl = am * (1.0 - el2)
        if pl < 0.0:
            raise ValueError('SPK type 10: the element set has decayed (semi-latus rectum < 0)')
        rl = am * (1.0 - ecose)
        rdotl = math.sqrt(am) * esine / rl
        rvdotl = math.sqrt(pl) / rl
        betal = math.sqrt(1.0 - el2)
        temp = esine / (1.0 + betal)
        sinu = am / rl * (sineo1 - aynl - axnl * temp)
        cosu = am / rl * (coseo1 - axnl + aynl * temp)
        su = math.atan2(sinu, cosu)
        sin2u = (cosu + cosu) * sinu
        cos2u = 1.0 - 2.0 * sinu * sinu
        temp = 1.0 / pl
        temp1 = 0.5 * j2 * temp
        temp2 = temp1 * temp
        mrt = rl * (1.0 - 1.5 * temp2 * betal * self.con41) + 0.5 * temp1 * self.x1mth2 * cos2u
        su = su - 0.25 * temp2 * self.x7thm1 * sin2u
        xnode = nodem + 1.5 * temp2 * cosip * sin2u
        xinc = inclm + 1.5 * temp2 * cosip * sinip * cos2u
        mvt = rdotl - nm * temp1 * self.x1mth2 * sin2u / xke
        rvdot = rvdotl + nm * temp1 * (self.x1mth2 * cos2u + 1.5 * self.con41) / xke
        sinsu, cossu = math.sin(su), math.cos(su)
        snod, cnod = math.sin(xnode), math.cos(xnode)
        sini, cosi = math.sin(xinc), math.cos(xinc)
        xmx, xmy = -snod * cosi, cnod * cosi
        uvec = np.array([xmx * sinsu + cnod * cossu, xmy * sinsu + snod * cossu, sini * sinsu])
        vvec = np.array([xmx * cossu - cnod * sinsu, xmy * cossu - snod * sinsu, sini * cossu])
        return uvec * (mrt * self.er), (mvt * uvec + rvdot * vvec) * (self.er * xke / 60.0)


def _j2000_to_teme(et: float, nut_obliquity: float, nut_longitude: float) -> np.ndarray:
    """
    J2000 -> TEME (true equator, mean equinox of date) at TDB `et`: IAU 1976 precession (zzeprc76), the IAU 1980 mean
    obliquity (zzmobliq), the nutation angles GIVEN (a type 10 segment stores them with each element set), and the turn
    about the true pole by the equation of the equinoxes, d psi cos(true obliquity), that takes the true equinox back to
    the mean one. The choice among the published variants (which obliquity in the equation of the equinoxes, none of
    its post-1997 lunar terms) is the one that puts HST where the reference's golden FITS header has it:
    tests/test_tle_observer.py.
    """
    t = et / JULIAN_CENTURY_S
    zeta = (2306.2181 * t + 0.30188 * t * t + 0.017998 * t**3) * _ARCSEC
    z = (2306.2181 * t + 1.09468 * t * t + 0.018203 * t**3) * _ARCSEC
    theta = (2004.3109 * t - 0.42665 * t * t - 0.041833 * t**3) * _ARCSEC
    mob = (84381.448 - 46.8150 * t - 0.00059 * t * t + 0.001813 * t**3) * _ARCSEC
    eps = mob + nut_obliquity
    prec = rotate(-z, 3) @ rotate(theta, 2) @ rotate(-zeta, 3)
    nut = rotate(-eps, 1) @ rotate(-nut_longitude, 3) @ rotate(mob, 1)
    return rotate(nut_longitude * math.cos(eps), 3) @ nut @ prec


@dataclass
class TleSegment:
    """
    One SPK segment of type 10: the two-line element sets of an Earth satellite (space-track format, converted by NAIF's
    mkspk), each with the nutation angles and rates of its epoch; `spke10` restated. State at `et`: SGP4 from the
    two element sets whose epochs bracket `et`, blended with the weight 1/2 + 1/2 cos(pi (et - t1) / (t2 - t1)) (the
    velocity takes the weight's derivative along), then TEME -> J2000 with the nutation angles interpolated between the two
    sets (cubic Hermite on the stored angles and rates). Pinned by the reference's golden header - HST on 2005-01-01:
    position to 3e-7 km of what TARGET RA / DEC / DISTANCE imply, radial-velocity plane to 4e-11 km/s.
    """

    target: int
    center: int
    frame: int
    et_begin: float
    et_end: float
    geophs: np.ndarray  # J2, J3, J4, KE, QO, SO, ER, AE
    epochs: np.ndarray  # (n,)
    packets: np.ndarray  # (n, 14): NDT20 NDD60 BSTAR INCL NODE0 ECC OMEGA M0 N0 EPOCH, nutation in obliquity / longitude, their rates
    spk_type: int = 10

    def covers(self, et: float) -> bool:
        return self.et_begin <= et <= self.et_end and len(self.epochs) > 0

    def _bracket(self, et: float) -> tuple[int, int]:
        n = len(self.epochs)
        i = int(np.searchsorted(self.epochs, et, side='right'))  # epochs[i - 1] <= et < epochs[i]
        if i <= 0:
            return 0, 0  # (before the first set / after the last: that set alone, spkr10)
        if i >= n:
            return n - 1, n - 1
        return i - 1, i

    def _j2000_to_teme_at(self, et: float, i1: int, i2: int) -> np.ndarray:
        n1, n2 = self.packets[i1][10:14], self.packets[i2][10:14]
        if i1 == i2:
            ang = n1[:2] + n1[2:] * (et - self.epochs[i1])
        else:
            h = self.epochs[i2] - self.epochs[i1]
            s = (et - self.epochs[i1]) / h
            ang = ((2 * s**3 - 3 * s**2 + 1) * n1[:2] + (s**3 - 2 * s**2 + s) * h * n1[2:]
                   + (-2 * s**3 + 3 * s**2) * n2[:2] + (s**3 - s**2) * h * n2[2:])  # fmt: skip
        return _j2000_to_teme(et, float(ang[0]), float(ang[1]))

    def state(self, et: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(position km, velocity km/s, acceleration km/s^2 = two-body estimate) of the satellite wrt the Earth, J2000"""
        i1, i2 = self._bracket(et)
        r1, v1 = _Sgp4(self.geophs, self.packets[i1]).state(et)
        if i1 == i2:
            r, v = r1, v1
        else:
            r2, v2 = _Sgp4(self.geophs, self.packets[i2]).state(et)
            t1, t2 = self.epochs[i1], self.epochs[i2]
            arg = (et - t1) * math.pi / (t2 - t1)
            w = 0.5 + 0.5 * math.cos(arg)
            dw = -0.5 * math.sin(arg) * math.pi / (t2 - t1)
            r = w * r1 + (1.0 - w) * r2
            v = w * v1 + (1.0 - w) * v2 + dw * (r1 - r2)
        m = self._j2000_to_teme_at(et, i1, i2)
        # the frame turns (precession 8e-12 rad/s, nutation 3e-12): 5e-8 km/s on a 7 000 km orbit
        h = 8.0
        dm = (self._j2000_to_teme_at(et + h, i1, i2) - self._j2000_to_teme_at(et - h, i1, i2)) / (2.0 * h)
        pos = m.T @ r
        vel = m.T @ v + dm.T @ r
        mu = (self.geophs[3] / 60.0) ** 2 * self.geophs[6] ** 3  # KE^2 ER^3 (km^3 / s^2)
        acc = -mu * pos / float(np.linalg.norm(pos)) ** 3
        return pos, vel, acc

    def motion(self, et: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        return self.state(et)

    def trimmed(self, et_lo: float, et_hi: float) -> 'TleSegment':
        """Copy holding only the element sets needed for [et_lo, et_hi]."""
        lo = max(self._bracket(et_lo)[0] - 1, 0)
        hi = min(self._bracket(et_hi)[1] + 1, len(self.epochs) - 1)
        return TleSegment(self.target, self.center, self.frame, max(self.et_begin, float(self.epochs[lo])), min(self.et_end, float(self.epochs[hi])),
                          np.array(self.geophs), np.array(self.epochs[lo : hi + 1]), np.array(self.packets[lo : hi + 1]))  # fmt: skip

    def to_json(self) -> dict:
        return {'target': self.target, 'center': self.center, 'frame': self.frame, 'spk_type': 10, 'et_begin': self.et_begin,
                'et_end': self.et_end, 'geophs': [float(v).hex() for v in self.geophs], 'epochs': [float(v).hex() for v in self.epochs],
                'packets': [[float(v).hex() for v in p] for p in self.packets]}  # fmt: skip

    @classmethod
    def from_json(cls, d: dict) -> 'TleSegment':
        unhex = lambda seq: np.array([float.fromhex(v) for v in seq], dtype=np.float64)  # noqa: E731
        return cls(d['target'], d['center'], d['frame'], d['et_begin'], d['et_end'], unhex(d['geophs']), unhex(d['epochs']),
                   np.array([unhex(p) for p in d['packets']], dtype=np.float64).reshape(-1, 14))  # fmt: skip


def _read_tle_segment(words: np.ndarray, target: int, center: int, frame: int, et_b: float, et_e: float, a0: int, a1: int) -> TleSegment:
    """
    A type 10 segment is a DAF "generic segment" (SPK Required Reading; sgfcon / sgfpkt): 8 constants, N packets of PKTSZ = 14
    doubles each behind PKTOFF offset words, their N reference epochs, an epoch directory, and 17 words of meta data
    (CONBAS NCON RDRBAS NRDR RDRTYP REFBAS NREF PDRBAS NPDR PDRTYP PKTBAS NPKT RSVBAS NRSV PKTSZ PKTOFF NMETA).
    """
    seg = words[a0 - 1 : a1]
    meta = [int(v) for v in seg[-17:]]
    conbas, ncon, refbas, nref, pktbas, npkt, pktsz, pktoff, nmeta = meta[0], meta[1], meta[5], meta[6], meta[10], meta[11], meta[14], meta[15], meta[16]
    if nmeta != 17 or ncon != 8 or pktsz != 14 or nref != npkt:
        raise ValueError('unexpected layout of an SPK type 10 segment')
    stride = pktsz + pktoff
    packets = np.array(seg[pktbas : pktbas + npkt * stride], dtype=np.float64).reshape(npkt, stride)[:, pktoff:]
    return TleSegment(target, center, frame, float(et_b), float(et_e), np.array(seg[conbas : conbas + ncon], dtype=np.float64),
                      np.array(seg[refbas : refbas + nref], dtype=np.float64), packets)  # fmt: skip


def read_spk_segments(path: str) -> list:
    """
    Read every type 2 / type 3 (`ChebySegment`) and type 10 (`TleSegment`) segment of a binary DAF/SPK
    file (other types are skipped). Layout: NAIF "DAF Required Reading" / "SPK Required Reading".
    """
    with open(path, 'rb') as f:
        raw = f.read()
    if raw[:7] != b'DAF/SPK':
        raise ValueError(f'{path!r} is not a DAF/SPK file')
    locfmt = raw[88:96].decode('ascii', 'replace')
    if locfmt.startswith('BIG'):
        e = '>'
    elif locfmt.startswith('LTL'):
        e = '<'
    else:
        raise ValueError(f'unknown DAF binary format {locfmt!r}')
    nd, ni = struct.unpack(e + 'ii', raw[8:16])
    fward = struct.unpack(e + 'i', raw[76:80])[0]
    if (nd, ni) != (2, 6):
        raise ValueError('not an SPK (ND, NI) != (2, 6)')
    words = np.frombuffer(raw, dtype=e + 'f8')  # 1-based DAF addresses -> words[a-1]
    segs: list[ChebySegment] = []
    rec = fward
    while rec > 0:
        base = (rec - 1) * 128  # double-word index of the summary record
        nxt, _prev, nsum = words[base : base + 3]
        for k in range(int(nsum)):
            off = base + 3 + k * 5
            et_b, et_e = words[off], words[off + 1]
            ints = np.frombuffer(
                raw[(off + 2) * 8 : (off + 5) * 8], dtype=e + 'i4'
            )
            target, center, frame, spk_type, a0, a1 = (int(v) for v in ints)
            if spk_type == 10:
                segs.append(_read_tle_segment(words, target, center, frame, et_b, et_e, a0, a1))
                continue
            if spk_type not in (2, 3):
                continue
            init, intlen, rsize, n = words[a1 - 4 : a1]
            rsize = int(rsize)
            n = int(n)
            records = np.array(
                words[a0 - 1 : a0 - 1 + rsize * n], dtype=np.float64
            ).reshape(n, rsize)
            segs.append(
                ChebySegment(
                    target=target,
                    center=center,
                    frame=frame,
                    spk_type=spk_type,
                    et_begin=float(et_b),
                    et_end=float(et_e),
                    init=float(init),
                    intlen=float(intlen),
                    first_record=0,
                    records=records,
                )
            )
        rec = int(nxt)
    return segs


@dataclass
class Ephemeris:
    """
    Ordered collection of Chebyshev segments. Later segments take precedence, like
    kernels furnished later in CSPICE (`planetmapper/base.py:939-977` sorts the kernel
    paths so that this ordering is deterministic).
    """

    segments: list = field(default_factory=list)  # ChebySegment / TleSegment

    @classmethod
    def from_spk_files(cls, paths: Iterable[str]) -> 'Ephemeris':
        segs: list = []
        for p in paths:
            segs.extend(read_spk_segments(p))
        return cls(segs)

    def _find(self, target: int, et: float):
        for seg in reversed(self.segments):
            if seg.target == target and seg.covers(et):
                return seg
        raise KeyError(f'no ephemeris data for body {target} at et={et}')

    def ssb_state(self, body: int, et: float):
        """(pos, vel, acc) of `body` relative to the solar system barycentre, J2000."""
        pos = np.zeros(3)
        vel = np.zeros(3)
        acc = np.zeros(3)
        cur = body
        hops = 0
        while cur != 0:
            seg = self._find(cur, et)
            if seg.frame != 1:
                raise ValueError('only J2000 (frame 1) segments are supported')
            p, v, a = seg.state(et)
            pos += p
            vel += v
            acc += a
            cur = seg.center
            hops += 1
            if hops > 8:
                raise RuntimeError('ephemeris chain too long')
        return pos, vel, acc

    def ssb_motion(self, body: int, et: float):
        """(pos, d pos / dt, d2 pos / dt2) of `body` wrt the SSB: `ChebySegment.motion` along the chain."""
        pos, vel, acc = np.zeros(3), np.zeros(3), np.zeros(3)
        cur, hops = body, 0
        while cur != 0:
            seg = self._find(cur, et)
            if seg.frame != 1:
                raise ValueError('only J2000 (frame 1) segments are supported')
            p, v, a = seg.motion(et)
            pos += p
            vel += v
            acc += a
            cur = seg.center
            hops += 1
            if hops > 8:
                raise RuntimeError('ephemeris chain too long')
        return pos, vel, acc

    def trimmed(self, bodies: Iterable[int], et_lo: float, et_hi: float) -> 'Ephemeris':
        """Subset sufficient to evaluate `bodies` wrt the SSB within [et_lo, et_hi]."""
        keep: list = []
        seen: set[int] = set()
        todo = list(bodies)
        while todo:
            b = todo.pop()
            if b == 0 or b in seen:
                continue
            seen.add(b)
            seg = self._find(b, 0.5 * (et_lo + et_hi))
            keep.append(seg.trimmed(et_lo, et_hi))
            todo.append(seg.center)
        return Ephemeris(keep)

    def to_json(self) -> dict:
        return {'segments': [s.to_json() for s in self.segments]}

    @classmethod
    def from_json(cls, d: dict) -> 'Ephemeris':
        return cls([(TleSegment if s.get('spk_type') == 10 else ChebySegment).from_json(s) for s in d['segments']])

    def dump(self, path: str) -> None:
        with open(path, 'w', encoding='utf-8') as f:
            json.dump(self.to_json(), f)

    @classmethod
    def load(cls, path: str) -> 'Ephemeris':
        with open(path, encoding='utf-8') as f:
            return cls.from_json(json.load(f))
