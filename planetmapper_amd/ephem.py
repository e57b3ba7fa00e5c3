"""
SPICE-free host ephemeris + orientation provider.

The reference obtains every geometric constant from NAIF CSPICE through spiceypy
(`planetmapper/base.py:795-839`, `planetmapper/body.py:501-606`). CSPICE is a
third-party dependency that is not vendored by the reference, so this module restates
the small part of it that the *host side* of the hot path needs:

* text kernels (PCK, `pck00010.tpc` style): ``BODYnnn_RADII``, ``_POLE_RA``,
  ``_POLE_DEC``, ``_PM``, ``_NUT_PREC_*`` -> IAU rotation model J2000 -> body-fixed;
* binary DAF/SPK ephemerides, Chebyshev segment types 2 and 3 (planets, barycentres,
  Sun, Earth) -> position / velocity / acceleration relative to the SSB.

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


def read_spk_segments(path: str) -> list[ChebySegment]:
    """
    Read every type 2 / type 3 segment of a binary DAF/SPK file (other types are
    skipped). Layout: NAIF "DAF Required Reading" / "SPK Required Reading".
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

    segments: list[ChebySegment] = field(default_factory=list)

    @classmethod
    def from_spk_files(cls, paths: Iterable[str]) -> 'Ephemeris':
        segs: list[ChebySegment] = []
        for p in paths:
            segs.extend(read_spk_segments(p))
        return cls(segs)

    def _find(self, target: int, et: float) -> ChebySegment:
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
        keep: list[ChebySegment] = []
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
        return cls([ChebySegment.from_json(s) for s in d['segments']])

    def dump(self, path: str) -> None:
        with open(path, 'w', encoding='utf-8') as f:
            json.dump(self.to_json(), f)

    @classmethod
    def load(cls, path: str) -> 'Ephemeris':
        with open(path, encoding='utf-8') as f:
            return cls.from_json(json.load(f))
