"""
UTC -> ET (TDB seconds past J2000) without CSPICE: the algorithm of `str2et` / `deltet`
that `BodyBase.__init__` uses (`planetmapper/base.py:818`):

    ET = (UTC - J2000_UTC) + DELTA_AT + DELTA_T_A + K sin(E),   E = M + EB sin(M),
    M = M0 + M1 * ET

with DELTA_AT the leap-second count. The constants and the leap-second table below are
the published values of the NAIF leap-seconds kernel naif0012.tls (valid until a new leap
second is announced); a different LSK can be supplied as a parsed text-kernel pool.
"""

from __future__ import annotations

import datetime
import math
import re

DELTA_T_A = 32.184
K = 1.657e-3
EB = 1.671e-2
M0, M1 = 6.239996, 1.99096871e-7

# (TAI - UTC, first UTC day it applies) -- naif0012.tls DELTET/DELTA_AT
LEAP_SECONDS = [
    (10, (1972, 1, 1)), (11, (1972, 7, 1)), (12, (1973, 1, 1)), (13, (1974, 1, 1)),
    (14, (1975, 1, 1)), (15, (1976, 1, 1)), (16, (1977, 1, 1)), (17, (1978, 1, 1)),
    (18, (1979, 1, 1)), (19, (1980, 1, 1)), (20, (1981, 7, 1)), (21, (1982, 7, 1)),
    (22, (1983, 7, 1)), (23, (1985, 7, 1)), (24, (1988, 1, 1)), (25, (1990, 1, 1)),
    (26, (1991, 1, 1)), (27, (1992, 7, 1)), (28, (1993, 7, 1)), (29, (1994, 7, 1)),
    (30, (1996, 1, 1)), (31, (1997, 7, 1)), (32, (1999, 1, 1)), (33, (2006, 1, 1)),
    (34, (2009, 1, 1)), (35, (2012, 7, 1)), (36, (2015, 7, 1)), (37, (2017, 1, 1)),
]  # fmt: skip

_MONTHS = {m: i + 1 for i, m in enumerate(
    ['JAN', 'FEB', 'MAR', 'APR', 'MAY', 'JUN', 'JUL', 'AUG', 'SEP', 'OCT', 'NOV', 'DEC'])}  # fmt: skip
_J2000 = datetime.datetime(2000, 1, 1, 12, 0, 0)


def leap_table_from_pool(pool: dict) -> list[tuple[int, tuple[int, int, int]]]:
    """`DELTET/DELTA_AT` of a parsed LSK (`ephem.parse_text_kernel`) -> LEAP_SECONDS format."""
    vals = pool['DELTET/DELTA_AT']
    out = []
    for n, d in zip(vals[0::2], vals[1::2]):
        m = re.match(r'@(\d{4})-([A-Z]{3})-(\d+)', str(d).upper())
        if not m:
            raise ValueError(f'cannot parse leap second date {d!r}')
        out.append((int(n), (int(m.group(1)), _MONTHS[m.group(2)], int(m.group(3)))))
    return out


def parse_utc(utc) -> datetime.datetime:
    """ISO-like UTC strings (`2005-01-01`, `2005-01-01T00:00:00.5`), datetimes, or MJD floats."""
    if isinstance(utc, (int, float)):
        return datetime.datetime(1858, 11, 17) + datetime.timedelta(days=float(utc))  # MJD
    if isinstance(utc, datetime.datetime):
        if utc.tzinfo is not None:
            utc = utc.astimezone(datetime.timezone.utc).replace(tzinfo=None)
        return utc
    s = str(utc).strip().rstrip('Z')
    for fmt in ('%Y-%m-%dT%H:%M:%S.%f', '%Y-%m-%dT%H:%M:%S', '%Y-%m-%d %H:%M:%S.%f', '%Y-%m-%d %H:%M:%S',
                '%Y-%m-%dT%H:%M', '%Y-%m-%d'):  # fmt: skip
        try:
            return datetime.datetime.strptime(s, fmt)
        except ValueError:
            continue
    raise ValueError(f'unsupported UTC format {utc!r} (use ISO 8601)')


def utc2et(utc, leap_seconds=None) -> float:
    """TDB seconds past J2000 of a UTC epoch (`spice.str2et`)."""
    table = LEAP_SECONDS if leap_seconds is None else leap_seconds
    t = parse_utc(utc)
    delta_at = 0
    for n, (y, m, d) in table:
        if t >= datetime.datetime(y, m, d):
            delta_at = n
    if delta_at == 0:
        raise ValueError('UTC epochs before 1972-01-01 are not supported')
    dt = t - _J2000
    # exact integer arithmetic for the whole seconds, then the sub-second part
    utc_s = dt.days * 86400 + dt.seconds + dt.microseconds * 1e-6
    tai = utc_s + delta_at
    et = tai + DELTA_T_A
    for _ in range(3):  # ET appears on both sides through M(ET); converges immediately
        mean_anom = M0 + M1 * et
        ecc_anom = mean_anom + EB * math.sin(mean_anom)
        et = tai + DELTA_T_A + K * math.sin(ecc_anom)
    return et


def et2utc(et: float, leap_seconds=None) -> datetime.datetime:
    """UTC datetime of a TDB epoch (`spice.et2datetime` of base.py:826); microsecond resolution."""
    table = LEAP_SECONDS if leap_seconds is None else leap_seconds
    mean_anom = M0 + M1 * et
    ecc_anom = mean_anom + EB * math.sin(mean_anom)
    tai = et - DELTA_T_A - K * math.sin(ecc_anom)
    delta_at = table[-1][0]
    for _ in range(3):  # the leap-second count depends on the UTC date it produces
        t = _J2000 + datetime.timedelta(seconds=tai - delta_at)
        new = 0
        for n, (y, m, d) in table:
            if t >= datetime.datetime(y, m, d):
                new = n
        if new == 0:
            raise ValueError('epochs before 1972-01-01 are not supported')
        if new == delta_at:
            break
        delta_at = new
    us = round((tai - delta_at) * 1e6)
    return _J2000 + datetime.timedelta(microseconds=us)
