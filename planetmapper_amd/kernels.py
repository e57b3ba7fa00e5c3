"""
Building a geometry block straight from NAIF kernel FILES, without CSPICE:

    g = geometry_from_kernels('jupiter', '2005-01-01', 'earth', '~/spice_kernels')
    body = BodyXY('jupiter', '2005-01-01', observer='earth', geometry=g, sz=500)
    # or in one step:  BodyXY('jupiter', '2005-01-01', observer='earth', kernels='~/spice_kernels', sz=500)

This is the host-side counterpart of `SpiceBase.load_spice_kernels` + `Body.__init__`
(`planetmapper/base.py:554-611, 909-1079`, `planetmapper/body.py:323-606`) for the kernel
types it can read: text PCK (`*.tpc`), text LSK (`*.tls`, optional) and binary SPK with
Chebyshev segments of type 2 / 3 (`*.bsp`: the DE planetary ephemerides and the satellite
ephemerides of the giant planets) or type 10 (two-line elements of a near-earth satellite: HST, the
reference's canonical observer, `tests/test_body.py:29-31`). Observers with other SPK types (e.g. a
spacecraft's type 13 / 18 reconstruction) need spiceypy on the host.
"""

from __future__ import annotations

import glob
import os
from pathlib import Path

from .ephem import Ephemeris, RotationModel, parse_text_kernel, read_spk_segments
from .geometry import GeometryBuilder, PMGeometry
from .timeconv import leap_table_from_pool, utc2et

# NAIF ids of the bodies a planetary observer is likely to name (naif_ids.req)
BODY_IDS = {
    'SOLAR SYSTEM BARYCENTER': 0, 'SSB': 0, 'SUN': 10,
    'MERCURY BARYCENTER': 1, 'VENUS BARYCENTER': 2, 'EARTH BARYCENTER': 3, 'EMB': 3, 'MARS BARYCENTER': 4,
    'JUPITER BARYCENTER': 5, 'SATURN BARYCENTER': 6, 'URANUS BARYCENTER': 7, 'NEPTUNE BARYCENTER': 8,
    'PLUTO BARYCENTER': 9,
    'MERCURY': 199, 'VENUS': 299, 'EARTH': 399, 'MOON': 301, 'MARS': 499, 'PHOBOS': 401, 'DEIMOS': 402,
    'JUPITER': 599, 'IO': 501, 'EUROPA': 502, 'GANYMEDE': 503, 'CALLISTO': 504, 'AMALTHEA': 505,
    'SATURN': 699, 'MIMAS': 601, 'ENCELADUS': 602, 'TETHYS': 603, 'DIONE': 604, 'RHEA': 605, 'TITAN': 606,
    'HYPERION': 607, 'IAPETUS': 608, 'PHOEBE': 609,
    'URANUS': 799, 'ARIEL': 701, 'UMBRIEL': 702, 'TITANIA': 703, 'OBERON': 704, 'MIRANDA': 705,
    'NEPTUNE': 899, 'TRITON': 801, 'PLUTO': 999, 'CHARON': 901,
    'HST': -48, 'HUBBLE SPACE TELESCOPE': -48,
}  # fmt: skip


def body_id(name: str | int) -> int:
    """`spice.bods2c` for the table above; integers / numeric strings pass through."""
    if isinstance(name, int):
        return name
    s = str(name).strip().upper()
    try:
        return int(s)
    except ValueError:
        pass
    if s not in BODY_IDS:
        raise ValueError(f'unknown body name {name!r} (give its NAIF id instead)')
    return BODY_IDS[s]


def sort_kernel_paths(paths):
    """Deeper paths first, then dirname / basename: `planetmapper/base.py:939-977`."""
    return sorted(
        paths,
        key=lambda p: (-len(Path(p).resolve().parts), os.path.dirname(p), os.path.basename(p), os.path.normpath(p), p),
    )


def find_kernels(kernel_path: str) -> dict[str, list[str]]:
    """`**/*.bsp`, `**/*.tpc`, `**/*.tls` under `kernel_path`, in the reference's load order."""
    root = os.path.expandvars(os.path.expanduser(kernel_path))
    out = {}
    for ext in ('bsp', 'tpc', 'tls'):
        out[ext] = sort_kernel_paths(glob.glob(os.path.join(root, '**', f'*.{ext}'), recursive=True))
    return out


def geometry_from_kernels(target, utc, observer='EARTH', kernel_path: str | None = None, *,
                          spk=None, pck=None, lsk=None, observer_velocity=None) -> PMGeometry:
    """
    Geometry block of `Body(target, utc, observer)` from kernel files. Either `kernel_path`
    (searched recursively like the reference) or explicit `spk` (list) / `pck` / `lsk` paths.
    Later kernels take precedence, as with `spice.furnsh`.
    """
    if kernel_path is not None:
        found = find_kernels(kernel_path)
        spk = list(spk or []) + found['bsp']
        pck = pck or (found['tpc'][-1] if found['tpc'] else None)
        lsk = lsk or (found['tls'][-1] if found['tls'] else None)
    if not spk or not pck:
        raise ValueError('need at least one SPK (*.bsp) and one text PCK (*.tpc) kernel')
    leap = None
    if lsk is not None:
        leap = leap_table_from_pool(parse_text_kernel(open(lsk, encoding='latin-1').read()))
    et = utc2et(utc, leap)
    segments = []
    for path in spk:
        try:
            segments.extend(read_spk_segments(path))
        except ValueError:
            continue  # not a DAF/SPK file (the reference's test kernels include such stubs)
    eph = Ephemeris(segments)
    pool = parse_text_kernel(open(pck, encoding='latin-1').read())
    tid = body_id(target)
    rot = RotationModel.from_pool(pool, tid)
    if not rot.pm:
        raise ValueError(f'the PCK has no orientation constants for body {tid}')
    return GeometryBuilder(eph, rot, tid).build(et, observer_id=body_id(observer), observer_velocity=observer_velocity)
