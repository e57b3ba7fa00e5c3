#!/usr/bin/env python3
"""
Geometry INPUTS for the motion-model tests (tests/test_motion_model.py), extracted from the SPICE
kernels the reference ships with its test-suite (tests/data/kernels: pck00010.tpc, de410s.bsp,
jup120_1996-2010.bsp). Run in the build container, where /root/reference exists:

    python tests/golden/make_motion_fixtures.py [/root/reference]

Data only: PCK constants of the target and the Chebyshev records (SPK types 2 / 3) within a day of the
epoch, as JSON with every float in hex. Nothing of the reference's source is read or copied.

    motion_jupiter_earth_1998.json  Jupiter (599, jup120) from Earth, 1998-07-01T12:00 TDB
    motion_jupiter_earth_2009.json  Jupiter from Earth near the end of jup120's coverage, 2009-08-14
    motion_mars_earth_2012.json     Mars (499, de410s) from Earth at 0.7 au, 2012-03-15: the fastest apparent
                                    motion and the largest light-time rate the bundled kernels offer
    motion_saturn_earth_2016.json   Saturn barycentre (6) as the centre of a Saturn spheroid, 2016-06-03
    motion_io_like_earth_2009.json  Io's PCK constants (periodic terms in W, RA, Dec) on Jupiter's ephemeris: the reference-binding test
    motion_moon_earth_2012.json     the Moon (301, de410s) from Earth: a real moon with its own segment (301 wrt 3) and periodic terms
(Jupiter / HST 2005 and Saturn / Earth 2005 are the package's own scenarios, planetmapper_amd/data/.)
"""

from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)

from planetmapper_amd import ephem  # noqa: E402

SPD = 86400.0
# (name, target id, body whose PCK constants are used, observer id, TDB seconds past J2000, kernels)
CASES = [
    ('jupiter_earth_1998', 599, 599, 399, (-1.5 * 365.25 + 0.0) * SPD, ('de410s', 'jup120')),
    ('jupiter_earth_2009', 599, 599, 399, (9.0 * 365.25 + 225.3) * SPD, ('de410s', 'jup120')),
    ('mars_earth_2012', 499, 499, 399, (12.0 * 365.25 + 74.2) * SPD, ('de410s',)),
    ('saturn_earth_2016', 6, 699, 399, (16.0 * 365.25 + 154.6) * SPD, ('de410s',)),
    # (the bundled kernels hold no moon of Jupiter: Io's orientation model - trigonometric terms in W, RA and Dec - on Jupiter's path)
    ('io_like_earth_2009', 599, 501, 399, (9.0 * 365.25 + 225.3) * SPD, ('de410s', 'jup120')),
    ('moon_earth_2012', 301, 301, 399, (12.0 * 365.25 + 74.2) * SPD, ('de410s',)),
]


def main(ref: str) -> None:
    kdir = os.path.join(ref, 'tests', 'data', 'kernels')
    files = {'de410s': os.path.join(kdir, 'file with spaces de410s.bsp'), 'jup120': os.path.join(kdir, 'jup120_1996-2010.bsp')}
    pool = ephem.parse_text_kernel(open(os.path.join(kdir, 'pck00010.tpc')).read())
    for name, target, pck_body, observer, et, kernels in CASES:
        eph = ephem.Ephemeris.from_spk_files([files[k] for k in kernels])  # later kernels take precedence (base.py:939-977)
        mini = eph.trimmed([target, 10, observer], et - SPD, et + SPD)
        rot = ephem.RotationModel.from_pool(pool, pck_body)
        out = {
            'description': f'target {target} (PCK of body {pck_body}) seen from {observer} at et = {et!r}; kernels: {", ".join(kernels)}',
            'target_id': target,
            'observer_id': observer,
            'et': et,
            'pck': rot.to_json(),
            'ephemeris': mini.to_json(),
        }
        path = os.path.join(HERE, f'motion_{name}.json')
        with open(path, 'w', encoding='utf-8') as f:
            json.dump(out, f, indent=0)
        print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
