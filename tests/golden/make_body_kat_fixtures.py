#!/usr/bin/env python3
"""
Known-answer tables of the reference's point-transform tests (tests/test_body.py, geometry
Body('Jupiter', observer='HST', utc='2005-01-01T00:00:00')): every list literal assigned inside the
listed test functions is lifted as data - (inputs, expected) tuples, nothing else - into
tests/golden/kat_body_transforms.json. tests/test_reference_kats.py replays them through the
BodyXY methods of the same names (oracle-backed on CPU, HIP engine on the GPU).

    python tests/golden/make_body_kat_fixtures.py [/root/reference]
"""

import json
import math
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

TESTS = [
    'test_lonlat2radec', 'test_radec2lonlat', 'test_angular_radec', 'test_angular_lonlat', 'test_km_radec',
    'test_km_lonlat', 'test_km_angular', 'test_limb_coordinates_from_radec', 'test_if_lonlat_visible',
    'test_illimination_angles_from_lonlat', 'test_azimuth_angle_from_lonlat', 'test_local_solar_time',
    'test_if_lonlat_illuminated', 'test_ring_plane_coordinates', 'test_radial_velocity_from_lonlat',
    'test_distance_from_lonalt', 'test_graphic_centric_lonlat',
]  # fmt: skip


class _NP:
    nan = float('nan')
    inf = float('inf')

    @staticmethod
    def array(x):
        return x


def _jsonable(v):
    if isinstance(v, float):
        if math.isnan(v):
            return 'nan'
        if math.isinf(v):
            return 'inf' if v > 0 else '-inf'
        return v
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    return v


def tables_of(func_text: str) -> dict:
    """name -> list literal, for every `name[: annotation] = [ ... ]` statement of the function"""
    out = {}
    for m in re.finditer(r'^\s{8}(\w+)(?:\s*:\s*[^=]*?)?\s*=\s*\[', func_text, flags=re.M | re.S):
        i = m.end() - 1
        depth, j = 0, i
        while True:
            c = func_text[j]
            depth += {'[': 1, ']': -1}.get(c, 0)
            j += 1
            if depth == 0:
                break
        try:
            val = eval(func_text[i:j], {'__builtins__': {}}, {'np': _NP, 'nan': _NP.nan, 'inf': _NP.inf, 'array': _NP.array})
        except Exception:  # a table built from expressions of the test itself: not a literal, skipped
            continue
        name = m.group(1)
        k = name
        n = 2
        while k in out:
            k = f'{name}_{n}'
            n += 1
        out[k] = _jsonable(val)
    return out


def main(ref: str) -> None:
    text = open(os.path.join(ref, 'tests', 'test_body.py'), encoding='utf-8').read()
    fixture = {'source': 'tests/test_body.py value tables (planetmapper v1.14.0), one entry per test function'}
    for t in TESTS:
        i = text.index(f'    def {t}(self')
        j = text.find('\n    def ', i + 10)
        fixture[t] = tables_of(text[i:j])
        print(t, {k: len(v) for k, v in fixture[t].items()})
    with open(os.path.join(HERE, 'kat_body_transforms.json'), 'w', encoding='utf-8') as f:
        json.dump(fixture, f)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
