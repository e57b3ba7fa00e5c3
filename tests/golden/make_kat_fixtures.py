#!/usr/bin/env python3
"""
Known-answer vectors for `map_img(..., interpolation='smooth')`, lifted from the expected-value
tables of the reference's own tests (tests/test_body_xy.py: `test_map_img` 'smooth' subtest and
`test_map_img_smooth_interpolation`). Also the (image -> NaN-cleaned image) table of
`test_replace_nans_with_interpolated_values`. Only value tables are taken - data, not code; the inputs are rebuilt in tests/test_api_host.py from the recipe the
reference documents.

    python tests/golden/make_kat_fixtures.py [/root/reference]
"""

import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def tables(text: str) -> list[list]:
    """every run of consecutive `({kwargs}, [[...]]),` lines in the file"""
    out, cur = [], []
    for line in text.split('\n'):
        m = re.match(r'\s*\((\{.*?\}), (\[\[.*\]\])\),?\s*$', line)
        if m:
            kw = eval(m.group(1), {}, {})  # literal dict of ints
            exp = eval(m.group(2), {}, {'nan': None})  # NaN -> null in JSON
            cur.append({'kwargs': kw, 'expected': exp})
        elif cur:
            out.append(cur)
            cur = []
    return out


def literal_after(text: str, marker: str):
    """the Python list literal that follows `marker` (bracket matching; nan / inf allowed)"""
    i = text.index(marker) + len(marker)
    i = text.index('[', i)
    depth, j = 0, i
    while True:
        depth += {'[': 1, ']': -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            break
    return eval(text[i:j], {}, {'nan': float('nan'), 'inf': float('inf')})


def main(ref: str) -> None:
    text = open(os.path.join(ref, 'tests', 'test_body_xy.py'), encoding='utf-8').read()
    # (image, cleaned image) pairs of test_replace_nans_with_interpolated_values
    pairs = literal_after(text, 'images: list[tuple[list, list]] =')
    with open(os.path.join(HERE, 'kat_replace_nans.json'), 'w', encoding='utf-8') as f:
        json.dump(
            {
                'source': 'tests/test_body_xy.py test_replace_nans_with_interpolated_values value table',
                'cases': [{'image': a, 'cleaned': b} for a, b in pairs],
            },
            f,
        )
    print('replace_nans cases', len(pairs))
    # {spline_smoothing: expected map} table of test_map_img (linear interpolation, 45 deg map)
    i = text.index('expected_smoothings: dict[float, list] =')
    j = text.index('}', i)
    table = eval(text[text.index('{', i) : j + 1], {}, {'nan': None})
    with open(os.path.join(HERE, 'kat_map_img_smoothing.json'), 'w', encoding='utf-8') as f:
        json.dump(
            {
                'source': 'tests/test_body_xy.py test_map_img expected_smoothings value table',
                'cases': [{'smoothing': k, 'expected': v} for k, v in table.items()],
            },
            f,
        )
    print('smoothing cases', len(table))
    found = [t for t in tables(text) if any('smooth_oversample_by' in e['kwargs'] for e in t)]
    assert len(found) == 2, len(found)
    fixture = {
        'source': 'tests/test_body_xy.py expected-value tables of the smooth interpolation tests',
        'map_img_6x5': found[0],
        'map_img_90x120': found[1],
    }
    with open(os.path.join(HERE, 'kat_map_img_smooth.json'), 'w', encoding='utf-8') as f:
        json.dump(fixture, f)
    print({k: len(v) for k, v in fixture.items() if isinstance(v, list)})


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
