"""
SPICE-free host geometry provider: UTC -> ET, kernel discovery order,
text PCK + DAF/SPK type 2/3 reading. Uses the reference's test kernels when the reference
checkout is present (build container); the committed mini-ephemeris otherwise.
"""

import os

import numpy as np
import pytest

from planetmapper_amd import ephem
from planetmapper_amd.kernels import body_id, geometry_from_kernels, sort_kernel_paths
from planetmapper_amd.scenarios import scenario_info
from planetmapper_amd.timeconv import utc2et

REF_KERNELS = '/root/reference/tests/data/kernels'


def test_utc2et_kat():
    """tests/test_body.py:110: et of 2005-01-01T00:00:00"""
    assert utc2et('2005-01-01T00:00:00') == 157809664.1839331
    assert utc2et('2005-01-01') == 157809664.1839331
    assert utc2et('2005-01-01 00:00:00.000') == 157809664.1839331
    assert utc2et('2000-01-01T12:00:00') == pytest.approx(64.18392728473108, abs=1e-9)
    import datetime

    assert utc2et(datetime.datetime(2005, 1, 1, tzinfo=datetime.timezone.utc)) == 157809664.1839331
    assert utc2et(53371.0) == 157809664.1839331  # MJD
    # a leap second boundary: 2005-12-31T23:59:59 -> 2006-01-01T00:00:00 is 2 SI seconds
    assert utc2et('2006-01-01') - utc2et('2005-12-31T23:59:59') == pytest.approx(2.0, abs=1e-6)
    with pytest.raises(ValueError):
        utc2et('not a date')


def test_body_ids_and_sorting():
    assert body_id('jupiter') == 599 and body_id(' Saturn ') == 699 and body_id('599') == 599 and body_id(3) == 3
    with pytest.raises(ValueError):
        body_id('<<< test >>>')
    # planetmapper/base.py:939-950 docstring examples
    assert sort_kernel_paths(['a/kernel.bsp', 'x/y/z/kernel.bsp']) == ['x/y/z/kernel.bsp', 'a/kernel.bsp']
    assert sort_kernel_paths(['b.bsp', 'a.bsp']) == ['a.bsp', 'b.bsp']


def test_text_kernel_parser():
    pool = ephem.parse_text_kernel(
        "comment\n\\begindata\n BODY599_RADII = ( 71492 71492 66854 )\n BODY599_PM = ( 284.95 870.5360000 0. )\n"
        " X = 1.5D2\n NAME = 'abc'\n X += ( 2 )\n\\begintext\n BODY599_RADII = ( 1 2 3 )\n"
    )
    assert pool['BODY599_RADII'] == [71492.0, 71492.0, 66854.0]
    assert pool['X'] == [150.0, 2.0] and pool['NAME'] == ['abc']


def test_mini_ephemeris_round_trip():
    """The committed Chebyshev subset reproduces the full-kernel sanity values of SURVEY B.2."""
    info = scenario_info('jupiter_hst_2005')
    eph = ephem.Ephemeris.from_json(info['ephemeris'])
    et = info['et']
    p, v, _ = eph.ssb_state(5, et)
    assert np.allclose(p, (-8.09124152e8, -9.73561763e7, -2.20324604e7), rtol=1e-8)
    assert np.allclose(v, (1.41061861, -11.34273193, -4.89625182), rtol=1e-8)
    p, v, _ = eph.ssb_state(10, et)
    assert np.allclose(p, (642956.20203259, -34905.06225126, -31992.48353122), rtol=1e-9)
    # acceleration = derivative of velocity
    h = 10.0
    a_fd = (eph.ssb_state(599, et + h)[1] - eph.ssb_state(599, et - h)[1]) / (2 * h)
    assert np.allclose(eph.ssb_state(599, et)[2], a_fd, rtol=1e-5, atol=1e-12)
    rot = ephem.RotationModel.from_json(info['pck'])
    R = rot.matrix(et)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.linalg.det(R) == pytest.approx(1.0)
    assert rot.spin_rate(et) == pytest.approx(np.deg2rad(870.536) / 86400, rel=1e-9)


@pytest.mark.skipif(not os.path.isdir(REF_KERNELS), reason='reference checkout not present')
def test_geometry_from_reference_kernel_directory():
    """
    Body('Jupiter', '2005-01-01', observer='EARTH') from the kernel directory of the
    reference's test-suite (recursive discovery, jup120 overriding de410s): the geometry must
    agree with the HST-based golden geometry up to the Earth-HST parallax (< 7000 km of 8e8 km).
    """
    g = geometry_from_kernels('jupiter', '2005-01-01T00:00:00', 'earth', REF_KERNELS)
    assert g.et == 157809664.1839331
    assert list(g.radii) == [71492.0, 71492.0, 66854.0] and g.west_positive == 1
    from planetmapper_amd.scenarios import load_scenario

    gh = load_scenario('jupiter_hst_2005')
    assert abs(g.lt_c - gh.lt_c) < 7000 / 299792.458
    assert np.allclose(g.R0[:], gh.R0[:], atol=1e-5)  # the light times differ by up to 0.02 s
    assert np.allclose(g.VT[:], gh.VT[:], atol=1e-6)
    d = np.array(g.T0[:]) - np.array(gh.T0[:])
    assert np.linalg.norm(d) < 7000.0
    # Earth's velocity from the SPK is the observer velocity
    assert np.allclose(g.VO[:], (-29.75984263, -5.1102393, -2.21484891), atol=1e-6)
    with pytest.raises(ValueError):
        geometry_from_kernels('jupiter', '2005-01-01', 'earth', spk=[], pck=None)
