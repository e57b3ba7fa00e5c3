"""
Calibration of the parity bars (tests/parity.py) on the CPU: two builds of the SAME
oracle source - strict IEEE evaluation order vs FMA contraction - are both valid
restatements of the reference, so their disagreement is the floating-point noise floor
of the reference's formulation. The conditioned tolerances must hold between them,
masks must be identical, and the flat 1e-9 deg bar must hold for >= 98 % of pixels.
"""

import numpy as np

from oracle import oracle
from parity import compare_planes


def test_strict_vs_fma_builds_of_the_oracle(jupiter):
    sz = 384
    x0 = y0 = (sz - 1) / 2
    disc = oracle.make_disc(x0, y0, 0.9 * x0, 12.5, sz, sz)
    names = oracle.PLANE_NAMES
    try:
        oracle.use_variant(None)
        a = oracle.backplanes_img(jupiter, disc, names)
        oracle.use_variant('fma')
        b = oracle.backplanes_img(jupiter, disc, names)
    finally:
        oracle.use_variant(None)
    stats = compare_planes(b, a, names, jupiter)
    # the floor is real: longitudes near the limb differ by more than the flat bar
    assert stats['LON-GRAPHIC'][0] > 1e-9
    assert stats['LON-GRAPHIC'][1] > 0.98
    assert stats['PHASE'][0] < 1e-11
