"""
Fixture of tests/test_smoothing_knife_edge.py: the one plane of the round-5 smoothing soak (seed 111241, case 3, plane 2 of
tests/soak_fuzz's generator: 84 x 76, degrees (2, 2), s = 507.57) on which the device's smoothing spline is not scipy's, and
ONE-ULP changes of ONE pixel of it under which scipy's own RectBivariateSpline changes its knot set the same way.
Needs numpy + scipy only (no reference code): python tests/golden/make_smoothing_knife_edge_fixture.py
"""
import os
import sys

import numpy as np
from scipy.interpolate import RectBivariateSpline

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE))]


def soak_case(seed, want):
    """the generator of tests/soak_fuzz.py --only smoothing / tools/probes/smoothing_case.py, on the CPU"""
    from oracle import oracle
    from planetmapper_amd.scenarios import load_scenario

    jupiter = load_scenario('jupiter_hst_2005')
    rng = np.random.default_rng(seed)
    for case in range(8):
        ny, nx = int(rng.integers(16, 90)), int(rng.integers(16, 90))
        ky, kx = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        n_planes = int(rng.integers(2, 7))
        yy, xx = np.mgrid[0:ny, 0:nx]
        sigma = float(rng.choice([0.3, 1.0, 4.0]))
        cube = np.empty((n_planes, ny, nx))
        for p in range(n_planes):
            cube[p] = np.sin(xx / rng.uniform(3, 15)) * np.cos(yy / rng.uniform(3, 15)) * rng.uniform(1, 20) + sigma * rng.standard_normal((ny, nx))
        flavour = int(rng.integers(0, 4))
        if flavour == 1:
            cube[0][rng.random((ny, nx)) < 0.03] = np.nan
        elif flavour == 2 and n_planes > 1:
            cube[1][ny // 3 : ny // 3 + 4, nx // 4 : nx // 4 + 5] = np.nan
            cube[0][rng.random((ny, nx)) < 0.01] = np.inf
        elif flavour == 3:
            cube[n_planes - 1][:] = np.nan
        x0, y0 = float(rng.uniform(0.35, 0.65) * nx), float(rng.uniform(0.35, 0.65) * ny)
        r0 = float(rng.uniform(0.25, 0.6) * min(nx, ny))
        rot = float(rng.uniform(0, 6.28))
        d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny)
        d.rotation_rad = rot
        lon, lat = oracle.rectangular_grid(jupiter, float(rng.choice([5.0, 9.0, 15.0])))
        xm, ym = oracle.xy_map(jupiter, d, lon, lat)
        if np.isfinite(xm).sum() < 20:
            continue
        s = float(rng.uniform(0.8, 3.0)) * ny * nx * sigma * sigma
        if case == want:
            return cube, (ky, kx), s
        rng.integers(0, n_planes)
    raise SystemExit('case not reached')


def knots(z, k, s):
    sp = RectBivariateSpline(np.arange(z.shape[0]), np.arange(z.shape[1]), z, kx=k[0], ky=k[1], s=s)
    ty, tx = sp.get_knots()
    return np.asarray(ty), np.asarray(tx)


if __name__ == '__main__':
    cube, k, s = soak_case(111241, 3)
    plane = cube[2]
    ty, tx = knots(plane, k, s)
    r = np.random.default_rng(0)
    flips = []
    tried = 0
    while len(flips) < 3 and tried < 3000:
        tried += 1
        i, j = int(r.integers(0, plane.shape[0])), int(r.integers(0, plane.shape[1]))
        up = bool(r.random() < 0.5)
        z = plane.copy()
        z[i, j] = np.nextafter(z[i, j], np.inf if up else -np.inf)
        ty2, tx2 = knots(z, k, s)
        if len(ty2) != len(ty) or np.any(ty2 != ty) or len(tx2) != len(tx) or np.any(tx2 != tx):
            flips.append((i, j, 1 if up else -1))
            print('flip', i, j, up, sorted(set(ty2) - set(ty)), sorted(set(ty) - set(ty2)), 'after', tried, 'trials')
    assert flips, 'no flip found'
    np.savez_compressed(os.path.join(HERE, 'smoothing_knife_edge.npz'), plane=plane, degrees=np.array(k), s=np.array(s), knots_y=ty, knots_x=tx,
                        flips=np.array(flips), trials=np.array(tried))
    print('written: base y knots', ty[k[0] + 1 : -k[0] - 1], 'flips', flips, 'trials', tried)
    # the three planes of the round-6 soak (191 000 fits) beyond the 1e-7 bar without a knife edge: ill-conditioned fits
    ill = {}
    for seed, case, pl in ((201558, 0, 0), (201514, 0, 0), (201498, 4, 2)):
        cube, k, s = soak_case(seed, case)
        ill[f'plane_{seed}'] = cube[pl]
        ill[f'degrees_{seed}'] = np.array(k)
        ill[f's_{seed}'] = np.array(s)
    healthy, k, s = soak_case(5, 2)
    ill['plane_healthy'], ill['degrees_healthy'], ill['s_healthy'] = healthy[1], np.array(k), np.array(s)  # (plane 0 of that cube has NaN pixels)
    np.savez_compressed(os.path.join(HERE, 'smoothing_ill_conditioned.npz'), **ill)
