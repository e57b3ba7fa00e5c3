"""The knot / smoothing-parameter search of one plane, traced (PM_SM_DEBUG) on the library named by PLANETMAPPER_HIP_LIB
and held against the oracle: python tools/probes/smoothing_trace.py [plane [interp_k [s_factor [size]]]]"""
import os, sys
os.environ['PM_DEBUG_ENV'] = '1'; os.environ['PM_SM_DEBUG'] = '1'
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario
from oracle import oracle
import test_gpu_splines_cube_scale as T

pl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sf = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
sz = int(sys.argv[4]) if len(sys.argv) > 4 else 512
g = load_scenario('jupiter_hst_2005')
e = Engine(0)
cube, states = T.make_cube(12, sz, sz, seed=5 + sz)
xm, ym = T.setup_maps(e, oracle, g, sz, sz, deg=2.0)
s = sf * sz * sz
a = T.map_resident(e, cube[pl:pl + 1], xm, ym, (k, k), True, spline_smoothing=s)
b = oracle.map_cube(cube[pl:pl + 1], xm, ym, (k, k), True, spline_smoothing=s)
fin = np.isfinite(b)
print('lib', os.environ.get('PLANETMAPPER_HIP_LIB', 'in-tree'), 'state', states[pl], 'max |HIP - oracle|', float(np.max(np.abs(a[fin] - b[fin]))), 'scale', float(np.abs(b[fin]).max()))
try:
    from scipy.interpolate import RectBivariateSpline
    c = oracle.clean_nans(cube[pl])
    sp = RectBivariateSpline(np.arange(sz), np.arange(sz), c, kx=k, ky=k, s=s)
    tx, ty = sp.get_knots()
    print('scipy knots', len(tx), len(ty), 'fp', sp.get_residual())
except Exception as ex:
    print('scipy:', ex)
e.close()
