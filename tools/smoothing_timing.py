#!/usr/bin/env python3
"""Time the interpolation modes of map_img on one GPU: P planes of sz x sz -> 1 deg map (GPU box)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

sz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = load_scenario('jupiter_hst_2005')
eng = Engine(0)
eng.set_geometry(g)
x0 = (sz - 1) / 2
eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
lons = np.arange(0.5, 360, 1.0)[::-1]
lats = np.arange(-89.5, 90, 1.0)
lon, lat = np.meshgrid(lons, lats)
xm, ym = eng.xy_map(lon, lat)
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:sz, 0:sz]
mu = np.sqrt(np.clip(1 - ((xx - x0) ** 2 + (yy - x0) ** 2) / (0.9 * x0) ** 2, 0, None))
cube = mu[None] + 0.05 * rng.standard_normal((P, sz, sz))
cube[rng.random(cube.shape) < 1e-3] = np.nan
npx = sz * sz
for name, kw in (('nearest', {}), ('linear', {}), ('cubic', {}), ('smooth', {}),
                 ('linear', dict(spline_smoothing=0.05**2 * npx)), ('cubic', dict(spline_smoothing=0.05**2 * npx))):
    eng.map_cube(cube[:1], xm, ym, name, True, **kw)
    t0 = time.perf_counter()
    out = eng.map_cube(cube, xm, ym, name, True, **kw)
    dt = time.perf_counter() - t0
    print(json.dumps({'interpolation': name, **kw, 'planes': P, 'size': sz, 'host_call_ms_per_plane': round(dt / P * 1e3, 2),
                      'finite': int(np.isfinite(out).sum())}))
