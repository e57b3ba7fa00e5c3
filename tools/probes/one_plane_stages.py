import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario
g = load_scenario('jupiter_hst_2005')
for sz in (1024, 4096):
    x0 = (sz - 1) / 2
    e = Engine(0); e.set_geometry(g); e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon = np.arange(0.5, 360, 1.0)[::-1]; lat = np.arange(-89.5, 90, 1.0)
    lon_g, lat_g = np.meshgrid(lon, lat)
    xy = e.backplanes_map(['PIXEL-X', 'PIXEL-Y'], np.ascontiguousarray(lon_g), np.ascontiguousarray(lat_g))
    xm, ym = xy['PIXEL-X'], xy['PIXEL-Y']
    img = np.random.default_rng(1).standard_normal((1, sz, sz))
    for _ in range(12): e.map_cube(img, xm, ym, 'linear', True)
    ts = []
    for _ in range(9):
        t = time.perf_counter(); e.map_cube(img, xm, ym, 'linear', True); ts.append(time.perf_counter() - t)
    print(sz, 'ms', round(float(np.median(ts)) * 1e3, 3), json.dumps({k: round(v, 3) for k, v in e.last_stages_ms().items()}))
    e.close()
