"""Host-fed cubes through the interpolations that need whole planes on the device (splines, 'smooth'): Engine.map_cube from a
numpy cube against the resident call - python tools/probes/host_spline_rate.py [planes] [size]"""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = load_scenario('jupiter_hst_2005'); x0 = (sz - 1) / 2
e = Engine(0); e.set_geometry(g); e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
rng = np.random.default_rng(2)
cube = rng.standard_normal((P, sz, sz))
pinned = e.pinned_empty(cube.shape); pinned[:] = cube
for deg in (1.0, 0.1) if P <= 16 else (1.0,):
    lon = np.arange(deg / 2, 360, deg)[::-1] if g.west_positive else np.arange(deg / 2, 360, deg)
    lat = np.arange(-90 + deg / 2, 90, deg)
    lon_g, lat_g = np.meshgrid(lon, lat)
    xy = e.backplanes_map(['PIXEL-X', 'PIXEL-Y'], np.ascontiguousarray(lon_g), np.ascontiguousarray(lat_g))
    xm, ym = xy['PIXEL-X'], xy['PIXEL-Y']
    dc = torch.from_numpy(cube).cuda(); dx = torch.from_numpy(xm).cuda(); dy = torch.from_numpy(ym).cuda()
    out = torch.empty((P,) + xm.shape, dtype=torch.float64, device='cuda')
    for interp in ('linear', 'cubic', 'smooth'):
        row = {'interpolation': interp, 'deg': deg, 'planes': P, 'cube_MB': round(cube.nbytes / 1e6)}
        for label, src in (('pageable', cube), ('pinned', pinned)):
            e.map_cube(src, xm, ym, interp, True)
            ts = []
            for _ in range(5):
                t = time.perf_counter(); r = e.map_cube(src, xm, ym, interp, True); ts.append(time.perf_counter() - t); del r
            row[f'ms_host_{label}'] = round(float(np.median(ts)) * 1e3, 2)
            row[f'GBps_cube_{label}'] = round(cube.nbytes / float(np.median(ts)) / 1e9, 1)
        e.map_cube_device(dc, np.float64, P, dx, dy, xm.shape[0], xm.shape[1], out, interp, True); e.synchronize()
        t = time.perf_counter()
        for _ in range(3): e.map_cube_device(dc, np.float64, P, dx, dy, xm.shape[0], xm.shape[1], out, interp, True)
        e.synchronize(); row['ms_resident'] = round((time.perf_counter() - t) / 3 * 1e3, 2)
        print(json.dumps(row), flush=True)
e.close()
