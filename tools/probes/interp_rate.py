"""Per-plane cost of every interpolation of map_img on a device-resident cube (64 planes of 1024^2 f64 -> the 1 deg map and a
0.1 deg map): python tools/probes/interp_rate.py -> one JSON line per (interpolation, map)."""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
gen = torch.Generator(device='cuda').manual_seed(5)
cube = torch.randn((P, sz, sz), generator=gen, device='cuda', dtype=torch.float64)
for deg in (1.0, 0.1):
    lon = np.arange(deg / 2, 360, deg)[::-1] if g.west_positive else np.arange(deg / 2, 360, deg)
    lat = np.arange(-90 + deg / 2, 90, deg)
    lon_g, lat_g = np.meshgrid(lon, lat); n0, n1 = lon_g.shape
    lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
    xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda'); ym = torch.empty_like(xm)
    e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
    out = torch.empty((P, n0, n1), dtype=torch.float64, device='cuda')
    for interp, kw in ((('nearest', {}), ('linear', {}), ('quadratic', {}), ('cubic', {}), (5, {}), ('cubic', {'spline_smoothing': 1.0}), ('smooth', {})) if P <= 64 else (('linear', {}), ('cubic', {}), ('smooth', {}))):
        if interp == 'smooth': e.set_smooth_options(5, 10_000)
        try:
            e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, interp, True, **kw); e.synchronize()
        except Exception as ex:
            print(json.dumps({'interpolation': str(interp), 'deg': deg, 'error': str(ex)[:200]})); continue
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps): e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, interp, True, **kw)
        e.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(json.dumps({'interpolation': str(interp) + (' s=1' if kw else ''), 'deg': deg, 'planes': P, 'ms': round(dt * 1e3, 3), 'us_per_plane': round(dt / P * 1e6, 1),
                          'plane_GBps': round(P * sz * sz * 8 / dt / 1e9, 1)}), flush=True)
e.close()
