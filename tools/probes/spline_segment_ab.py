"""A/B of PM_OPT_SPLINE_SEGMENT (one lane per line vs segmented solves): python tools/probes/spline_segment_ab.py P SIZE [seg ...] -> JSON lines."""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd import _lib
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
segs = [int(v) for v in sys.argv[3:]] or [-1, 0, 64, 128, 256, 512]
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
gen = torch.Generator(device='cuda').manual_seed(5)
cube = torch.randn((P, sz, sz), generator=gen, device='cuda', dtype=torch.float64)
deg = 1.0
lon = np.arange(deg / 2, 360, deg)[::-1] if g.west_positive else np.arange(deg / 2, 360, deg)
lat = np.arange(-90 + deg / 2, 90, deg)
lon_g, lat_g = np.meshgrid(lon, lat); n0, n1 = lon_g.shape
lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda'); ym = torch.empty_like(xm)
e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
out = torch.empty((P, n0, n1), dtype=torch.float64, device='cuda')
ref = {}
for interp in ('cubic', 'quadratic', 5):
    for seg in segs:
        e.set_option(_lib.PM_OPT_SPLINE_SEGMENT, seg)
        e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, interp, True); e.synchronize()
        used = e.get_option(_lib.PM_OPT_LAST_SPLINE_SEGMENT)
        res = out.clone()
        if seg == segs[0]: ref[interp] = res
        same = bool(torch.equal(torch.nan_to_num(res), torch.nan_to_num(ref[interp])))
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps): e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, interp, True)
        e.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(json.dumps({'interpolation': str(interp), 'planes': P, 'size': sz, 'option': seg, 'segment': used, 'ms': round(dt * 1e3, 3),
                          'us_per_plane': round(dt / P * 1e6, 1), 'bits_equal_first': same}), flush=True)
e.close()
