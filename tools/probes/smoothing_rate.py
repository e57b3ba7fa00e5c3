"""Smoothing-spline map_img (spline_smoothing > 0) of a resident cube, timed: python tools/probes/smoothing_rate.py
[planes [size [data [s_factor [degree]]]]] - data 'randn' (s = s_factor: far below the noise, the knot search runs to the
interpolating spline: the longest search there is) or 'signal' (structure + unit noise, s = s_factor * n_pixels: the case
the option exists for). PM_SM_DEBUG=1 PM_DEBUG_ENV=1 traces the rounds. One JSON line."""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
data = sys.argv[3] if len(sys.argv) > 3 else 'randn'
sf = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
skip = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # planes of the seeded cube to leave out at its start
g = load_scenario('jupiter_hst_2005')
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
gen = torch.Generator(device='cuda').manual_seed(5)
cube = torch.randn((P + skip, sz, sz), generator=gen, device='cuda', dtype=torch.float64)[skip:].contiguous()
s = sf
if data == 'signal':
    yy, xx = torch.meshgrid(torch.arange(sz, device='cuda', dtype=torch.float64), torch.arange(sz, device='cuda', dtype=torch.float64), indexing='ij')
    for p in range(P):
        cube[p] += torch.sin(xx / (7.0 + p % 13)) * torch.cos(yy / (11.0 + 0.5 * (p % 17))) * (3 + p % 5)
    s = sf * sz * sz
deg = 1.0
lon = np.arange(deg / 2, 360, deg)[::-1] if g.west_positive else np.arange(deg / 2, 360, deg)
lat = np.arange(-90 + deg / 2, 90, deg)
lon_g, lat_g = np.meshgrid(lon, lat); n0, n1 = lon_g.shape
lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda'); ym = torch.empty_like(xm)
e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
out = torch.empty((P, n0, n1), dtype=torch.float64, device='cuda')
times = []
for _ in range(reps + 1):
    t0 = time.perf_counter()
    e.map_cube_device(cube, np.float64, P, xm, ym, n0, n1, out, (k, k), True, spline_smoothing=s); e.synchronize()
    times.append(time.perf_counter() - t0)
    print(f'call {len(times)}: {times[-1] * 1e3:.1f} ms', file=sys.stderr, flush=True)
dt = min(times[1:] or times)
print(json.dumps({'workload': f'smoothing spline k={k}, {data}, s={s:g}', 'planes': P, 'size': sz, 'ms': round(dt * 1e3, 2), 'ms_per_plane': round(dt / P * 1e3, 3),
                  'finite': bool(torch.isfinite(out[torch.isfinite(out) | ~torch.isnan(out)]).all().item())}), flush=True)
e.close()
