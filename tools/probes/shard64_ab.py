"""Rank 0 of 8 of the sharded host-fed cube, alone on this GPU (64 planes of 1024^2 f64 from pinned host memory onto the 1 deg
map, PM_MEM_HOST_CUBE), for 16 and 2 copy threads: python tools/probes/shard64_ab.py [planes] -> JSON lines. For A/B runs of
pool / pipeline knobs (PM_DEBUG_ENV=1 PM_POOL_SPIN_US=0 ...) in separate processes on the same box."""
import sys, time, json, os
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd import _lib
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sz = 1024
g = load_scenario('jupiter_hst_2005')
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
lon = np.arange(0.5, 360, 1.0)[::-1] if g.west_positive else np.arange(0.5, 360, 1.0)
lon_g, lat_g = np.meshgrid(lon, np.arange(-89.5, 90, 1.0)); n0, n1 = lon_g.shape
lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda'); ym = torch.empty_like(xm)
e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
cube = e.pinned_empty((P, sz, sz)); cube[...] = np.random.default_rng(1).standard_normal((P, sz, sz))
out = torch.empty((P, n0, n1), dtype=torch.float64, device='cuda')
for threads in (16, 2):
    e.set_option(_lib.PM_OPT_HOST_COPY_THREADS, threads); e.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)
    for route in (-1, 3):
        e.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, route)
        for _ in range(8):
            e.map_cube_host_to_device(cube, xm, ym, n0, n1, out); e.synchronize()
        ts = []
        for _ in range(15):
            t = time.perf_counter(); e.map_cube_host_to_device(cube, xm, ym, n0, n1, out); e.synchronize(); ts.append(time.perf_counter() - t)
        e.set_option(_lib.PM_OPT_TRACE, 4)
        e.map_cube_host_to_device(cube, xm, ym, n0, n1, out); e.synchronize()
        st = e.last_stages_ms()
        e.set_option(_lib.PM_OPT_TRACE, 0)
        print(json.dumps({'threads': threads, 'route_asked': route, 'route': e.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE), 'spin_us': os.environ.get('PM_POOL_SPIN_US', 'default'),
                          'ms_median': round(float(np.median(ts)) * 1e3, 3), 'ms_min': round(min(ts) * 1e3, 3),
                          'stages': {k: round(v, 3) for k, v in st.items() if v and not k.startswith(('exchange', 'agree', 'sharded'))}}), flush=True)
e.close()
