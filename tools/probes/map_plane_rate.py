"""Throughput of the map-space planes (pm_backplanes_map, device-resident) on grids far finer than the 1 deg one of the
headline: python tools/probes/map_plane_rate.py  -> one JSON line per (degree interval, plane set)."""
import sys, time, json, ctypes
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine, plane_mask, PLANE_INDEX, NUM_PLANES
from planetmapper_amd import _lib
from planetmapper_amd.scenarios import load_scenario
from oracle import oracle  # plane names only

for scen in ('jupiter_hst_2005', 'saturn_earth_2005'):
    g = load_scenario(scen)
    e = Engine(0); e.set_geometry(g); e.set_disc(511.3, 510.2, 400.0, 0.3, 1024, 1024, True)
    sets = {'5 planes': ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION'], 'x/y map': ['PIXEL-X', 'PIXEL-Y'],
            '26 planes': list(oracle.PLANE_NAMES)}
    for deg in (1.0, 0.1, 0.05):
        lon = np.arange(deg / 2, 360, deg); lat = np.arange(-90 + deg / 2, 90, deg)
        if g.west_positive: lon = lon[::-1]
        lon_g, lat_g = np.meshgrid(lon, lat)
        n0, n1 = lon_g.shape
        lon_d = torch.from_numpy(np.ascontiguousarray(lon_g)).cuda(); lat_d = torch.from_numpy(np.ascontiguousarray(lat_g)).cuda()
        for label, names in sets.items():
            if deg < 0.1 and label == '26 planes': continue
            bufs = {n: torch.empty((n0, n1), dtype=torch.float64, device='cuda') for n in names}
            ptrs = (ctypes.c_void_p * NUM_PLANES)()
            for n, a in bufs.items(): ptrs[PLANE_INDEX[n]] = a.data_ptr()
            mask = plane_mask(names)
            call = lambda: e._lib.pm_backplanes_map(e._ctx, mask, ctypes.c_void_p(lon_d.data_ptr()), ctypes.c_void_p(lat_d.data_ptr()), n0, n1, 0.0, ptrs, _lib.PM_MEM_DEVICE)
            rc = call(); assert rc == 0, rc
            for _ in range(20): call()
            e.synchronize()
            reps = 200 if deg >= 1 else 30
            t0 = time.perf_counter()
            for _ in range(reps): call()
            e.synchronize(); dt = (time.perf_counter() - t0) / reps
            nbytes = n0 * n1 * (16 + 8 * len(names))
            print(json.dumps({'scenario': scen, 'deg': deg, 'cells': n0 * n1, 'planes': label, 'ms': round(dt * 1e3, 4), 'Gcell_s': round(n0 * n1 / dt / 1e9, 2),
                              'GBps': round(nbytes / dt / 1e9, 1), 'frac_of_8TBps': round(nbytes / dt / 8e12, 3)}), flush=True)
            del bufs
    e.close()
