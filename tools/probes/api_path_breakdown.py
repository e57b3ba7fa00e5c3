"""Where the drop-in surface's time goes: Engine.backplanes_img for the lon / lat family, the illumination family and the
five planes at once (4096^2, fresh numpy arrays), and the five getters of BodyXY - python tools/probes/api_path_breakdown.py [trace]"""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np
from planetmapper_amd import BodyXY, _lib
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005'); sz = 4096; x0 = (sz - 1) / 2
e = Engine(0); e.set_geometry(g); e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
sets = {'lonlat': ['LON-GRAPHIC', 'LAT-GRAPHIC'], 'illum': ['PHASE', 'INCIDENCE', 'EMISSION'], 'five': ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION'], 'one': ['EMISSION']}
def med(fn, reps=9):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t); del r
    return round(float(np.median(ts)) * 1e3, 2), round(min(ts) * 1e3, 2)
for k, names in sets.items():
    e.backplanes_img(names)
    print(json.dumps({'engine_call': k, 'planes': len(names), 'ms_median_min': med(lambda: e.backplanes_img(names))}), flush=True)
def both():
    a = e.backplanes_img(sets['lonlat']); b = e.backplanes_img(sets['illum']); return a, b
print(json.dumps({'engine_call': 'lonlat then illum', 'ms_median_min': med(both)}), flush=True)
if len(sys.argv) > 1:
    e.set_option(_lib.PM_OPT_TRACE, 1)
    for k in ('lonlat', 'illum', 'five'):
        print('--- trace', k, file=sys.stderr, flush=True); e.backplanes_img(sets[k])
    e.set_option(_lib.PM_OPT_TRACE, 0)
import ctypes
from planetmapper_amd.engine import PLANE_INDEX, plane_mask
def into(arrs):
    ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
    for n, a in arrs.items(): ptrs[PLANE_INDEX[n]] = a.ctypes.data
    return lambda: e._check(e._lib.pm_backplanes_img(e._ctx, plane_mask(list(arrs)), 0.0, ptrs, _lib.PM_MEM_HOST))
reused = {n: np.empty((sz, sz)) for n in sets['five']}
for a in reused.values(): a[:] = 0
print(json.dumps({'engine_call': 'five, into the SAME pageable arrays (pages already there)', 'ms_median_min': med(into(reused))}), flush=True)
pinned = {n: e.pinned_empty((sz, sz)) for n in sets['five']}
print(json.dumps({'engine_call': 'five, into pinned arrays', 'ms_median_min': med(into(pinned))}), flush=True)
body = BodyXY('Jupiter', geometry=g, engine=e, nx=sz, ny=sz)
getters = ('get_lon_img', 'get_lat_img', 'get_phase_angle_img', 'get_incidence_angle_img', 'get_emission_angle_img')
def five():
    return [getattr(body, n)() for n in getters]
ts = []
for _ in range(9):
    body.set_disc_params(x0, x0, 0.9 * x0, 0.0)  # (clears the cache - and frees the previous arrays - outside the clock)
    t = time.perf_counter(); r = five(); ts.append(time.perf_counter() - t); del r
print(json.dumps({'bodyxy': 'five getters, cold cache', 'ms_median_min': (round(float(np.median(ts)) * 1e3, 2), round(min(ts) * 1e3, 2))}), flush=True)
e.close()
