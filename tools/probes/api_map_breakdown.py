"""The drop-in surface on FINE maps (degree_interval 0.1: 1800 x 3600 cells): Observation.get_mapped_data for the
interpolations, the x / y map getter and a backplane map, cold cache each time, against the engine's own calls -
python tools/probes/api_map_breakdown.py [degree_interval] [image size] [planes]"""
import sys, time, json
sys.path[:0] = ['/root/repo']
import numpy as np
from planetmapper_amd import Observation
from planetmapper_amd.scenarios import load_scenario

deg = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = load_scenario('jupiter_hst_2005'); x0 = (sz - 1) / 2
rng = np.random.default_rng(3)
data = rng.standard_normal((P, sz, sz))
obs = Observation(data=data, geometry=g)
obs.set_disc_params(x0, x0, 0.9 * x0, 0.0)
def cold(fn, reps=7):
    ts = []
    for _ in range(reps):
        obs.set_disc_params(x0, x0, 0.9 * x0, 0.0)  # clears the disc-dependent cache (x / y maps, mapped data)
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t); del r
    return round(float(np.median(ts)) * 1e3, 2), round(min(ts) * 1e3, 2)
print(json.dumps({'get_x_map (x and y maps to the host)': cold(lambda: obs.get_x_map(degree_interval=deg))}), flush=True)
for interp in ('nearest', 'linear', 'cubic', 'smooth'):
    print(json.dumps({'get_mapped_data': interp, 'deg': deg, 'planes': P, 'ms_median_min': cold(lambda: obs.get_mapped_data(interp, degree_interval=deg))}), flush=True)
def warm_xy(interp):
    obs.get_x_map(degree_interval=deg)
    t = time.perf_counter(); r = obs.map_img(data, interpolation=interp, degree_interval=deg); dt = time.perf_counter() - t; del r
    return round(dt * 1e3, 2)
for interp in ('linear', 'smooth'):
    print(json.dumps({'map_img with the x / y maps cached': interp, 'ms': [warm_xy(interp) for _ in range(5)]}), flush=True)
e = obs._bind()
lon, lat = obs._get_lonlat_map(degree_interval=deg)[:, :, 0].copy(), obs._get_lonlat_map(degree_interval=deg)[:, :, 1].copy()
def eng_xy():
    return e.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon, lat)
ts = []
for _ in range(7):
    t = time.perf_counter(); r = eng_xy(); ts.append(time.perf_counter() - t); del r
print(json.dumps({'engine.backplanes_map x / y': (round(float(np.median(ts)) * 1e3, 2), round(min(ts) * 1e3, 2))}), flush=True)
xy = eng_xy(); xm, ym = xy['PIXEL-X'], xy['PIXEL-Y']
for interp in ('linear', 'smooth'):
    ts = []
    for _ in range(7):
        t = time.perf_counter(); r = e.map_cube(data, xm, ym, interp, True); ts.append(time.perf_counter() - t); del r
    print(json.dumps({'engine.map_cube': interp, 'ms_median_min': (round(float(np.median(ts)) * 1e3, 2), round(min(ts) * 1e3, 2))}), flush=True)
t = time.perf_counter(); m = obs.get_backplane_map('EMISSION', degree_interval=deg); print(json.dumps({'get_backplane_map EMISSION (first: the whole family)': round((time.perf_counter() - t) * 1e3, 2)}))
