#!/usr/bin/env python3
"""
Bench-step variants on one GPU, same process, alternating blocks (wall clock around device syncs):
  serial   frame kernel, then pm_mapped_data, one stream (bench.py)
  overlap  pm_mapped_data of each frame on a SECOND engine context / stream with no event waits
           between the streams inside the loop (the two launches of a step are independent)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
sz = 4096
dev = torch.device('cuda', 0)
g = load_scenario('jupiter_hst_2005')
x0 = (sz - 1) / 2
main = torch.cuda.current_stream()
eng = Engine(0)
eng.set_stream(main.cuda_stream)
eng2 = Engine(0)  # its own non-blocking stream
for e in (eng, eng2):
    e.set_geometry(g)
    e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in HEADLINE}
lons = np.arange(0.5, 360, 1.0)[::-1] if g.west_positive else np.arange(0.5, 360, 1.0)
lon_h, lat_h = np.meshgrid(lons, np.arange(-89.5, 90, 1.0))
n0, n1 = lon_h.shape
lon_d, lat_d = torch.from_numpy(np.ascontiguousarray(lon_h)).to(dev), torch.from_numpy(np.ascontiguousarray(lat_h)).to(dev)
xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
data = torch.rand((sz, sz), dtype=torch.float64, device=dev)
out = torch.empty((1, n0, n1), dtype=torch.float64, device=dev)
torch.cuda.synchronize()


def step_serial():
    eng.backplanes_img_device(planes)
    eng.mapped_data_device(data, np.float64, 1, lon_d, lat_d, n0, n1, xm, ym, out)


def step_overlap():
    eng2.mapped_data_device(data, np.float64, 1, lon_d, lat_d, n0, n1, xm, ym, out)
    eng.backplanes_img_device(planes)


def sync():
    torch.cuda.synchronize()
    eng2.synchronize()


res = {'serial': [], 'overlap': []}
for fn in (step_serial, step_overlap):
    for _ in range(400):
        fn()
sync()
for rep in range(6):
    order = (('serial', step_serial), ('overlap', step_overlap))
    for name, fn in order if rep % 2 == 0 else order[::-1]:
        for _ in range(100):
            fn()
        sync()
        t = time.perf_counter()
        for _ in range(300):
            fn()
        sync()
        res[name].append((time.perf_counter() - t) / 300 * 1e3)
print(json.dumps({k: round(float(np.mean(v)), 4) for k, v in res.items()} | {'runs': res}))
