#!/usr/bin/env python3
"""
Bench-step variants on one GPU (the final round-3 build: the frame kernel leaves a fifth of the VALU idle):
(a) bench.py's step - frame kernel, then pm_mapped_data (x/y map + reprojection, one launch) - in one stream;
(b) pm_mapped_data of the same step on a SECOND stream and context, concurrent with the frame kernel (the two
    calls share no data: the reprojected plane is the observation's image, not a backplane), joined at the end.
Prints ms per step of both; same process, alternating blocks.
"""
import json, os, sys
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
side = torch.cuda.Stream(priority=int(os.environ.get('SIDE_PRIORITY', '0')))
engs = []
for st in (main, side):
    e = Engine(0)
    e.set_stream(st.cuda_stream)
    e.set_geometry(g)
    e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    engs.append(e)
eng, eng2 = engs
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in HEADLINE}
lons = np.arange(0.5, 360, 1.0)[::-1] if g.west_positive else np.arange(0.5, 360, 1.0)
lon_h, lat_h = np.meshgrid(lons, np.arange(-89.5, 90, 1.0))
n0, n1 = lon_h.shape
lon_d, lat_d = torch.from_numpy(np.ascontiguousarray(lon_h)).to(dev), torch.from_numpy(np.ascontiguousarray(lat_h)).to(dev)
xm, ym = (torch.empty((n0, n1), dtype=torch.float64, device=dev) for _ in range(2))
data = torch.rand((1, sz, sz), dtype=torch.float64, device=dev)
out = torch.empty((1, n0, n1), dtype=torch.float64, device=dev)


def step_serial():
    eng.backplanes_img_device(planes)
    eng.mapped_data_device(data, np.float64, 1, lon_d, lat_d, n0, n1, xm, ym, out)


ev_start, ev_done = torch.cuda.Event(), torch.cuda.Event()


def step_overlap():
    ev_start.record(main)  # the side stream starts where the step starts (after the previous step's join)
    side.wait_event(ev_start)
    with torch.cuda.stream(side):
        eng2.mapped_data_device(data, np.float64, 1, lon_d, lat_d, n0, n1, xm, ym, out)
        ev_done.record(side)
    eng.backplanes_img_device(planes)
    main.wait_event(ev_done)  # join: the step ends when both have


res = {'serial': [], 'overlap': []}
for name, fn in (('serial', step_serial), ('overlap', step_overlap)):
    for _ in range(300):
        fn()
torch.cuda.synchronize()
for rep in range(6):
    for name, fn in (('serial', step_serial), ('overlap', step_overlap)) if rep % 2 == 0 else (('overlap', step_overlap), ('serial', step_serial)):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / 200)
print(json.dumps({k: round(float(np.mean(v)), 4) for k, v in res.items()}))
