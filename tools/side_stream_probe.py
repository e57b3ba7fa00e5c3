#!/usr/bin/env python3
"""
Bench-step variants on one GPU: (a) frame kernel -> x/y map -> reprojection in one stream (bench.py),
(b) the x/y map of the NEXT frame on a second stream, concurrent with the frame kernel.
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
side = torch.cuda.Stream()
eng = Engine(0)
eng.set_stream(main.cuda_stream)
eng.set_geometry(g)
eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
eng2 = Engine(0)  # second context bound to the side stream (a context owns one stream)
eng2.set_stream(side.cuda_stream)
eng2.set_geometry(g)
eng2.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in HEADLINE}
lons = np.arange(0.5, 360, 1.0)[::-1] if g.west_positive else np.arange(0.5, 360, 1.0)
lon_h, lat_h = np.meshgrid(lons, np.arange(-89.5, 90, 1.0))
n0, n1 = lon_h.shape
lon_d, lat_d = torch.from_numpy(np.ascontiguousarray(lon_h)).to(dev), torch.from_numpy(np.ascontiguousarray(lat_h)).to(dev)
xm = [torch.empty((n0, n1), dtype=torch.float64, device=dev) for _ in range(2)]
ym = [torch.empty((n0, n1), dtype=torch.float64, device=dev) for _ in range(2)]
data = torch.rand((sz, sz), dtype=torch.float64, device=dev)
out = torch.empty((1, n0, n1), dtype=torch.float64, device=dev)


def step_serial():
    eng.backplanes_img_device(planes)
    eng.xy_map_device(lon_d, lat_d, n0, n1, xm[0], ym[0])
    eng.map_cube_device(data, np.float64, 1, xm[0], ym[0], n0, n1, out)


ev_map = [torch.cuda.Event() for _ in range(2)]
ev_used = [torch.cuda.Event() for _ in range(2)]
state = {'i': 0}


def step_overlap():
    k = state['i'] & 1
    state['i'] += 1
    # the map of this frame on the side stream, concurrent with this frame's backplanes
    side.wait_event(ev_used[k])  # buffer k free again (its reprojection two steps ago is done)
    with torch.cuda.stream(side):
        eng2.xy_map_device(lon_d, lat_d, n0, n1, xm[k], ym[k])
        ev_map[k].record(side)
    eng.backplanes_img_device(planes)
    main.wait_event(ev_map[k])
    eng.map_cube_device(data, np.float64, 1, xm[k], ym[k], n0, n1, out)
    ev_used[k].record(main)


for e in ev_used:
    e.record(main)
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
