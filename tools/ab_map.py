#!/usr/bin/env python3
"""
A/B timing of two builds of libplanetmapper_hip.so on the GENERAL kernels in one process:
the x/y map through k_map (PM_OPT_GENERAL_KERNEL) on a 1 deg grid, and the 4096^2 headline
frame through k_disc<FLAGS>. usage: python tools/ab_map.py libA.so libB.so
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planetmapper_amd import _lib, engine as eng_mod
from planetmapper_amd.scenarios import load_scenario

dev = torch.device('cuda', 0)
g = load_scenario('jupiter_hst_2005')
sz = 4096
x0 = (sz - 1) / 2
lons = np.arange(0.5, 360, 1.0)[::-1]
lon_h, lat_h = np.meshgrid(lons, np.arange(-89.5, 90, 1.0))
n0, n1 = lon_h.shape
lon_d, lat_d = torch.from_numpy(np.ascontiguousarray(lon_h)).to(dev), torch.from_numpy(np.ascontiguousarray(lat_h)).to(dev)
xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}
engines = []
for path in sys.argv[1:3]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    e = eng_mod.Engine(0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_geometry(g)
    e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    e.set_option(1, 1)  # PM_OPT_GENERAL_KERNEL
    engines.append(e)
for label, fn in (('k_map (x/y map, 180x360)', lambda e: e.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)),
                  ('k_disc<1> (4096^2, 5 planes)', lambda e: e.backplanes_img_device(planes))):
    res = [[], []]
    for _ in range(50):
        for e in engines:
            fn(e)
    torch.cuda.synchronize()
    for rep in range(6):
        for i in ([0, 1] if rep % 2 == 0 else [1, 0]):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for a, b in evs:
                a.record(); fn(engines[i]); b.record()
            torch.cuda.synchronize()
            res[i] += [a.elapsed_time(b) for a, b in evs[5:]]
    print(json.dumps({'kernel': label, 'A_ms': round(float(np.mean(res[0])), 5), 'B_ms': round(float(np.mean(res[1])), 5),
                      'B/A': round(float(np.mean(res[1]) / np.mean(res[0])), 4)}))
