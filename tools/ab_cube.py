#!/usr/bin/env python3
"""
A/B timing of two builds of libplanetmapper_hip.so on the cube reprojection (BASELINE config 5:
P x 1024 x 1024 f64 cube -> 1 deg rectangular map, bilinear) in ONE process on ONE GPU.
usage: python tools/ab_cube.py libA.so libB.so [planes]; alternating blocks of back-to-back calls.
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planetmapper_amd import _lib, engine as eng_mod
from planetmapper_amd.scenarios import load_scenario



def rectangular_grid(west_positive, degree_interval=1.0):
    lons = np.arange(degree_interval / 2, 360, degree_interval)
    if west_positive:
        lons = lons[::-1]
    lats = np.arange(-90 + degree_interval / 2, 90, degree_interval)
    lon, lat = np.meshgrid(lons, lats)
    return np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat)


planes = int(sys.argv[3]) if len(sys.argv) > 3 else 512
sz = 1024
dev = torch.device('cuda', 0)
g = load_scenario('jupiter_hst_2005')
gen = torch.Generator(device=dev).manual_seed(5)
cube = torch.randn((planes, sz, sz), generator=gen, device=dev, dtype=torch.float64)
cube[torch.rand((planes, sz, sz), generator=gen, device=dev) < 1e-3] = float('nan')
lon_h, lat_h = rectangular_grid(bool(g.west_positive))
n0, n1 = lon_h.shape
lon_d, lat_d = torch.from_numpy(lon_h).to(dev), torch.from_numpy(lat_h).to(dev)
xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
outs = [torch.empty((planes, n0, n1), dtype=torch.float64, device=dev) for _ in range(2)]
engines = []
x0 = (sz - 1) / 2
for path in sys.argv[1:3]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    e = eng_mod.Engine(0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_geometry(g)
    e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    engines.append(e)
engines[0].xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
for interpolation in ('linear', 'nearest'):
    res = [[], []]
    for _ in range(100):
        for i, e in enumerate(engines):
            e.map_cube_device(cube, np.float64, planes, xm, ym, n0, n1, outs[i], interpolation)
    torch.cuda.synchronize()
    same = bool(torch.equal(torch.nan_to_num(outs[0], nan=-1.0), torch.nan_to_num(outs[1], nan=-1.0)))
    BLOCK, ROUNDS = 40, 6
    for rep in range(ROUNDS):
        order = [0, 1] if rep % 2 == 0 else [1, 0]
        for i in order:
            e = engines[i]
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(BLOCK)]
            for a, b in evs:
                a.record(); e.map_cube_device(cube, np.float64, planes, xm, ym, n0, n1, outs[i], interpolation); b.record()
            torch.cuda.synchronize()
            res[i] += [a.elapsed_time(b) for a, b in evs[5:]]
    print(json.dumps({'interpolation': interpolation, 'planes': planes, 'A_ms': round(float(np.mean(res[0])), 4),
                      'B_ms': round(float(np.mean(res[1])), 4), 'B/A': round(float(np.mean(res[1]) / np.mean(res[0])), 4),
                      'identical_output': same}))
