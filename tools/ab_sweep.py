#!/usr/bin/env python3
"""
A/B timing of two builds of libplanetmapper_hip.so in ONE process on ONE GPU (clock and
device variance between gpurun sessions is ~10 %, larger than most kernel tweaks).
usage: python tools/ab_sweep.py libA.so libB.so [planes] ; interleaved, median of 30.
"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planetmapper_amd import _lib, engine as eng_mod
from planetmapper_amd.scenarios import load_scenario

names = sys.argv[3].split(',') if len(sys.argv) > 3 else ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
sz = 4096
dev = torch.device('cuda', 0)
g = load_scenario(os.environ.get('SCENARIO', 'jupiter_hst_2005'))
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}
engines = []
for path in sys.argv[1:3]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    e = eng_mod.Engine(0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_geometry(g)
    engines.append(e)
x0 = (sz - 1) / 2
for r0 in (0.9 * x0, 1e5):
    res = [[], []]
    for e in engines:
        e.set_disc(x0, x0, r0, 0.0, sz, sz, True)
        for _ in range(3):
            e.backplanes_img_device(planes)
    torch.cuda.synchronize()
    for rep in range(30):
        for i, e in enumerate(engines):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); e.backplanes_img_device(planes); b.record()
            torch.cuda.synchronize()
            res[i].append(a.elapsed_time(b))
    print(json.dumps({'r0': r0, 'A_ms': round(float(np.median(res[0])), 4), 'B_ms': round(float(np.median(res[1])), 4),
                      'B/A': round(float(np.median(res[1]) / np.median(res[0])), 4)}))
