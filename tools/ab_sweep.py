#!/usr/bin/env python3
"""
A/B timing of two builds of libplanetmapper_hip.so in ONE process on ONE GPU (clock and
device variance between gpurun sessions is ~10 %, larger than most kernel tweaks).
usage: python tools/ab_sweep.py libA.so libB.so [planes] ; alternating blocks of back-to-back launches.
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
if os.environ.get('TRIAXIAL'):  # a triaxial body: the TRI variant of the frame kernel (TRIAXIAL = b / a, WDOT_SCALE: slower spin)
    g.radii[1] = g.radii[0] * float(os.environ['TRIAXIAL'])
    g.wdot *= float(os.environ.get('WDOT_SCALE', '1'))
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}
engines = []
for path, envs in zip(sys.argv[1:3], (os.environ.get('AB_ENV_A', ''), os.environ.get('AB_ENV_B', ''))):
    os.environ['PM_DEBUG_ENV'] = '1'  # (the library reads its A/B knobs only behind this gate)
    for kv in filter(None, envs.split(',')):  # e.g. AB_ENV_B=PM_LT_MODE=2 (read by the library at pm_create)
        os.environ[kv.split('=')[0]] = kv.split('=', 1)[1]
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    e = eng_mod.Engine(0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_geometry(g)
    engines.append(e)
x0 = (sz - 1) / 2
BLOCK, ROUNDS = 60, 6
for r0 in (0.9 * x0, 1e5):
    res = [[], []]
    for e in engines:
        e.set_disc(x0, x0, r0, 0.0, sz, sz, True)
    # run-in: the shader clock needs ~0.1 s of sustained load to settle (and drops again within
    # milliseconds of idling), so launches are issued back to back without host syncs
    for _ in range(200):
        for e in engines:
            e.backplanes_img_device(planes)
    torch.cuda.synchronize()
    for rep in range(ROUNDS):
        for i, e in enumerate(engines if rep % 2 == 0 else engines[::-1]):
            i = i if rep % 2 == 0 else 1 - i
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(BLOCK)]
            for a, b in evs:
                a.record(); e.backplanes_img_device(planes); b.record()
            torch.cuda.synchronize()
            res[i] += [a.elapsed_time(b) for a, b in evs[10:]]  # the first launches after a switch settle
    print(json.dumps({'r0': r0, 'A_ms': round(float(np.mean(res[0])), 4), 'B_ms': round(float(np.mean(res[1])), 4),
                      'B/A': round(float(np.mean(res[1]) / np.mean(res[0])), 4)}))
