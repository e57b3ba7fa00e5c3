"""Per-call cost of the image-plane entry point for frames of BASELINE configs 1-2 and smaller (outputs resident in HBM):
python tools/probes/small_frame_latency.py   -> one JSON line per (size, planes): enqueue rate, with and without a sync per call"""
import sys, time, json, ctypes
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine, plane_mask, PLANE_INDEX, NUM_PLANES
from planetmapper_amd import _lib
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005')
e = Engine(0); e.set_geometry(g)
sets = {'5 planes': ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION'],
        '8 planes': ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION', 'RING-RADIUS', 'RING-LON-GRAPHIC', 'RADIAL-VELOCITY'],
        '26 planes': None}
from oracle import oracle  # names only
sets['26 planes'] = list(oracle.PLANE_NAMES)
for sz in (128, 256, 512, 1024, 2048):
    e.set_disc(sz / 2 + 0.3, sz / 2 - 0.2, sz * 0.4, 0.3, sz, sz, True)
    for label, names in sets.items():
        bufs = {n: torch.empty((sz, sz), dtype=torch.float64, device='cuda') for n in names}
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in bufs.items(): ptrs[PLANE_INDEX[n]] = a.data_ptr()
        mask = plane_mask(names)
        call = lambda: e._lib.pm_backplanes_img(e._ctx, mask, 0.0, ptrs, _lib.PM_MEM_DEVICE)
        for _ in range(50): call()
        e.synchronize()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n): call()
        e.synchronize(); t1 = time.perf_counter()
        for _ in range(n): call(); e.synchronize()
        t2 = time.perf_counter()
        print(json.dumps({'size': sz, 'planes': label, 'us_per_call_enqueued': round((t1 - t0) / n * 1e6, 2),
                          'us_per_call_synced': round((t2 - t1) / n * 1e6, 2),
                          'GBps_enqueued': round(len(names) * sz * sz * 8 / ((t1 - t0) / n) / 1e9, 1)}), flush=True)
e.close()
