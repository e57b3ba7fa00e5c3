"""Throughput of the array-valued coordinate transforms (pm_transform, device-resident, 16 M points) and of pm_radec_query:
python tools/probes/transform_rate.py -> one JSON line per transform."""
import sys, time, json, ctypes
sys.path[:0] = ['/root/repo']
import numpy as np, torch
from planetmapper_amd.engine import Engine
from planetmapper_amd import _lib
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005')
sz = 4096
e = Engine(0); e.set_geometry(g); x0 = (sz - 1) / 2; e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
n = sz * sz
yy, xx = torch.meshgrid(torch.arange(sz, dtype=torch.float64, device='cuda'), torch.arange(sz, dtype=torch.float64, device='cuda'), indexing='ij')
xy = (xx.reshape(-1).contiguous(), yy.reshape(-1).contiguous())
oa = torch.empty(n, dtype=torch.float64, device='cuda'); ob = torch.empty_like(oa)
vp = ctypes.c_void_p
def tf(src, dst, a, b, flags=0):
    rc = e._lib.pm_transform(e._ctx, _lib.COORDS[src], _lib.COORDS[dst], n, vp(a.data_ptr()), vp(b.data_ptr()), 0.0, flags, vp(oa.data_ptr()), vp(ob.data_ptr()), _lib.PM_MEM_DEVICE)
    assert rc == 0, rc
coords = {'xy': xy}
for dst in ('radec', 'angular', 'km', 'lonlat'):
    tf('xy', dst, *xy); e.synchronize()
    a, b = oa.clone(), ob.clone()
    if dst == 'lonlat':  # off-disc points are NaN: give them coordinates
        a = torch.where(torch.isfinite(a), a, torch.full_like(a, 123.4)); b = torch.where(torch.isfinite(b), b, torch.full_like(b, -12.3))
    coords[dst] = (a, b)
for src in coords:
    for dst in coords:
        if src == dst: continue
        a, b = coords[src]
        for _ in range(3): tf(src, dst, a, b)
        e.synchronize(); reps = 10
        t0 = time.perf_counter()
        for _ in range(reps): tf(src, dst, a, b)
        e.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(json.dumps({'transform': f'{src}2{dst}', 'points': n, 'ms': round(dt * 1e3, 3), 'Gpoint_s': round(n / dt / 1e9, 2), 'GBps': round(n * 32 / dt / 1e9, 1)}), flush=True)
out8 = torch.empty((8, n), dtype=torch.float64, device='cuda')
a, b = coords['radec']
call = lambda: e._lib.pm_radec_query(e._ctx, n, vp(a.data_ptr()), vp(b.data_ptr()), 0.0, 1, vp(out8.data_ptr()), _lib.PM_MEM_DEVICE)
assert call() == 0
e.synchronize(); t0 = time.perf_counter()
for _ in range(5): call()
e.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps({'transform': 'radec_query (8 outputs)', 'points': n, 'ms': round(dt * 1e3, 3), 'Gpoint_s': round(n / dt / 1e9, 2), 'GBps': round(n * 80 / dt / 1e9, 1)}), flush=True)
e.close()
