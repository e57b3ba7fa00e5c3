"""Bytes of a plane the bilinear reprojection of BASELINE config 5 touches, by block size (CPU arithmetic on the oracle's x / y map): python tools/probes/reproject_footprint.py [size] [deg]"""
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np
from oracle import oracle
from planetmapper_amd.scenarios import load_scenario

g = load_scenario('jupiter_hst_2005')
sz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
x0 = (sz - 1) / 2
d = oracle.make_disc(x0, x0, 0.9 * x0, 0.0, sz, sz)
lon, lat = oracle.rectangular_grid(g, deg)
xm, ym = oracle.xy_map(g, d, lon, lat)
ok = np.isfinite(xm)
ix = np.clip(np.floor(xm[ok]), 0, sz - 2).astype(np.int64)
iy = np.clip(np.floor(ym[ok]), 0, sz - 2).astype(np.int64)
addr = np.concatenate([((iy + dy) * sz + (ix + dx)) * 8 for dy in (0, 1) for dx in (0, 1)])
out = {'plane': f'{sz}x{sz} f64', 'map_deg': deg, 'cells': int(ok.size), 'visible_cells': int(ok.sum()), 'bytes_touched_per_plane': {}}
for b in (8, 16, 32, 64, 128, 256):
    out['bytes_touched_per_plane'][str(b)] = int(np.unique(addr // b).size * b)
print(json.dumps(out))
