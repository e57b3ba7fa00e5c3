#!/usr/bin/env python3
"""Time pm_backplanes_img for disc sizes from all-off-disc to all-on-disc (GPU box)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario

names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
sz = 4096
dev = torch.device('cuda', 0)
eng = Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
g = load_scenario(os.environ.get('SCENARIO', 'jupiter_hst_2005'))
if os.environ.get('TRIAXIAL'):
    g.radii[1] = g.radii[0] * 0.97  # a triaxial body: the TRI variant of the fast kernel
eng.set_geometry(g)
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}
x0 = (sz - 1) / 2
# The shader clock follows the load within milliseconds (and needs ~0.1 s of sustained work to
# settle after an idle gap), so every configuration gets its own untimed run-in.
PREHEAT = int(os.environ.get('PREHEAT', '400'))
for r0 in (1.0, 0.45 * x0, 0.9 * x0, 1e5):
    eng.set_disc(x0, x0, r0, 0.0, sz, sz, True)
    for _ in range(PREHEAT):
        eng.backplanes_img_device(planes)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in evs:
        a.record(); eng.backplanes_img_device(planes); b.record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in evs])
    frac = float(torch.isfinite(planes[names[0]]).double().mean())
    print(json.dumps({'r0': r0, 'on_disc': round(frac, 4), 'ms': round(float(ms), 4),
                      'GB/s': round(sz * sz * 8 * len(names) / ms / 1e6, 1),
                      'ns_per_on_disc_px': round(ms * 1e6 / max(1, frac * sz * sz), 4)}))
