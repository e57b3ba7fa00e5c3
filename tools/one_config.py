#!/usr/bin/env python3
"""Launch pm_backplanes_img a few times for one disc radius (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planetmapper_amd.engine import Engine
from planetmapper_amd.scenarios import load_scenario
r0 = float(sys.argv[1]); names = sys.argv[2].split(',')
sz = 4096; dev = torch.device('cuda', 0)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_geometry(load_scenario(os.environ.get('SCENARIO', 'jupiter_hst_2005')))
planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}
x0 = (sz - 1) / 2
eng.set_disc(x0, x0, r0, 0.0, sz, sz, True)
for _ in range(5):
    eng.backplanes_img_device(planes)
torch.cuda.synchronize()
