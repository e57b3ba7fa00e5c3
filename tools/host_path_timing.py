#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-buffer path (numpy in / numpy out) on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planetmapper_amd import BodyXY
names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
b = BodyXY('jupiter', scenario='jupiter_hst_2005', sz=4096)
b.prefetch_backplane_imgs(names)  # warm up (allocations)
for _ in range(3):
    b.set_x0(b.get_x0())  # invalidate cache
    t = time.perf_counter(); b.prefetch_backplane_imgs(names); dt = time.perf_counter() - t
    print(f'host-buffer path 4096^2 x5 planes: {dt*1e3:.1f} ms = {4096*4096/dt/1e6:.0f} Mpix/s (D2H {5*134.2/dt/1e3:.1f} GB/s)')
