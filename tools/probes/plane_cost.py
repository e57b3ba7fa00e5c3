#!/usr/bin/env python3
"""What each group of image planes costs on top of the headline set (4096^2, Jupiter / HST): same-process timing."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def main():
    import torch

    from planetmapper_amd._lib import PLANE_NAMES
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    sz = 4096
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_geometry(load_scenario('jupiter_hst_2005'))
    eng.set_disc((sz - 1) / 2, (sz - 1) / 2, 0.9 * (sz - 1) / 2, 0.0, sz, sz, True)
    bufs = {n: torch.empty((sz, sz), dtype=torch.float64, device='cuda') for n in PLANE_NAMES}
    head = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
    sets = {
        'headline': head, '+azimuth': head + ['AZIMUTH'], '+centric': head + ['LON-CENTRIC', 'LAT-CENTRIC'],
        '+lst': head + ['LOCAL-SOLAR-TIME'], '+state': head + ['DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER'],
        '+distance only': head + ['DISTANCE'], '+ring': head + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE'],
        'disc15': [n for n in PLANE_NAMES if n not in ('RA', 'DEC', 'PIXEL-X', 'PIXEL-Y', 'KM-X', 'KM-Y', 'ANGULAR-X', 'ANGULAR-Y',
                                                       'LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC')],
        'sky8': ['RA', 'DEC', 'PIXEL-X', 'PIXEL-Y', 'KM-X', 'KM-Y', 'ANGULAR-X', 'ANGULAR-Y'],
        'limb3': ['LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC'], 'all26': list(PLANE_NAMES),
    }  # fmt: skip
    for _ in range(300):
        eng.backplanes_img_device({n: bufs[n] for n in head})
    for name, names in sets.items():
        outs = {n: bufs[n] for n in names}
        ts = []
        for rnd in range(5):
            for _ in range(10):
                eng.backplanes_img_device(outs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                eng.backplanes_img_device(outs)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 40)
        t = float(np.median(ts))
        print(json.dumps({'set': name, 'planes': len(names), 'ms': round(t, 4), 'GBps': round(len(names) * sz * sz * 8 / t / 1e6, 1)}), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
