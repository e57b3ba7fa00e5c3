#!/usr/bin/env python3
"""Same-process A/B of the all-26-plane request: one launch (PM_OPT_FUSE_PLANES 1) vs one launch per group (0)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd._lib import PLANE_NAMES
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    sz = 4096
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for scen, r0, rot in (('jupiter_hst_2005', 0.9 * (sz - 1) / 2, 0.0), ('saturn_earth_2005', 800.0, float(np.deg2rad(20.0)))):
        eng.set_geometry(load_scenario(scen))
        eng.set_disc((sz - 1) / 2, (sz - 1) / 2, r0, rot, sz, sz, True)
        planes = {n: torch.empty((sz, sz), dtype=torch.float64, device='cuda') for n in PLANE_NAMES}
        for _ in range(300):
            eng.backplanes_img_device(planes)
        res = {}
        for rnd in range(6):
            for fuse in (1, 0):
                eng.set_option(_lib.PM_OPT_FUSE_PLANES, fuse)
                for _ in range(20):
                    eng.backplanes_img_device(planes)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50):
                    eng.backplanes_img_device(planes)
                b.record()
                torch.cuda.synchronize()
                res.setdefault(fuse, []).append(a.elapsed_time(b) / 50)
        alg = sz * sz * 8 * 26
        out = {'scenario': scen, 'ms_one_launch': round(float(np.median(res[1])), 4), 'ms_per_group': round(float(np.median(res[0])), 4),
               'GBps_one_launch': round(alg / np.median(res[1]) / 1e6, 1), 'frac_of_8TBps': round(alg / np.median(res[1]) / 1e6 / 8000, 3)}
        print(json.dumps(out), flush=True)
    eng.set_option(_lib.PM_OPT_FUSE_PLANES, 0)
    eng.close()


if __name__ == '__main__':
    main()
