"""
The headline frame (4096^2, 5 planes, device-resident) through each image kernel: the spheroid fast path, its
triaxial variant (radii (a, 0.97 a, c)) and the general kernel (PM_OPT_GENERAL_KERNEL); ms per launch from events
after a clock run-in, median of `--reps` blocks of 20 launches, interleaved so that clock drift hits all alike.
PM_GENERAL_LEGACY=1 in the environment times the J2000 kernel of rounds 1-3 as "general".
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--reps', type=int, default=9)
    args = ap.parse_args()
    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    sz = args.size
    g = load_scenario('jupiter_hst_2005')
    gt = g.copy()
    gt.radii[1] = 0.97 * gt.radii[0]
    names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
    planes = {n: torch.empty((sz, sz), dtype=torch.float64, device='cuda') for n in names}
    cases = {}
    gm = g.copy()  # a real moon: Io's shape and spin at Jupiter's distance (the triaxial variant's closed-form light time)
    gm.radii[0], gm.radii[1], gm.radii[2] = 1829.4, 1819.4, 1815.7
    gm.wdot = 4.11e-5
    gm.diameter_arcsec = g.diameter_arcsec * 1829.4 / g.radii[0]
    for label, geom, general in (('spheroid', g, 0), ('triaxial', gt, 0), ('moon', gm, 0), ('general', g, 1), ('general_triaxial', gt, 1)):
        eng = Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_geometry(geom)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
        eng.set_option(_lib.PM_OPT_GENERAL_KERNEL, general)
        cases[label] = eng
    for _ in range(300):
        cases['spheroid'].backplanes_img_device(planes)
    ts = {k: [] for k in cases}
    for rep in range(args.reps):
        for label, eng in cases.items():
            for _ in range(5):
                eng.backplanes_img_device(planes)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                eng.backplanes_img_device(planes)
            b.record()
            torch.cuda.synchronize()
            ts[label].append(a.elapsed_time(b) / 20)
    print(json.dumps({'size': sz, 'legacy_general': bool(os.environ.get('PM_GENERAL_LEGACY')),
                      'ms': {k: {'median': round(float(np.median(v)), 4), 'min': round(min(v), 4)} for k, v in ts.items()},
                      'kernel': {k: e.get_option(_lib.PM_OPT_LAST_DISC_KERNEL) for k, e in cases.items()}}), flush=True)


if __name__ == '__main__':
    main()
