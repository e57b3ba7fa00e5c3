"""
A/B of the light-time paths of k_disc_sph on the headline frame (test tooling: uses the oracle as the checker).
PM_LT_MODE=0 closed form (default), 1 the reference's sequence, 2 Newton step on the seed.
Prints per plane: median / p99 / max of |HIP - oracle64|, |HIP - truth|, |oracle64 - truth| over 64 bands of 4 rows.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle  # noqa: E402
from planetmapper_amd.scenarios import load_scenario  # noqa: E402
from planetmapper_amd.engine import Engine  # noqa: E402

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']


def stats(e):
    return {'median': float(np.median(e)), 'p99': float(np.quantile(e, 0.99)), 'max': float(e.max()),
            'inside_1e-9': float((e <= 1e-9).mean())}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'jupiter_hst_2005'
    g = load_scenario(which)
    sz = 4096
    x0 = y0 = (sz - 1) / 2
    r0 = 0.9 * x0
    d = oracle.make_disc(x0, y0, r0, 0.0, sz, sz)
    bands = [(int(b), 4) for b in np.linspace(200, sz - 208, 64)]
    tru = [oracle.backplanes_img_rows_quad(g, d, HEADLINE, a, n) for a, n in bands]
    full = oracle.backplanes_img(g, d, HEADLINE)
    o64 = [{k: full[k][a:a + n] for k in HEADLINE} for a, n in bands]
    out = {}
    KEYS = os.environ.get('LT_PLANES', ' '.join(HEADLINE)).split()
    for mode in os.environ.get('LT_MODES', '0 1 2').split():
        os.environ['PM_DEBUG_ENV'] = '1'  # (the library reads its A/B knobs only behind this gate)
        os.environ['PM_LT_MODE'] = mode
        eng = Engine(0)
        try:
            eng.set_geometry(g)
            eng.set_disc(x0, y0, r0, 0.0, sz, sz, True)
            hip = eng.backplanes_img(HEADLINE)
        finally:
            eng.close()
        rep = {}
        for k in KEYS:
            ho, ht, ot = [], [], []
            for (a, n), t, o in zip(bands, tru, o64):
                m = np.isfinite(t[k])
                assert np.array_equal(np.isnan(hip[k][a:a + n]), np.isnan(t[k])), (mode, k)
                ho.append(np.abs(hip[k][a:a + n] - o[k])[m])
                ht.append(np.abs(hip[k][a:a + n] - t[k])[m])
                ot.append(np.abs(o[k] - t[k])[m])
            rep[k] = {'hip_vs_o64': stats(np.concatenate(ho)), 'hip_vs_truth': stats(np.concatenate(ht)),
                      'o64_vs_truth': stats(np.concatenate(ot))}
            print(mode, k, json.dumps(rep[k]), flush=True)
        out[mode] = rep
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/lt_mode_probe.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
