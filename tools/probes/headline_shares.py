"""Share of on-disc pixels of the 4096^2 headline frame inside the flat 1e-9 deg (HIP vs oracle) for the scenario's geometry and for the same epoch with the observer taken from the golden header (rounds 1-5): python tools/probes/headline_shares.py"""
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np
from oracle import oracle
from planetmapper_amd import ephem
from planetmapper_amd.engine import Engine
from planetmapper_amd.geometry import GeometryBuilder
from planetmapper_amd.scenarios import load_scenario, scenario_info

d = scenario_info('jupiter_hst_2005')
eph = ephem.Ephemeris.from_json(d['ephemeris']); rot = ephem.RotationModel.from_json(d['pck'])
gb = GeometryBuilder(eph, rot, 599); h = d['header']
g_new = load_scenario('jupiter_hst_2005')
g_old = gb.build(d['et'], observer_velocity=list(g_new.VO[:]), target_ra_dec_dist_lt=(h['PLANMAP TARGET RA'], h['PLANMAP TARGET DEC'], h['PLANMAP DISTANCE'], h['PLANMAP LIGHT-TIME']))
names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
sz = 4096; x0 = (sz - 1) / 2
e = Engine(0); oracle.set_num_threads(16)
for label, g in (('from the TLE ephemeris', g_new), ('from the header', g_old)):
    for r0 in (0.9 * x0, 0.9 * x0 + 0.37):
        e.set_geometry(g); e.set_disc(x0, x0, r0, 0.0, sz, sz, True)
        out = e.backplanes_img(names)
        ref = oracle.backplanes_img(g, oracle.make_disc(x0, x0, r0, 0.0, sz, sz), names)
        res = {}
        for n in names:
            fin = np.isfinite(ref[n]); dd = np.abs(out[n] - ref[n])[fin]
            if 'LON' in n: dd = np.minimum(dd, 360 - dd)
            res[n] = [round(float(np.mean(dd <= 1e-9)), 5), float(dd.max())]
        print(json.dumps({'observer': label, 'r0': r0, 'T0': list(g.T0[:]), 'share_within_1e-9_deg_and_max': res}), flush=True)
e.close()
