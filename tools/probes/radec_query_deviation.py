"""Deviation of pm_radec_query (B0 kernel and the J2000 one behind PM_OPT_GENERAL_KERNEL) from the oracle on 200 000 sky points around Jupiter and Saturn: masks, max / p99, share inside 1e-9 deg."""
import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
from planetmapper_amd.engine import Engine
from planetmapper_amd import _lib
from planetmapper_amd.scenarios import load_scenario
from oracle import oracle
e = Engine(0)
rng = np.random.default_rng(7)
for name in ('jupiter_hst_2005','saturn_earth_2005'):
    g = load_scenario(name)
    e.set_geometry(g); e.set_disc(2.5,3.1,3.9,0.0,7,10,True)
    t0 = np.array(g.T0[:]); ra0 = np.rad2deg(np.arctan2(t0[1], t0[0])) % 360.0; dec0 = np.rad2deg(np.arcsin(t0[2]/np.linalg.norm(t0)))
    span = 1.2*g.diameter_arcsec/3600.0
    ra = ra0 + rng.uniform(-span, span, 200000)/np.cos(np.deg2rad(dec0)); dec = dec0 + rng.uniform(-span, span, 200000)
    ref = oracle.radec_query(g, ra, dec, alt=0.0, ring_only_visible=True).T
    for gen in (0,1):
        e.set_option(_lib.PM_OPT_GENERAL_KERNEL, gen)
        got = e.radec_query(ra, dec, alt=0.0, ring_only_visible=True)
        print(name, 'general' if gen else 'b0', 'mask equal', np.array_equal(np.isnan(got), np.isnan(ref)), 'mismatches', int((np.isnan(got)!=np.isnan(ref)).sum()))
        fin = np.isfinite(ref) & np.isfinite(got)
        d = np.abs(got-ref); d[[0,3,5]] = np.minimum(d[[0,3,5]], 360-d[[0,3,5]])
        for k,nm in enumerate(['lon','lat','ring r','ring lon','ring d','limb lon','limb lat','limb d']):
            v = d[k][fin[k]]
            print('   %-9s max %.3e  p99 %.3e  share<1e-9 %.4f' % (nm, v.max(), np.quantile(v,0.99), (v<1e-9).mean()))
e.close()
