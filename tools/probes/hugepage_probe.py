"""Do transparent huge pages take the page-fault cost out of results written into FRESH numpy arrays? Engine.backplanes_img-style
call (five planes of 4096^2) into new arrays as they are, into new arrays with madvise(MADV_HUGEPAGE) on their pages first, and
into arrays whose pages exist: python tools/probes/hugepage_probe.py"""
import ctypes, json, sys, time
sys.path[:0] = ['/root/repo']
import numpy as np
from planetmapper_amd import _lib
from planetmapper_amd.engine import Engine, PLANE_INDEX, plane_mask
from planetmapper_amd.scenarios import load_scenario

print(json.dumps({'thp_enabled': open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(),
                  'thp_defrag': open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip()}))
libc = ctypes.CDLL('libc.so.6', use_errno=True)
MADV_HUGEPAGE = 14
g = load_scenario('jupiter_hst_2005'); sz = 4096; x0 = (sz - 1) / 2
e = Engine(0); e.set_geometry(g); e.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
names = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
def call(arrs):
    ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
    for n, a in arrs.items(): ptrs[PLANE_INDEX[n]] = a.ctypes.data
    e._check(e._lib.pm_backplanes_img(e._ctx, plane_mask(names), 0.0, ptrs, _lib.PM_MEM_HOST))
def fresh(huge):
    arrs = {n: np.empty((sz, sz)) for n in names}
    if huge:
        for a in arrs.values():
            lo = (a.ctypes.data + 4095) & ~4095; hi = (a.ctypes.data + a.nbytes) & ~4095
            rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), MADV_HUGEPAGE)
            assert rc == 0, ctypes.get_errno()
    return arrs
call(fresh(False))
for label, mk in (('fresh', lambda: fresh(False)), ('fresh + MADV_HUGEPAGE', lambda: fresh(True)), ('fresh', lambda: fresh(False)), ('fresh + MADV_HUGEPAGE', lambda: fresh(True))):
    ts = []
    for _ in range(7):
        arrs = mk()
        t = time.perf_counter(); call(arrs); ts.append(time.perf_counter() - t)
        del arrs
    print(json.dumps({'into': label, 'ms_median': round(float(np.median(ts)) * 1e3, 2), 'ms_min': round(min(ts) * 1e3, 2)}), flush=True)
reused = fresh(False); call(reused)
ts = []
for _ in range(7):
    t = time.perf_counter(); call(reused); ts.append(time.perf_counter() - t)
print(json.dumps({'into': 'the same arrays again', 'ms_median': round(float(np.median(ts)) * 1e3, 2), 'ms_min': round(min(ts) * 1e3, 2)}))
e.close()
