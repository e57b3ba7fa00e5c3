#!/usr/bin/env python3
"""
PCIe-inclusive timing of the host-buffer (numpy in / numpy out) path on the GPU box: what a
drop-in caller of get_*_img() / get_mapped_data() sees. One JSON object per line.

  frame      4096^2 x 5 planes into fresh pageable numpy arrays / reused arrays / pinned arrays
  cube       512 x 1024^2 f64 host cube -> 1 deg map: pageable cube (pipelined copy), pinned cube
             copied (PM_OPT_ZERO_COPY=0), gathered in place (1), through the block table (2)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from planetmapper_amd import _lib  # noqa: E402
from planetmapper_amd.engine import Engine  # noqa: E402
from planetmapper_amd.scenarios import load_scenario  # noqa: E402

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']


def rectangular_grid(west_positive, step=1.0):
    lons = np.arange(step / 2, 360, step)
    if west_positive:
        lons = lons[::-1]
    lats = np.arange(-90 + step / 2, 90, step)
    lon, lat = np.meshgrid(lons, lats)
    return np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat)


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t)
        del r  # (releasing a previous result - munmap of its pages - is not part of the call)
    return min(ts), float(np.median(ts))


def frame(eng, g, sz, reps):
    import ctypes

    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    nbytes = len(HEADLINE) * sz * sz * 8
    eng.backplanes_img(HEADLINE)  # warm-up: scratch + staging allocations
    out = {}

    def fresh():
        return eng.backplanes_img(HEADLINE)  # np.empty per plane: first-touch page faults included

    lo, med = best(fresh, reps)
    out['a'] = fresh()
    yield {'case': 'frame -> fresh pageable numpy arrays', 'ms_best': lo * 1e3, 'ms_median': med * 1e3,
           'GBps': nbytes / lo / 1e9, 'Mpix_s': sz * sz / lo / 1e6}
    keep = out['a']
    ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
    from planetmapper_amd.engine import PLANE_INDEX, plane_mask

    def into(arrs):
        for n, a in arrs.items():
            ptrs[PLANE_INDEX[n]] = a.ctypes.data
        eng._check(eng._lib.pm_backplanes_img(eng._ctx, plane_mask(HEADLINE), 0.0, ptrs, _lib.PM_MEM_HOST))

    lo, med = best(lambda: into(keep), reps)
    yield {'case': 'frame -> reused pageable arrays', 'ms_best': lo * 1e3, 'ms_median': med * 1e3, 'GBps': nbytes / lo / 1e9,
           'Mpix_s': sz * sz / lo / 1e6}
    t = time.perf_counter()
    pinned = {n: eng.pinned_empty((sz, sz)) for n in HEADLINE}
    t_alloc = time.perf_counter() - t
    lo, med = best(lambda: into(pinned), reps)
    yield {'case': 'frame -> pinned arrays (pm_host_alloc)', 'ms_best': lo * 1e3, 'ms_median': med * 1e3,
           'GBps': nbytes / lo / 1e9, 'Mpix_s': sz * sz / lo / 1e6, 'pinned_alloc_ms': t_alloc * 1e3}
    for n in HEADLINE:
        assert np.array_equal(pinned[n], keep[n], equal_nan=True), n


def cube(eng, g, planes, reps, chunks=(32,)):
    sz = 1024
    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon, lat = rectangular_grid(bool(g.west_positive))
    xm, ym = eng.xy_map(lon, lat)
    rng = np.random.default_rng(5)
    t = time.perf_counter()
    pinned = eng.pinned_empty((planes, sz, sz))
    t_alloc = time.perf_counter() - t
    base = rng.standard_normal((8, sz, sz))
    base[rng.random(base.shape) < 1e-3] = np.nan
    for p in range(planes):
        pinned[p] = base[p % 8] * (1.0 + 0.01 * p)
    pageable = np.array(pinned)  # touched pageable copy
    nbytes = pageable.nbytes
    res = {}

    def run(c, key):
        res.pop(key, None)
        res[key] = eng.map_cube(c, xm, ym, 'linear', True)

    eng.map_cube(pageable[:8], xm, ym, 'linear', True)  # warm-up
    for chunk_mib in chunks:
        eng.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, chunk_mib << 20)
        eng.map_cube(pageable[:64], xm, ym, 'linear', True)  # (re)allocations of the ring
        eng.set_option(_lib.PM_OPT_ZERO_COPY, 0)
        lo, med = best(lambda: run(pageable, 'pageable'), reps)
        yield {'case': 'cube pageable -> pipelined copy of whole planes', 'chunk_MiB': chunk_mib, 'planes': planes, 'ms_best': lo * 1e3,
               'ms_median': med * 1e3, 'cube_GBps': nbytes / lo / 1e9, 'Mpix_s': planes * sz * sz / lo / 1e6}
        eng.set_option(_lib.PM_OPT_ZERO_COPY, -1)
        lo, med = best(lambda: run(pageable, 'pageable_blocks'), reps)
        yield {'case': 'cube pageable -> block table collected by the copy threads', 'chunk_MiB': chunk_mib, 'planes': planes,
               'ms_best': lo * 1e3, 'ms_median': med * 1e3, 'cube_GBps_equivalent': nbytes / lo / 1e9,
               'Mpix_s': planes * sz * sz / lo / 1e6}
        eng.set_option(_lib.PM_OPT_ZERO_COPY, 0)
        lo, med = best(lambda: run(pinned, 'pinned_copy'), reps)
        yield {'case': 'cube pinned -> pipelined DMA (zero copy off)', 'chunk_MiB': chunk_mib, 'planes': planes,
               'ms_best': lo * 1e3, 'ms_median': med * 1e3, 'cube_GBps': nbytes / lo / 1e9,
               'Mpix_s': planes * sz * sz / lo / 1e6, 'pinned_alloc_ms': t_alloc * 1e3}
        eng.set_option(_lib.PM_OPT_ZERO_COPY, -1)
    eng.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, 32 << 20)
    eng.set_option(_lib.PM_OPT_ZERO_COPY, 1)
    lo, med = best(lambda: run(pinned, 'zero_copy'), reps)
    yield {'case': 'cube pinned -> gathered in place (zero copy)', 'planes': planes, 'ms_best': lo * 1e3,
           'ms_median': med * 1e3, 'cube_GBps_equivalent': nbytes / lo / 1e9, 'Mpix_s': planes * sz * sz / lo / 1e6}
    eng.set_option(_lib.PM_OPT_ZERO_COPY, 2)
    lo, med = best(lambda: run(pinned, 'blocks'), reps)
    yield {'case': 'cube pinned -> table of sampled 256-byte blocks', 'planes': planes, 'ms_best': lo * 1e3,
           'ms_median': med * 1e3, 'cube_GBps_equivalent': nbytes / lo / 1e9, 'Mpix_s': planes * sz * sz / lo / 1e6}
    eng.set_option(_lib.PM_OPT_ZERO_COPY, 3)
    lo, med = best(lambda: run(pinned, 'host_blocks'), reps)
    yield {'case': 'cube pinned -> table of sampled 16-byte blocks collected by the copy threads', 'planes': planes,
           'ms_best': lo * 1e3, 'ms_median': med * 1e3, 'cube_GBps_equivalent': nbytes / lo / 1e9, 'Mpix_s': planes * sz * sz / lo / 1e6}
    eng.set_option(_lib.PM_OPT_ZERO_COPY, -1)
    out_pinned = eng.pinned_empty((planes,) + xm.shape)

    def direct():
        eng._check(eng._lib.pm_map_cube(eng._ctx, pinned.ctypes.data, 0, planes, xm.ctypes.data, ym.ctypes.data,
                                        xm.shape[0], xm.shape[1], _lib.PM_INTERP_LINEAR, 1, out_pinned.ctypes.data,
                                        _lib.PM_MEM_HOST))

    lo, med = best(direct, reps)
    yield {'case': 'cube pinned -> gathered in place, pinned output', 'planes': planes, 'ms_best': lo * 1e3,
           'ms_median': med * 1e3, 'cube_GBps_equivalent': nbytes / lo / 1e9, 'Mpix_s': planes * sz * sz / lo / 1e6}
    # where the output leg goes: the same calls into a reused (touched) pageable array / a pinned array
    out_touched = np.zeros((planes,) + xm.shape)

    def call(cube_arr, out_arr):
        eng._check(eng._lib.pm_map_cube(eng._ctx, cube_arr.ctypes.data, 0, planes, xm.ctypes.data, ym.ctypes.data,
                                        xm.shape[0], xm.shape[1], _lib.PM_INTERP_LINEAR, 1, out_arr.ctypes.data,
                                        _lib.PM_MEM_HOST))

    for label, zc in (('DMA', 0), ('gather', 1), ('fetched block table', 2), ('collected block table', 3)):
        eng.set_option(_lib.PM_OPT_ZERO_COPY, zc)
        for oname, oarr in (('reused pageable output', out_touched), ('pinned output', out_pinned)):
            lo, med = best(lambda: call(pinned, oarr), reps)
            yield {'case': f'cube pinned ({label}) -> {oname}', 'planes': planes, 'ms_best': lo * 1e3, 'ms_median': med * 1e3}
        lo, med = best(lambda: call(pageable, out_pinned), reps)
        yield {'case': f'cube pageable -> pinned output (zero_copy={zc})', 'planes': planes, 'ms_best': lo * 1e3, 'ms_median': med * 1e3}
    eng.set_option(_lib.PM_OPT_ZERO_COPY, -1)
    for k in ('pinned_copy', 'zero_copy', 'blocks', 'host_blocks', 'pageable_blocks'):
        assert np.array_equal(res[k], res['pageable'], equal_nan=True), k
    assert np.array_equal(out_pinned, res['pageable'], equal_nan=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--planes', type=int, default=512)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', choices=['frame', 'cube'])
    ap.add_argument('--copy-threads', type=int, default=0, help='PM_OPT_HOST_COPY_THREADS (0: library default)')
    ap.add_argument('--chunks-mib', default='32', help='comma-separated PM_OPT_HOST_CHUNK_BYTES values to sweep (cube)')
    args = ap.parse_args()
    eng = Engine(0)
    if args.copy_threads:
        eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, args.copy_threads)
    g = load_scenario('jupiter_hst_2005')
    if args.only in (None, 'frame'):
        for r in frame(eng, g, args.size, args.reps):
            print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    if args.only in (None, 'cube'):
        for r in cube(eng, g, args.planes, args.reps, tuple(int(c) for c in args.chunks_mib.split(','))):
            print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
