#!/usr/bin/env python3
"""
Soak run of the host <-> HBM leg (pm_hostpipe.hip; test infrastructure, GPU box only): random cubes
(size, plane count, dtype, NaN / inf content), random maps, every PM_OPT_ZERO_COPY route, random chunk
sizes and copy-thread counts, pageable and pinned cubes and outputs - each result must be
BIT-IDENTICAL to the device-resident kernel's; and frames of random size into fresh numpy arrays
against the same planes computed into device buffers. Threads, staging slots and out-of-order
copy-outs are what this exercises.

    python tests/soak_hostpath.py [--iterations 300] [--seed 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iterations', type=int, default=300)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()
    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine, dtype_code
    from planetmapper_amd.scenarios import load_scenario

    rng = np.random.default_rng(args.seed)
    eng = Engine(0)
    g = load_scenario('jupiter_hst_2005')
    eng.set_geometry(g)
    dtypes = [np.float64, np.float32, np.int16, np.int32, np.uint8]
    names = ['LON-GRAPHIC', 'EMISSION', 'RA', 'RING-RADIUS', 'PIXEL-X']
    bad = 0
    t0 = time.time()
    moved = 0
    for it in range(args.iterations):
        nx, ny = (int(v) for v in rng.integers(48, 1400, 2))
        planes = int(rng.integers(1, 41))
        while planes * nx * ny * 8 > 600e6:
            planes = max(1, planes // 2)
        dtype = dtypes[int(rng.integers(0, len(dtypes)))]
        r0 = float(min(nx, ny) * rng.uniform(0.15, 0.6))
        eng.set_disc(float(rng.uniform(0.3, 0.7) * nx), float(rng.uniform(0.3, 0.7) * ny), r0, float(rng.uniform(0, 6.28)), nx, ny, True)
        step = float(rng.choice([0.5, 1.0, 2.0, 5.0]))
        lons = np.arange(step / 2, 360, step)[::-1] if g.west_positive else np.arange(step / 2, 360, step)
        lon, lat = np.meshgrid(lons, np.arange(-90 + step / 2, 90, step))
        xm, ym = eng.xy_map(np.ascontiguousarray(lon), np.ascontiguousarray(lat))
        n0, n1 = xm.shape
        if np.issubdtype(dtype, np.floating):
            cube = (rng.standard_normal((planes, ny, nx)) * 5).astype(dtype)
            cube[rng.random(cube.shape) < 2e-3] = np.nan
            if rng.random() < 0.3:
                p = int(rng.integers(0, planes))
                cube[p][ny // 3 : ny // 3 + 9, nx // 4 : nx // 2] = np.inf
        else:
            info = np.iinfo(dtype)
            cube = rng.integers(info.min, info.max, (planes, ny, nx), dtype=dtype)
        interp = 'linear' if rng.random() < 0.7 else 'nearest'
        mode = int(rng.integers(-1, 5))  # -1 the library chooses, 0 whole, 1 in place, 2 fetched, 3 collected, 4 hybrid
        chunk = int(rng.choice([1, 2, 8, 32, 64])) << 20
        threads = int(rng.choice([1, 2, 5, 8, 16]))
        pin_cube, pin_out = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        eng.set_option(_lib.PM_OPT_ZERO_COPY, mode)
        eng.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, chunk)
        eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, threads)
        src = eng.pinned_copy(cube) if pin_cube else cube
        out = eng.pinned_empty((planes, n0, n1)) if pin_out else np.empty((planes, n0, n1))
        out[...] = -12345.0
        eng._check(eng._lib.pm_map_cube(
            eng._ctx, src.ctypes.data, dtype_code(cube.dtype), planes, xm.ctypes.data, ym.ctypes.data, n0, n1,
            _lib.PM_INTERP_LINEAR if interp == 'linear' else _lib.PM_INTERP_NEAREST, 1, out.ctypes.data, _lib.PM_MEM_HOST,
        ))  # fmt: skip
        raw = cube.view(np.int16) if cube.dtype == np.uint16 else cube
        dcube, dxm, dym = torch.from_numpy(raw).cuda(), torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
        dout = torch.empty((planes, n0, n1), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        eng.map_cube_device(dcube, cube.dtype, planes, dxm, dym, n0, n1, dout, interp, True)
        eng.synchronize()
        same_cube = bool(np.array_equal(out, dout.cpu().numpy(), equal_nan=True))
        # a frame into fresh numpy arrays against device buffers (sparse / whole / the library's choice)
        eng.set_option(_lib.PM_OPT_SPARSE_FRAME, int(rng.integers(-1, 2)))
        fresh = eng.backplanes_img(names)
        dev = {n: torch.empty((ny, nx), dtype=torch.float64, device='cuda') for n in names}
        eng.backplanes_img_device(dev)
        eng.synchronize()
        same_frame = all(np.array_equal(fresh[n], dev[n].cpu().numpy(), equal_nan=True) for n in names)
        moved += cube.nbytes + out.nbytes + len(names) * nx * ny * 8
        rec = {'it': it, 'nx': nx, 'ny': ny, 'planes': planes, 'dtype': np.dtype(dtype).name, 'map': [n0, n1], 'interp': interp,
               'zero_copy': mode, 'chunk_MiB': chunk >> 20, 'threads': threads, 'pinned_cube': pin_cube, 'pinned_out': pin_out,
               'cube_identical': same_cube, 'frame_identical': same_frame}  # fmt: skip
        if not (same_cube and same_frame):
            bad += 1
            print(json.dumps(rec), flush=True)
        elif it % 25 == 0:
            print(json.dumps(rec), flush=True)
        del src, out, dcube, dout
    print(json.dumps({'iterations': args.iterations, 'failed': bad, 'GB_moved': round(moved / 1e9, 1), 'seconds': round(time.time() - t0, 1)}))
    eng.close()
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
