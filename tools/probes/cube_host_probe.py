#!/usr/bin/env python3
"""
Where the time of a host-fed cube call goes (config 5 geometry): `PM_DEBUG_ENV=1 PM_HOSTPIPE_TRACE=1 python
tools/probes/cube_host_probe.py [--planes 512,64] [--threads 16,2] [--chunk-mib 32]` prints the library's own
stage trace (stderr) and the wall time of each step (x/y map + pm_map_cube(PM_MEM_HOST_CUBE) + finish).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--planes', default='512,64')
    ap.add_argument('--threads', default='0')
    ap.add_argument('--chunk-mib', type=int, default=32)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--route', type=int, default=-1)
    ap.add_argument('--pageable', action='store_true')
    args = ap.parse_args()
    import torch

    import bench
    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    g = load_scenario('jupiter_hst_2005')
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    sz = 1024
    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lon_h, lat_h = bench.rectangular_grid(bool(g.west_positive), 1.0)
    n0, n1 = lon_h.shape
    lon_d, lat_d = torch.from_numpy(lon_h).cuda(), torch.from_numpy(lat_h).cuda()
    xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda')
    ym = torch.empty_like(xm)
    pmax = max(int(p) for p in args.planes.split(','))
    gen = torch.Generator(device='cuda').manual_seed(5)
    cube_d = torch.randn((pmax, sz, sz), generator=gen, device='cuda', dtype=torch.float64)
    cube_h = np.empty((pmax, sz, sz)) if args.pageable else eng.pinned_empty((pmax, sz, sz))
    torch.from_numpy(cube_h).copy_(cube_d)
    out = torch.empty((pmax, n0, n1), dtype=torch.float64, device='cuda')
    eng.set_option(_lib.PM_OPT_HOST_CHUNK_BYTES, args.chunk_mib << 20)
    eng.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, args.route)
    for threads in (int(t) for t in args.threads.split(',')):
        eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, threads)
        for planes in (int(p) for p in args.planes.split(',')):
            eng.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)
            ts = []
            for rep in range(args.reps + 2):
                torch.cuda.synchronize()
                print(f'--- threads {threads} planes {planes} rep {rep}', file=sys.stderr, flush=True)
                t = time.perf_counter()
                eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
                t1 = time.perf_counter()
                eng.map_cube_host_to_device(cube_h[:planes], xm, ym, n0, n1, out[:planes])
                t2 = time.perf_counter()
                eng.synchronize()
                t3 = time.perf_counter()
                if rep >= 2:
                    ts.append(t3 - t)
                print(f'    xy_map {1e3 * (t1 - t):.3f} ms, map_cube {1e3 * (t2 - t1):.3f} ms, sync {1e3 * (t3 - t2):.3f} ms', file=sys.stderr)
            print({'threads': threads, 'planes': planes, 'route': eng.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE),
                   'ms_median': round(float(np.median(ts)) * 1e3, 3), 'ms_min': round(min(ts) * 1e3, 3),
                   'ns_per_plane': [eng.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(4)]}, flush=True)
    eng.close()


if __name__ == '__main__':
    main()
