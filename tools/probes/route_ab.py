"""
Same-process A/B of the host-cube routes (PM_OPT_HOST_CUBE_ROUTE) on the block of one rank of an N-rank run:
ceil(512 / N) planes of 1024^2 f64 from pinned host memory -> 1 deg map, device output, for several copy-thread
counts. Forced routes 2 (GPU fetch), 3 (collected), 4 (hybrid) and the library's own choice (-1), interleaved
round-robin so that box and clock drift hit all alike; median / min of `--reps` calls each.
Usage (GPU box): python tools/probes/route_ab.py [--n 8] [--threads 2,4,16] [--reps 9]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--threads', default='2,4,16')
    ap.add_argument('--reps', type=int, default=9)
    ap.add_argument('--routes', default='2,3,4,-1')
    ap.add_argument('--fetch-bytes', type=int, default=128, help='PM_OPT_FETCH_BLOCK_BYTES: 128 or 256')
    args = ap.parse_args()
    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    sz, planes = 1024, -(-512 // args.n)
    g = load_scenario('jupiter_hst_2005')
    eng = Engine(0)
    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_option(_lib.PM_OPT_FETCH_BLOCK_BYTES, args.fetch_bytes)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    lons = np.arange(0.5, 360, 1.0)[::-1] if g.west_positive else np.arange(0.5, 360, 1.0)
    lon, lat = np.meshgrid(lons, np.arange(-89.5, 90, 1.0))
    xm, ym = eng.xy_map(np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat))
    n0, n1 = xm.shape
    dxm, dym = torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda()
    cube = eng.pinned_empty((planes, sz, sz))
    cube[...] = np.random.default_rng(1).standard_normal(cube.shape)
    out = torch.empty((planes, n0, n1), dtype=torch.float64, device='cuda')
    ref = None
    routes = [int(r) for r in args.routes.split(',')]
    for threads in [int(t) for t in args.threads.split(',')]:
        eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, threads)
        eng.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)
        ts = {r: [] for r in routes}
        for rep in range(args.reps + (8 if -1 in routes else 2)):
            for r in routes:
                eng.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, r)
                t = time.perf_counter()
                eng.map_cube_host_to_device(cube, dxm, dym, n0, n1, out)
                eng.synchronize()
                dt = time.perf_counter() - t
                if rep >= (8 if -1 in routes else 2):
                    ts[r].append(dt)
                got = out.clone()
                if ref is None:
                    ref = got
                assert torch.equal(torch.nan_to_num(ref, nan=-1.0), torch.nan_to_num(got, nan=-1.0)), (threads, r)
        rec = {'N': args.n, 'fetch_block_bytes': args.fetch_bytes, 'planes': planes, 'copy_threads': threads,
               'ms': {str(r): {'median': round(float(np.median(v)) * 1e3, 3), 'min': round(min(v) * 1e3, 3)} for r, v in ts.items()},
               'auto_route': eng.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE),
               'hybrid_fetch_permille': eng.get_option(_lib.PM_OPT_HYBRID_FETCH_PERMILLE),
               'route_ns_per_plane': {str(r): eng.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(5)}}
        print(json.dumps(rec), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
