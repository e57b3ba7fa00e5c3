"""
Ranks of a multi-rank run on ONE MI355X (test infrastructure; started by tests/test_multirank_one_gpu.py and
by bench.py's shared-GPU section as child processes):

  capi      pm_comm_* / pm_map_cube_sharded (the C ABI's sharded cube, pm_comm.hip) at world sizes 2 ... 8 over
            the loopback transport of tests/loopback (PM_RCCL_LIBRARY; RCCL itself refuses two ranks on one
            GPU): ranks are the threads of this process (--threads) or one process each (--rank R of --world W,
            the unique id handed over through a file)
  pyproto   distributed.map_cube_sharded_pipelined over a gloo process group, one process per rank, every
            rank a REAL Engine on device 0 with its own pinned host block

Every rank checks for itself that the gathered cube equals, bit for bit, what it gets by mapping the whole cube
alone, and prints one JSON line; exit status 0 = every check of this process passed.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [REPO, HERE]
LOOPBACK = os.path.join(HERE, 'loopback', 'libpm_loopback_nccl.so')


def grid(west_positive: bool, interval: float):
    lons = np.arange(interval / 2, 360, interval)
    if west_positive:
        lons = lons[::-1]
    lats = np.arange(-90 + interval / 2, 90, interval)
    lon, lat = np.meshgrid(lons, lats)
    return np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat)


def uneven_planes(world: int) -> int:
    """a cube whose last rank gets no plane at all (and, from 4 ranks on, the one before it a short block)"""
    return {2: 1, 3: 4}.get(world, 2 * world - 3)


_SHARED: dict = {}  # ranks that are THREADS of one process share the cube and its one-rank reference (config 5: 4 GiB each)
_SHARED_LOCK = threading.Lock()


def shared(key, make):
    with _SHARED_LOCK:
        if key not in _SHARED:
            _SHARED.clear()  # (one case at a time: the previous case's cube and reference go)
            _SHARED[key] = make()
        return _SHARED[key]


def make_cube(planes: int, sz: int, redo_plane: int | None):
    rng = np.random.default_rng(4242)
    if planes * sz * sz > (1 << 27):  # (plane by plane: the masks of a 4 GiB cube are 4.5 GiB more)
        cube = np.empty((planes, sz, sz))
        for p in range(planes):
            cube[p] = rng.standard_normal((sz, sz)) + 2.0
            cube[p][rng.random((sz, sz)) < 2e-3] = np.nan
    else:
        cube = rng.standard_normal((planes, sz, sz)) + 2.0
        cube[rng.random(cube.shape) < 2e-3] = np.nan
    if redo_plane is not None:
        # an -inf block: the pixels in its interior have no finite neighbour and take the plane's nanmedian -
        # the plane is mapped, flagged, and redone behind the pipeline AFTER its exchange may have started
        cube[redo_plane, sz // 3: sz // 3 + sz // 4, sz // 4: sz // 2] = -np.inf
    return cube


def setup_engine(sz: int, interval: float):
    import torch

    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    g = load_scenario('jupiter_hst_2005')
    eng = Engine(0)
    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.2, sz, sz, True)
    lon, lat = grid(bool(g.west_positive), interval)
    xm, ym = eng.xy_map(lon, lat)
    n0, n1 = xm.shape
    return eng, torch.from_numpy(xm).cuda(), torch.from_numpy(ym).cuda(), n0, n1


def reference(eng, cube, dxm, dym, n0, n1):
    """the whole cube mapped by this rank alone"""
    import torch

    dcube = torch.from_numpy(cube).cuda()
    ref = torch.empty((cube.shape[0], n0, n1), dtype=torch.float64, device='cuda')
    eng.map_cube_device(dcube, np.float64, cube.shape[0], dxm, dym, n0, n1, ref)
    eng.synchronize()
    return ref


def same(a, b) -> bool:
    import torch

    return bool(torch.equal(torch.nan_to_num(a, nan=-1.25e300), torch.nan_to_num(b, nan=-1.25e300))) and bool(
        torch.equal(torch.isnan(a), torch.isnan(b)))


# ------------------------------------------------------------------ C ABI over the loopback transport
def capi_rank(rank: int, world: int, uid: bytes, args, report: dict) -> None:
    import torch

    from planetmapper_amd import _lib
    from planetmapper_amd.distributed import Comm, shard_bounds

    eng, dxm, dym, n0, n1 = setup_engine(args.size, args.interval)
    checks = {}
    try:
        comm = Comm(eng, world, rank, uid)
        try:
            for case in args.cases.split(','):
                planes = args.planes
                redo = None
                if case == 'uneven':
                    planes = uneven_planes(world)
                if case == 'redo':
                    redo = planes // 2
                if args.threads:
                    cube, ref = shared((case, planes, args.size, redo), lambda: (lambda c: (c, reference(eng, c, dxm, dym, n0, n1)))(make_cube(planes, args.size, redo)))
                else:
                    cube = make_cube(planes, args.size, redo)
                    ref = reference(eng, cube, dxm, dym, n0, n1)
                a, b, per_rank = shard_bounds(planes, world, rank)
                out = torch.full((world, per_rank, n0, n1), -7.0, dtype=torch.float64, device='cuda')
                block = np.ascontiguousarray(cube[a:b])
                if case in ('device', 'uneven', 'redo', 'fail', 'send_fault'):
                    local = torch.from_numpy(block).cuda() if b > a else None
                    host = False
                elif case == 'host':
                    local = eng.pinned_copy(block) if b > a else None
                    host = True
                elif case == 'pageable':
                    local = block if b > a else None
                    host = True
                else:
                    raise SystemExit(f'unknown case {case}')
                if case == 'fail' and rank == 1 % world:
                    local = None  # "local_cube is NULL but this rank owns planes": this rank's mapping fails
                err = None
                try:
                    comm.map_cube_sharded(local, np.float64, planes, dxm, dym, n0, n1, out, host_cube=host)
                    eng.synchronize()
                except Exception as e:  # noqa: BLE001
                    err = e
                if case == 'fail':
                    want = ValueError if rank == 1 % world else _lib.PeerFailedError
                    checks[case] = isinstance(err, want)
                    # the communicator survives a failed mapping: the same call with every block in place works
                    local = torch.from_numpy(block).cuda() if b > a else None
                    out.fill_(-7.0)
                    comm.map_cube_sharded(local, np.float64, planes, dxm, dym, n0, n1, out)
                    eng.synchronize()
                    checks[case + '_then_ok'] = same(out.reshape(-1, n0, n1)[:planes], ref)
                    continue
                if case == 'send_fault':
                    # the transport fails on one rank in the middle of the exchanges: EVERY rank gets an error back
                    # (nobody hangs) and the communicator reports itself broken afterwards
                    checks[case] = isinstance(err, _lib.EngineError)
                    err2 = None
                    try:
                        comm.map_cube_sharded(local, np.float64, planes, dxm, dym, n0, n1, out)
                    except Exception as e:  # noqa: BLE001
                        err2 = e
                    checks[case + '_comm_broken'] = isinstance(err2, _lib.EngineError)
                    continue
                if err is not None:
                    checks[case] = f'{type(err).__name__}: {err}'
                    continue
                got = out.reshape(-1, n0, n1)
                ok = same(got[:planes], ref)
                if world * per_rank > planes:
                    ok = ok and bool(torch.isnan(got[planes:]).all())  # padding of the short last blocks
                if case == 'redo':
                    ok = ok and (eng.last_redo_planes() > 0) == (a <= redo < b)
                    ok = ok and bool(torch.isfinite(ref[redo]).any())
                checks[case] = ok
        finally:
            comm.close()
    finally:
        eng.close()
    report[rank] = checks


def loopback_stats() -> list[int]:
    lib = ctypes.CDLL(LOOPBACK)
    out = (ctypes.c_long * 6)()
    lib.pm_loopback_stats(out)
    return list(out)


def run_capi(args) -> int:
    assert os.environ.get('PM_RCCL_LIBRARY') == LOOPBACK, 'PM_RCCL_LIBRARY must name the loopback transport'
    from planetmapper_amd.distributed import Comm

    report: dict = {}
    if args.threads:
        uid = Comm.unique_id()
        errors = []

        def body(r):
            try:
                capi_rank(r, args.world, uid, args, report)
            except BaseException as e:  # noqa: BLE001
                errors.append(f'rank {r}: {type(e).__name__}: {e}')

        ts = [threading.Thread(target=body, args=(r,)) for r in range(args.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(args.deadline)
        hung = [i for i, t in enumerate(ts) if t.is_alive()]
        ok = not hung and not errors and len(report) == args.world and all(all(v is True for v in c.values()) for c in report.values())
        print(json.dumps({'mode': 'capi-threads', 'world': args.world, 'ok': ok, 'hung': hung, 'errors': errors,
                          'checks': {str(k): v for k, v in sorted(report.items())},
                          'loopback_stats[groups,sends,recvs,bytes,allreduces,aborts]': loopback_stats()}), flush=True)
        if hung:
            os._exit(3)  # (threads blocked in C cannot be joined)
        return 0 if ok else 1
    # one process per rank: rank 0 makes the id
    path = args.uid_file
    if args.rank == 0:
        uid = Comm.unique_id()
        with open(path + '.tmp', 'wb') as f:
            f.write(uid)
        os.replace(path + '.tmp', path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise SystemExit('no unique id from rank 0')
            time.sleep(0.01)
        uid = open(path, 'rb').read()
    capi_rank(args.rank, args.world, uid, args, report)
    ok = all(v is True for v in report[args.rank].values())
    print(json.dumps({'mode': 'capi-procs', 'world': args.world, 'rank': args.rank, 'ok': ok, 'checks': report[args.rank],
                      'loopback_stats[groups,sends,recvs,bytes,allreduces,aborts]': loopback_stats()}), flush=True)
    return 0 if ok else 1


# ------------------------------------------------------------------ the torch.distributed protocol over gloo
def run_pyproto(args) -> int:
    import torch
    import torch.distributed as dist

    from planetmapper_amd import _lib
    from planetmapper_amd.distributed import map_cube_sharded_pipelined, shard_bounds

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    checks = {}
    timing = {}
    try:
        eng, dxm, dym, n0, n1 = setup_engine(args.size, args.interval)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)  # collectives are ordered behind torch's current stream
        for case in args.cases.split(','):
            planes = args.planes
            redo = None
            if case == 'uneven':
                planes = uneven_planes(world)
            if case == 'redo':
                redo = planes // 2
            cube = make_cube(planes, args.size, redo)
            a, b, per_rank = shard_bounds(planes, world, rank)
            ref = reference(eng, cube, dxm, dym, n0, n1)
            gathered = torch.full((world, per_rank, n0, n1), -7.0, dtype=torch.float64, device='cuda')
            block = eng.pinned_copy(np.ascontiguousarray(cube[a:b])) if b > a else eng.pinned_empty((1, args.size, args.size))
            engine = eng
            if case == 'fail' and rank == 1 % world:
                class Failing:  # this rank's mapping raises in the middle of the protocol
                    def __getattr__(self, name):
                        return getattr(eng, name)

                    def map_cube_host_to_device(self, *a_, **k_):
                        raise RuntimeError('injected: this rank cannot map its planes')

                engine = Failing()
            err = None
            t0 = time.perf_counter()
            try:
                map_cube_sharded_pipelined(engine, block, np.float64, planes, dxm, dym, n0, n1, gathered, rank, world, host_cube=True)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = e
            timing[case] = round((time.perf_counter() - t0) * 1e3, 3)
            if case == 'fail':
                want = RuntimeError if rank == 1 % world else _lib.PeerFailedError
                checks[case] = isinstance(err, want) and not (rank != 1 % world and not isinstance(err, _lib.PeerFailedError))
                # the group is usable afterwards
                map_cube_sharded_pipelined(eng, block, np.float64, planes, dxm, dym, n0, n1, gathered, rank, world, host_cube=True)
                torch.cuda.synchronize()
                checks[case + '_then_ok'] = same(gathered.reshape(-1, n0, n1)[:planes], ref)
                continue
            if err is not None:
                checks[case] = f'{type(err).__name__}: {err}'
                continue
            got = gathered.reshape(-1, n0, n1)
            ok = same(got[:planes], ref)
            if world * per_rank > planes:
                ok = ok and bool(torch.isnan(got[planes:]).all())
            if case == 'redo':
                ok = ok and (eng.last_redo_planes() > 0) == (a <= redo < b) and bool(torch.isfinite(ref[redo]).any())
            checks[case] = ok
        eng.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()
    ok = all(v is True for v in checks.values())
    print(json.dumps({'mode': 'pyproto-gloo', 'world': world, 'rank': rank, 'ok': ok, 'checks': checks, 'ms': timing}), flush=True)
    return 0 if ok else 1


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('mode', choices=['capi', 'pyproto'])
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--threads', action='store_true')
    ap.add_argument('--uid-file', default='')
    ap.add_argument('--cases', default='device,host,pageable,uneven,redo,fail')
    ap.add_argument('--planes', type=int, default=0, help='0: 20 per rank (several exchanges per block at a 1 deg map)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--interval', type=float, default=1.0)
    ap.add_argument('--deadline', type=float, default=300.0)
    args = ap.parse_args()
    if args.planes <= 0:
        args.planes = 20 * (int(os.environ.get('WORLD_SIZE', args.world)) if args.mode == 'pyproto' else args.world)
    return run_capi(args) if args.mode == 'capi' else run_pyproto(args)


if __name__ == '__main__':
    sys.exit(main())
