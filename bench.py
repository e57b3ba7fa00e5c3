#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): Mpix/s for the full backplane set
(LON/LAT-GRAPHIC, PHASE, INCIDENCE, EMISSION) + map reprojection of a 4096 x 4096 frame.

One step = one frame through the hot path, everything resident in HBM:
  1. pm_backplanes_img   5 planes, 4096^2           (kernel k_disc<ILLUM>)
  2. pm_xy_map           1 deg rectangular grid     (kernel k_map, 180 x 360)
  3. pm_map_cube         1 data plane -> (180, 360) (kernel k_reproject<f64>)
With N > 1 GPUs every rank processes its own frame (weak scaling, no data-path
collective in the backplane stage) and the reprojected planes - one per rank, i.e. the
wavelength planes of `Observation.get_mapped_data` sharded one per GPU - are combined
by ONE RCCL all-gather, the only exchange step the path has.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096]
Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
                   --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def rectangular_grid(west_positive: bool, degree_interval: float = 1.0):
    """BodyXY.generate_map_coordinates('rectangular') body_xy.py:2899-2907."""
    lons = np.arange(degree_interval / 2, 360, degree_interval)
    if west_positive:
        lons = lons[::-1]
    lats = np.arange(-90 + degree_interval / 2, 90, degree_interval)
    lon, lat = np.meshgrid(lons, lats)
    return np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat)


def measured_traffic(kernel_prefix: str):
    """
    HBM bytes per launch of the dominant kernel from the PMC passes of tools/pmc_profile.sh
    (WRITE_SIZE + FETCH_SIZE, separate rocprofv3 --pmc runs; profiles/traffic.json). PMC
    counters cannot be collected inside this process, so the last profiled value is reported;
    None if no profile has been recorded.
    """
    try:
        with open(os.path.join(REPO, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        for k, v in t.items():
            if k.startswith(kernel_prefix):
                return int(v['hbm_bytes'])
    except (OSError, ValueError, KeyError):
        pass
    return None


def algorithmic_bytes(nx: int, ny: int, n_planes: int) -> int:
    """the image kernel reads nothing and writes 8 B per plane per pixel."""
    return nx * ny * 8 * n_planes


def host_cores() -> int:
    """CPU cores this process may actually use: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(g, sz: int, budget_s: float = 12.0) -> dict:
    """
    The CPU oracle (a port: the reference's own Python + CSPICE path cannot run here)
    timed on this box's host cores on the same 5-plane frame; all cores via OpenMP.
    """
    from oracle import oracle

    cores = host_cores()
    oracle.set_num_threads(cores)
    x0 = y0 = (sz - 1) / 2
    disc = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
    lon, lat = rectangular_grid(bool(g.west_positive))
    img = np.zeros((sz, sz))
    # bounded sample: whole frames until the budget is used (at least one)
    t0 = time.perf_counter()
    frames = 0
    while True:
        oracle.backplanes_img(g, disc, HEADLINE)
        xm, ym = oracle.xy_map(g, disc, lon, lat)
        oracle.map_cube(img, xm, ym, 'linear', True)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= 8:
            break
    return {
        'value': round(frames * sz * sz / dt / 1e6, 3),
        'unit': 'Mpix/s',
        'cores': cores,
        'kind': 'port',
        'sample': f'{frames} full {sz}x{sz} frame(s), 5 planes + 1 deg reprojection, OpenMP over rows',
    }


def other_workloads(args) -> None:
    """Secondary BASELINE configs (4 and 5); same JSON contract, their own metric names."""
    import torch
    import torch.distributed as dist

    from planetmapper_amd.distributed import map_cube_sharded_device, shard_bounds
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    eng = Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == 'saturn':
        # Saturn-like spheroid, 4096^2, r0 = 800 px, rotation 20 deg
        sz = args.size
        names = HEADLINE + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
        g = load_scenario('saturn_earth_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 800.0 * sz / 4096, float(np.deg2rad(20.0)), sz, sz, True)
        planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in names}

        def step():
            eng.backplanes_img_device(planes)

        units = sz * sz
        metric = 'Mpix/s Saturn + rings backplane set (lon/lat/phase/inc/emi/ring radius/lon/distance), 4096^2 frame'
        workload = f'Saturn-like spheroid seen from Earth 2005-01-01, {sz}x{sz}, r0={800.0 * sz / 4096:g} px, 8 planes'
        alg = sz * sz * 8 * len(names)
        scaling = 'weak'
    else:
        # IFU cube P x 1024 x 1024 f64, 1 deg map, planes sharded over ranks
        sz = 1024
        g = load_scenario('jupiter_hst_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
        a, b, per_rank = shard_bounds(args.planes, world, rank)
        gen = torch.Generator(device=dev).manual_seed(5 + rank)
        cube = torch.randn((per_rank, sz, sz), generator=gen, device=dev, dtype=torch.float64)
        cube[torch.rand((per_rank, sz, sz), generator=gen, device=dev) < 1e-3] = float('nan')
        lon_h, lat_h = rectangular_grid(bool(g.west_positive))
        n0, n1 = lon_h.shape
        lon_d, lat_d = torch.from_numpy(lon_h).to(dev), torch.from_numpy(lat_h).to(dev)
        xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
        ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
        gathered = torch.empty((world, per_rank, n0, n1), dtype=torch.float64, device=dev)

        def step():
            eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
            map_cube_sharded_device(eng, cube, np.float64, per_rank, xm, ym, n0, n1, gathered, rank)

        units = args.planes * sz * sz
        metric = 'Mpix/s of cube pixels map-projected (get_mapped_data, 1 deg rectangular map, bilinear)'
        workload = (
            f'synthetic IFU cube {args.planes}x{sz}x{sz} f64 resident in HBM, {per_rank} planes per GPU, '
            f'map {n0}x{n1}, RCCL all-gather of mapped planes' + ('' if world > 1 else ' (skipped at N=1)')
        )
        alg = n0 * n1 * (16 + 40 * per_rank)
        scaling = 'strong'
    for _ in range(min(args.preheat_steps, 200)):  # untimed clock ramp (see main)
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    barrier()
    dt = time.perf_counter() - t0
    eng.synchronize()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if rank == 0:
        n_units = units * (world if scaling == 'weak' else 1)
        print(json.dumps({
            'metric': metric, 'value': round(n_units * args.steps / dt / 1e6, 2), 'unit': 'Mpix/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload},
            'roofline': {'bound': 'hbm', 'achieved': round(alg / (step_ms * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         'traffic': None, 'kernel_ms': round(step_ms, 4), 'algorithmic_bytes': alg},
        }), flush=True)  # fmt: skip
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--preheat-steps', type=int, default=500, help='untimed clock-ramp steps before the warmup steps')
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument(
        '--workload', default='frame', choices=['frame', 'saturn', 'cube'],
        help="frame: BASELINE headline (default); saturn: config 4 (Saturn + rings, 8 planes); "
        "cube: config 5 (512 x 1024^2 f64 cube -> 1 deg map, planes sharded over the GPUs)",
    )  # fmt: skip
    ap.add_argument('--planes', type=int, default=512, help='cube workload: total planes')
    args = ap.parse_args()
    if args.workload != 'frame':
        return other_workloads(args)

    import torch
    import torch.distributed as dist

    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N > 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the engine has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    sz = args.size
    g = load_scenario('jupiter_hst_2005')
    x0 = y0 = (sz - 1) / 2
    eng = Engine(local_rank)
    # launch on torch's current stream so torch events / RCCL order with our kernels
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_geometry(g)
    eng.set_disc(x0, y0, 0.9 * x0, 0.0, sz, sz, True)  # BodyXY.centre_disc body_xy.py:791

    planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in HEADLINE}
    lon_h, lat_h = rectangular_grid(bool(g.west_positive))
    n0, n1 = lon_h.shape
    lon_d = torch.from_numpy(lon_h).to(dev)
    lat_d = torch.from_numpy(lat_h).to(dev)
    xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
    ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
    # synthetic data plane: limb-darkened disc + noise (SURVEY 8d config 3 recipe)
    gen = torch.Generator(device=dev).manual_seed(20050101 + rank)
    yy, xx = torch.meshgrid(
        torch.arange(sz, device=dev, dtype=torch.float64),
        torch.arange(sz, device=dev, dtype=torch.float64),
        indexing='ij',
    )
    mu = torch.sqrt(torch.clamp(1 - ((xx - x0) ** 2 + (yy - y0) ** 2) / (0.9 * x0) ** 2, min=0))
    data = mu + 0.05 * torch.randn((sz, sz), generator=gen, device=dev, dtype=torch.float64)
    del yy, xx, mu
    from planetmapper_amd.distributed import map_cube_sharded_device

    # two result buffers used alternately: a gather may stay in flight for two frames before its
    # buffer is written again (RCCL latency at 8 ranks is not known to be below one 0.19 ms step)
    gathered = [torch.empty((world, 1, n0, n1), dtype=torch.float64, device=dev) for _ in range(2)]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i: int | None) -> None:
        if i is not None:
            ev0[i].record()
        eng.backplanes_img_device(planes)
        if i is not None:
            ev1[i].record()
        eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
        # this rank's plane -> its slot; slots exchanged by one RCCL all-gather (N > 1 only),
        # left in flight so that it overlaps the next frame's backplane kernel
        k = counter[0] & 1
        counter[0] += 1
        pending[k] = map_cube_sharded_device(
            eng, data, np.float64, 1, xm, ym, n0, n1, gathered[k], rank, 'linear', True, async_op=True, previous=pending[k]
        )

    pending = [None, None]  # work handles of the all-gathers still in flight, one per buffer
    counter = [0]

    def barrier() -> None:
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed clock ramp: a fresh box starts in a low power state and the shader clock takes
    # tens of milliseconds of sustained load to settle; without it the first ~50 frames (all
    # of a short run) are timed on the way up. Same step as the timed one, never counted.
    # (a fixed count, not a wall-clock loop: every rank must issue the same collectives)
    for _ in range(args.preheat_steps):
        step(None)
    barrier()
    for _ in range(args.warmup):
        step(None)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    eng.synchronize()  # surfaces deferred device-side errors
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    frac_on_disc = float(torch.isfinite(planes['LON-GRAPHIC']).double().mean().item())

    if rank == 0:
        alg = algorithmic_bytes(sz, sz, len(HEADLINE))
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        line = {
            'metric': 'Mpix/s full backplane set (lat/lon/inc/emi/phase) + map-reproject, 4096^2 frame',
            'value': round(world * sz * sz * args.steps / dt / 1e6, 2),
            'unit': 'Mpix/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': f'Jupiter/HST 2005-01-01 geometry, {sz}x{sz} frame per GPU, centred disc '
                f'(r0=0.9*x0, {frac_on_disc:.3f} of pixels on disc), planes {"/".join(HEADLINE)}, '
                '+ x/y map and bilinear reprojection of 1 f64 plane to a 1 deg rectangular map (180x360)',
                'frame': [sz, sz],
                'planes': len(HEADLINE),
                'map': [n0, n1],
                'parallelism': f'frames (and their mapped planes) sharded 1 per GPU x{world}'
                + (', RCCL all-gather of mapped planes' if world > 1 else ''),
                'preheat_steps': args.preheat_steps,
            },
            'roofline': {
                'kernel': 'pm::k_disc_sph<1, false> (DF_ILLUM, spheroid)',
                'bound': 'hbm',
                'achieved': round(achieved, 2),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5),
                'traffic': measured_traffic('pm::k_disc_sph<1,') if sz == 4096 else None,
                'kernel_ms': round(kernel_ms, 4),
                'algorithmic_bytes': alg,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(g, sz)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
