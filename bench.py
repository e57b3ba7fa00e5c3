#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): Mpix/s for the full backplane set
(LON/LAT-GRAPHIC, PHASE, INCIDENCE, EMISSION) + map reprojection of a 4096 x 4096 frame.

One step = one frame through the hot path, everything resident in HBM:
  1. pm_backplanes_img   5 planes, 4096^2           (kernel k_disc_sph<ILLUM>)
  2. pm_mapped_data      x/y map of the 1 deg rectangular grid (180 x 360) + bilinear reprojection of
                         1 data plane onto it = get_mapped_data (kernel k_mapped_data<f64>; the same
                         results as pm_xy_map + pm_map_cube, kernels k_map_xy + k_reproject)
With N > 1 GPUs every rank processes its own frame and maps its own data plane (weak scaling).
Frames are independent units: by default NOTHING is exchanged between the ranks in the headline
step (`config.gather_mapped` false; `--gather-mapped` adds an RCCL all-gather of the N mapped planes
per step for whoever wants to price it - numbers with and without it are not comparable). The
collective the north star names belongs to the plane-sharded cube, which is the `cube_host` section
below: that section, not the headline, is the strong-scaling figure of record.

Beside the headline line the same JSON object carries (rank 0, after the timed region):
  roofline / step_roofline / fp64   the dominant kernel and the whole step against the HBM peak,
                                    FP64 rate against the vector peak
  cpu_baseline(_1thread), cpu_model the CPU oracle on this box's host cores
  host_path                         PCIe-inclusive time of the same frame into numpy arrays
  api_path                          the same through the drop-in surface: BodyXY.get_*_img() x 5 +
                                    Observation.get_mapped_data(), cold cache, Python shim included
  interpolations                    every interpolation of map_img on a resident cube: us per plane of 1024^2
                                    onto the 1 deg and a 0.1 deg map (N = 1)
  cube_host                         BASELINE config 5, the north star's scaling case: a 512-plane
                                    1024^2 f64 cube in HOST memory, planes sharded over the ranks,
                                    each rank feeding its block over its own PCIe link, the mapped
                                    planes all-gathered (a plain form: one all-gather per step; and the
                                    pipelined protocol: exchanges started behind the mapping + agreement
                                    on success); the driver's N = 1, 2, 4, 8 runs give the scaling curve.
                                    At N = 1 also: shard_proxy (rank 0's share of an N-rank run, alone
                                    on this GPU; its `model` keys are NOT measured) and shared_gpu (the
                                    N-rank code executed by N = 2, 4 real ranks on this one GPU over a
                                    gloo group: correctness + host-side contention, not scaling)
At N > 1 rank 0 prints the headline line BEFORE these sections start (marked `extras: pending`) and the complete
line after them; a section that hangs is abandoned after PM_BENCH_EXTRAS_TIMEOUT_S (300 s) with the complete
line printed (the section marked `timed out`) and exit status 75 - the headline stands, the status says a section was lost.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload frame|saturn|all26|cube|cube-host]
`--gpus N` with N > 1 launches itself: unless it already runs under torch.distributed.run
(WORLD_SIZE set), the process - before touching the GPU - starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` on this
file as a child, relays its output and exits with its status.
`--rehearse` runs launcher, process group (gloo), barriers, collectives and reporting on CPU with an
empty step (no engine, no numbers): the CPU test of the N > 1 plumbing.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HEADLINE = ['LON-GRAPHIC', 'LAT-GRAPHIC', 'PHASE', 'INCIDENCE', 'EMISSION']
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector FP64
EXTRAS_TIMED_OUT_STATUS = 75  # EX_TEMPFAIL: an optional N > 1 section was abandoned at its deadline; the headline line is complete
METRIC = 'Mpix/s full backplane set (lat/lon/inc/emi/phase) + map-reproject, 4096^2 frame'


# ------------------------------------------------------------------ launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--preheat-steps', type=int, default=500, help='untimed clock-ramp steps before the warmup steps')
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='headline line only: no host_path / cube_host sections')
    ap.add_argument(
        '--workload', default='frame', choices=['frame', 'saturn', 'all26', 'maps', 'cube', 'cube-host'],
        help='frame: BASELINE headline (default); saturn: config 4 (Saturn + rings, 8 planes); all26: every default '
        'backplane of a 4096^2 frame (what save_observation asks for); maps: all 26 map-space planes of a 0.1 deg '
        'rectangular grid (what save_mapped_observation(degree_interval=0.1) asks of the map chain); cube: config 5 with '
        'the cube resident in HBM; cube-host: config 5 fed from host memory (planes sharded over the GPUs)',
    )  # fmt: skip
    ap.add_argument('--planes', type=int, default=512, help='cube workloads: total planes')
    ap.add_argument('--rehearse', action='store_true', help='CPU / gloo rehearsal of the N > 1 plumbing (no GPU work)')
    ap.add_argument('--shared-gpu', action='store_true',
                    help='N > 1 ranks that all use GPU 0, process group over gloo (RCCL refuses two ranks on one card): the '
                    'N-rank code of the cube-host workload on a one-GPU box - one PCIe link and one CPU quota shared by '
                    'the ranks, so the times measure host-side contention, not scaling')
    ap.add_argument('--no-shared-gpu', action='store_true', help='default run at N = 1: skip the cube_host.shared_gpu section')
    ap.add_argument('--gather-mapped', action='store_true',
                    help='headline at N > 1: also all-gather the N mapped planes every step (the frames are '
                         'independent, so by default nothing is exchanged; the sharded-cube case with its RCCL '
                         'all-gather is the cube_host section / --workload cube-host)')
    ap.add_argument('--side-stream', action='store_true',
                    help='frame workload: pm_mapped_data (and the all-gather) on a second engine context bound to a '
                    'side stream, next to the frame kernel (+1.7 % Mpix/s, but the frame kernel itself is timed 2.5 % '
                    'slower while it shares the chip: default off so that roofline.frac is the kernel alone)')
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """
    `python bench.py --gpus N` typed directly: one child per GPU through torch.distributed.run.
    Runs BEFORE this process imports torch or touches HIP (a process that has initialised the GPU
    must not exec / fork workers), relays the children's output and returns their exit status.
    """
    cmd = [
        sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
        '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__),
    ] + sys.argv[1:]  # fmt: skip
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault('OMP_NUM_THREADS', '1')
    # torch.distributed.run reports ANY non-zero exit of a rank as its own status 1: a rank that leaves with a status that
    # means something (EXTRAS_TIMED_OUT_STATUS) also notes it in this file, and the launcher's 1 is translated back
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        env['PM_BENCH_STATUS_FILE'] = os.path.join(tmp, 'status')
        rc = subprocess.run(cmd, env=env).returncode
        if rc != 0 and os.path.exists(env['PM_BENCH_STATUS_FILE']):
            try:
                rc = int(open(env['PM_BENCH_STATUS_FILE']).read().strip() or rc)
            except ValueError:
                pass
    return rc


# ------------------------------------------------------------------ helpers
def rectangular_grid(west_positive: bool, degree_interval: float = 1.0):
    """BodyXY.generate_map_coordinates('rectangular') body_xy.py:2899-2907."""
    lons = np.arange(degree_interval / 2, 360, degree_interval)
    if west_positive:
        lons = lons[::-1]
    lats = np.arange(-90 + degree_interval / 2, 90, degree_interval)
    lon, lat = np.meshgrid(lons, lats)
    return np.ascontiguousarray(lon % 360), np.ascontiguousarray(lat)


def library_sha256() -> str:
    import hashlib

    from planetmapper_amd import _lib

    with open(_lib.LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def profile_record(kernel_prefix, workload: str = 'frame') -> tuple[dict, bool]:
    """
    Per-launch PMC figures of a kernel (or, for a tuple of prefixes, the sum over the kernels of a step) from
    the separate rocprofv3 --pmc passes of tools/pmc_profile.sh: profiles/traffic.json for the headline frame,
    profiles/traffic_<workload>.json for the others (WRITE_SIZE + FETCH_SIZE bytes, FP64 operations).
    PMC counters cannot be collected inside this process, so the last profiled values are reported -
    but only while they belong to THIS build: the file carries the sha256 of the library it was measured
    on. Returns (record, stale): stale = the file exists but was measured on another build (record empty).
    """
    name = 'traffic.json' if workload == 'frame' else f'traffic_{workload}.json'
    try:
        with open(os.path.join(REPO, 'profiles', name)) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}, False
    if t.get('_library_sha256') != library_sha256():
        return {}, True
    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    total: dict = {}
    for pre in prefixes:
        for k, v in t.items():
            if k.startswith(pre) and isinstance(v, dict):
                for field, val in v.items():
                    total[field] = total.get(field, 0.0) + val
                break
        else:
            return {}, False  # a kernel of the step is missing from the file
    return total, False


def rocprof_fields(rec: dict, alg: int) -> dict:
    """
    The profiler's own figure beside the events': the average duration of the step's kernels in the `rocprofv3 --kernel-trace
    --stats` pass of tools/pmc_profile.sh on THIS build (stamped like the traffic), and the fraction it gives. `kernel_ms` /
    `frac` are this run's HIP events on this box; boxes of the pool differ by +-10 %, store-bound kernels by more.
    """
    if 'rocprof_avg_ns' not in rec:
        return {}
    ms = rec['rocprof_avg_ns'] * 1e-6
    return {'kernel_ms_rocprof': round(ms, 4), 'frac_rocprof': round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}


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


def cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(g, sz: int, threads: int, budget_s: float, max_frames: int) -> dict:
    """
    The CPU oracle (a port: the reference's own Python + CSPICE path cannot run here) timed on this
    box's host cores on the same 5-plane frame + reprojection, OpenMP over rows on `threads` threads.
    """
    from oracle import oracle

    oracle.set_num_threads(threads)
    x0 = y0 = (sz - 1) / 2
    disc = oracle.make_disc(x0, y0, 0.9 * x0, 0.0, sz, sz)
    lon, lat = rectangular_grid(bool(g.west_positive))
    img = np.zeros((sz, sz))
    t0 = time.perf_counter()
    frames = 0
    while True:  # bounded sample: whole frames until the budget is used (at least one)
        oracle.backplanes_img(g, disc, HEADLINE)
        xm, ym = oracle.xy_map(g, disc, lon, lat)
        oracle.map_cube(img, xm, ym, 'linear', True)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= max_frames:
            break
    return {
        'value': round(frames * sz * sz / dt / 1e6, 3),
        'unit': 'Mpix/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'{frames} full {sz}x{sz} frame(s), 5 planes + 1 deg reprojection, OpenMP over rows',
    }


def cpu_baseline_other(workload: str, g, sz: int, names, threads: int, planes: int, budget_s: float = 10.0) -> dict:
    """
    The CPU oracle on a bounded sample of a secondary workload (same unit as its `value`): saturn / all26 - whole
    frames with the workload's plane set until the budget is used (at least one); cube - the x/y map of the 1 deg
    grid + as many planes of a 1024^2 f64 cube as fit the budget (planes are independent units).
    """
    from oracle import oracle

    oracle.set_num_threads(threads)
    x0 = (sz - 1) / 2
    t0 = time.perf_counter()
    if workload == 'cube':
        disc = oracle.make_disc(x0, x0, 0.9 * x0, 0.0, sz, sz)
        lon, lat = rectangular_grid(bool(g.west_positive))
        rng = np.random.default_rng(5)
        block = rng.standard_normal((16, sz, sz))
        xm, ym = oracle.xy_map(g, disc, lon, lat)
        done = 0
        while True:
            oracle.map_cube(block, xm, ym, 'linear', True)
            done += block.shape[0]
            dt = time.perf_counter() - t0
            if dt > budget_s or done >= planes:
                break
        return {'value': round(done * sz * sz / dt / 1e6, 1), 'unit': 'Mpix/s', 'cores': threads, 'kind': 'port',
                'sample': f'x/y map of the 1 deg grid + {done} of the {planes} planes ({sz}x{sz} f64), bilinear, OpenMP over map rows'}
    if workload == 'maps':
        disc = oracle.make_disc(x0, x0, 0.9 * x0, 0.0, sz, sz)
        lon, lat = rectangular_grid(bool(g.west_positive), 0.1)
        rows = 0
        while True:  # (map rows are independent units: blocks of 225 rows = 810 000 cells)
            oracle.backplanes_map(g, disc, list(names), lon[rows : rows + 225], lat[rows : rows + 225])
            rows += 225
            dt = time.perf_counter() - t0
            if dt > budget_s or rows >= lon.shape[0]:
                break
        return {'value': round(rows * lon.shape[1] / dt / 1e6, 3), 'unit': 'Mcell/s', 'cores': threads, 'kind': 'port',
                'sample': f'{rows} of the {lon.shape[0]} rows of the 0.1 deg grid ({lon.shape[1]} cells each), {len(list(names))} planes, OpenMP over cells'}
    r0, rot = (800.0 * sz / 4096, 20.0) if workload == 'saturn' else (0.9 * x0, 0.0)
    disc = oracle.make_disc(x0, x0, r0, rot, sz, sz)
    frames = 0
    while True:
        oracle.backplanes_img(g, disc, list(names))
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= 4:
            break
    return {'value': round(frames * sz * sz / dt / 1e6, 3), 'unit': 'Mpix/s', 'cores': threads, 'kind': 'port',
            'sample': f'{frames} full {sz}x{sz} frame(s), {len(list(names))} planes, OpenMP over rows'}


class Dist:
    """The process group of this run (or a single process): barrier, max-over-ranks, cleanup."""

    def __init__(self, args):
        self.rank = int(os.environ.get('RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.rehearse = args.rehearse
        if self.world != args.gpus:
            raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE is {self.world}')
        import torch

        self.torch = torch
        self.shared_gpu = bool(getattr(args, 'shared_gpu', False))
        if self.shared_gpu:
            self.local_rank = 0  # every rank on the one card
        if self.rehearse:
            self.dev = torch.device('cpu')
        else:
            if not torch.cuda.is_available():
                raise SystemExit('bench.py needs a GPU (the engine has no CPU fallback)')
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device('cuda', self.local_rank)
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.rehearse or self.shared_gpu:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dist = dist

    def sync(self) -> None:
        if not self.rehearse:
            self.torch.cuda.synchronize()

    def barrier(self) -> None:
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, gathered, mine, async_op=False):
        if self.world == 1:
            return None
        return self.dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), async_op=async_op)

    def close(self) -> None:
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_steps(d: Dist, step, args, preheat: int):
    """
    The contract's timed region: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier
    + device synchronisation on both sides; the time is the maximum over the ranks.
    Before that an untimed clock ramp: a fresh box starts in a low power state and the shader clock
    takes tens of milliseconds of sustained load to settle (a fixed count, not a wall-clock loop:
    every rank must issue the same collectives).
    """
    for _ in range(preheat):
        step(None)
    d.barrier()
    for _ in range(args.warmup):
        step(None)
    d.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    d.barrier()
    return d.max_over_ranks(time.perf_counter() - t0)


# ------------------------------------------------------------------ rehearsal (CPU, gloo)
def rehearse(args) -> None:
    d = Dist(args)
    torch = d.torch
    n0, n1 = 180, 360
    gathered = torch.zeros((d.world, 1, n0, n1), dtype=torch.float64)
    fail_rank = int(os.environ.get('PM_BENCH_FAIL_RANK', '-1'))

    def step(i):
        if i is not None and d.rank == fail_rank:
            raise RuntimeError('rehearsal: injected failure on this rank')
        gathered[d.rank].fill_(float(d.rank + 1))
        d.all_gather(gathered, gathered[d.rank])

    dt = timed_steps(d, step, args, preheat=2)
    assert all(float(gathered[r, 0, 0, 0]) == r + 1 for r in range(d.world))
    if d.rank == 0:
        print(json.dumps({
            'metric': METRIC, 'value': 0.0, 'unit': 'Mpix/s', 'n_gpus': d.world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'none', 'rehearsal': True,
            'config': {'workload': 'CPU / gloo rehearsal of launcher, barriers, all-gather and reporting: no engine work'},
        }), flush=True)  # fmt: skip
    d.close()


# ------------------------------------------------------------------ config 5 fed from host memory
def cube_host_section(d: Dist, eng, g, planes: int, steps_fed: int, steps_resident: int, sz: int = 1024,
                      degree_interval: float = 1.0, partial: dict | None = None) -> dict:
    """
    BASELINE config 5 as the reference runs it (observation.py:876-905 maps a HOST cube): P x 1024^2
    f64 planes in host memory, contiguous blocks of ceil(P / N) planes per rank
    (`distributed.shard_bounds`), each rank feeding its block from its own pinned host buffer over
    its own PCIe link (`PM_MEM_HOST_CUBE`, by the route the library measured fastest on that rank), the
    mapped planes all-gathered exchange by exchange behind the mapping, a closing agreement on success
    (`distributed.map_cube_sharded_pipelined`). Returns step times with the host feed and with the block
    already resident in HBM (mean of the timed region, max over ranks; rank 0's own per-step median /
    min / max beside it) and, at N = 1, `shard_proxy`: rank 0's step of a 2 / 4 / 8-rank run measured on
    this GPU, with the scaling it predicts.
    """
    torch = d.torch
    from planetmapper_amd.distributed import shard_bounds

    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    a, b, per_rank = shard_bounds(planes, d.world, d.rank)
    mine_n = b - a
    lon_h, lat_h = rectangular_grid(bool(g.west_positive), degree_interval)
    n0, n1 = lon_h.shape
    lon_d, lat_d = torch.from_numpy(lon_h).to(d.dev), torch.from_numpy(lat_h).to(d.dev)
    xm = torch.empty((n0, n1), dtype=torch.float64, device=d.dev)
    ym = torch.empty((n0, n1), dtype=torch.float64, device=d.dev)
    gen = torch.Generator(device=d.dev).manual_seed(5 + d.rank)
    cube_d = torch.randn((max(mine_n, 1), sz, sz), generator=gen, device=d.dev, dtype=torch.float64)
    cube_d[torch.rand(cube_d.shape, generator=gen, device=d.dev) < 1e-3] = float('nan')
    t = time.perf_counter()
    cube_h = eng.pinned_empty((max(mine_n, 1), sz, sz))  # what a loader reads this rank's planes into
    t_pin = time.perf_counter() - t
    torch.from_numpy(cube_h).copy_(cube_d)
    gathered = torch.full((d.world, per_rank, n0, n1), float('nan'), dtype=torch.float64, device=d.dev)

    from planetmapper_amd.distributed import agree_on_success, exchange_planes, map_cube_sharded_pipelined

    slot = gathered[d.rank]

    def step(fed: bool, pipelined: bool = True, stages: dict | None = None):
        # the x/y map of the grid, then this rank's planes
        eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
        if pipelined:
            # exchange by exchange: each one mapped and its all-gather started while the next is collected /
            # copied / mapped; a closing agreement on success (distributed.map_cube_sharded_pipelined)
            map_cube_sharded_pipelined(eng, cube_h if fed else cube_d, np.float64, planes, xm, ym, n0, n1, gathered, d.rank, d.world,
                                       host_cube=fed, stages=stages)
            return
        # the plain form: map the block, finish it, ONE all-gather of the slots (no callback, no overlap)
        if mine_n > 0:
            if fed:
                eng.map_cube_host_to_device(cube_h[:mine_n], xm, ym, n0, n1, slot[:mine_n])
            else:
                eng.map_cube_device(cube_d, np.float64, mine_n, xm, ym, n0, n1, slot[:mine_n])
        eng.synchronize()
        d.all_gather(gathered, slot)

    def run(fed: bool, steps: int, pipelined: bool = True):
        # (untimed: for a host-fed cube the call that probes the routes and, where a hybrid is a candidate, the six
        #  whole calls of its trial - the timed steps run on the route the library has committed to)
        for _ in range(8 if fed else 2):
            step(fed, pipelined)
        d.barrier()
        own = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t = time.perf_counter()
            step(fed, pipelined)
            d.sync()
            own.append(time.perf_counter() - t)
        d.barrier()
        return d.max_over_ranks(time.perf_counter() - t0) / steps, own

    def spread(ts) -> dict:
        return {'median': round(float(np.median(ts)) * 1e3, 3), 'min': round(min(ts) * 1e3, 3), 'max': round(max(ts) * 1e3, 3),
                'reps': len(ts)}

    from planetmapper_amd import _lib

    real = hasattr(eng, 'set_option')  # (the CPU rehearsal's engine double has no options)
    pix = planes * sz * sz
    block_bytes = mine_n * sz * sz * 8
    sec = {
        'workload': f'host-resident IFU cube {planes}x{sz}x{sz} f64 -> 1 deg map ({n0}x{n1}), bilinear, '
        f'{per_rank} planes per rank from pinned host memory, pipelined all-gather of the mapped planes '
        f'({exchange_planes(per_rank, n0, n1)} planes per exchange) + agreement on success'
        + ('' if d.world > 1 else ' (no collective at N=1)'),
        'collective_backend': None if d.world == 1 else ('gloo (host-staged; every rank on GPU 0)' if d.shared_gpu else 'nccl (RCCL)'),
        'rccl_ranks': 0 if d.shared_gpu else d.world,
        'ranks': d.world,
        'planes': planes,
        'planes_per_rank': per_rank,
        'all_gather_bytes_per_rank': per_rank * n0 * n1 * 8,
        'pinned_alloc_ms': round(t_pin * 1e3, 1),
        'scaling': 'strong',
    }
    equal = lambda: bool(torch.equal(torch.nan_to_num(ref, nan=-1.0), torch.nan_to_num(gathered, nan=-1.0)))  # noqa: E731
    if d.world > 1:
        # The plain form first: its pieces (one engine call, one all-gather) are what every other N > 1 code of
        # this repo already does - whatever the pipelined form below does on this node, these numbers stand.
        t_res_plain, own = run(False, steps_resident, pipelined=False)
        ref = gathered.clone()
        t_fed_plain, own_fed_plain = run(True, steps_fed, pipelined=False)
        sec.update({
            'ms_per_step_host_fed_plain_allgather': round(t_fed_plain * 1e3, 3),
            'ms_per_step_resident_plain_allgather': round(t_res_plain * 1e3, 4),
            'rank0_step_ms_host_fed_plain_allgather': spread(own_fed_plain),
            'Mpix_s_host_fed_plain_allgather': round(pix / t_fed_plain / 1e6, 1),
            'fed_equals_resident_plain_allgather': equal(),
        })
        if partial is not None:
            # (what stands if the pipelined protocol below never comes back: the watchdog of the caller prints this)
            partial.update(sec, ms_per_step_host_fed=sec['ms_per_step_host_fed_plain_allgather'], Mpix_s_host_fed=sec['Mpix_s_host_fed_plain_allgather'])
    # the pipelined protocol. A failure on any rank is agreed on (every rank raises or none does), so the ranks
    # stay in step for whatever follows; the caller records it and keeps the headline.
    error = None
    try:
        t_res, own_res = run(False, steps_resident)
        if d.world == 1:
            ref = gathered.clone()
        same = equal()
        t_fed, own_fed = run(True, steps_fed)
        same = same and equal()
    except Exception as e:  # noqa: BLE001
        error = e
    if d.world > 1:
        try:
            agree_on_success(error)
        except Exception as e:  # noqa: BLE001
            sec['pipelined_error'] = f'{type(e).__name__}: {e}'[:500]
            if 'ms_per_step_host_fed_plain_allgather' in sec:
                sec['ms_per_step_host_fed'] = sec['ms_per_step_host_fed_plain_allgather']
                sec['Mpix_s_host_fed'] = sec['Mpix_s_host_fed_plain_allgather']
            return sec
    elif error is not None:
        raise error
    if d.world > 1:
        # every rank feeding and mapping its block at the same time, nothing exchanged: the per-rank host leg alone
        def nocoll():
            eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
            map_cube_sharded_pipelined(eng, cube_h, np.float64, planes, xm, ym, n0, n1, gathered, d.rank, d.world, host_cube=True,
                                       gather=False)
            d.sync()

        nocoll()
        d.barrier()
        ts = []
        c0, w0 = time.process_time(), time.perf_counter()
        for _ in range(steps_fed):
            t = time.perf_counter()
            nocoll()
            ts.append(time.perf_counter() - t)
        busy = (time.process_time() - c0) / max(time.perf_counter() - w0, 1e-9)  # CPUs this process kept busy (all its threads)
        d.barrier()
        sec['ms_per_step_host_fed_no_collective'] = round(d.max_over_ranks(float(np.median(ts))) * 1e3, 3)
        sec['rank0_step_ms_host_fed_no_collective'] = spread(ts)
        sec['cpus_busy_per_rank_no_collective'] = round(d.max_over_ranks(busy), 2)
        sec['cpus_available'] = host_cores()
    route = eng.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE) if real else None
    route_ns = {str(r): eng.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(_lib.NUM_CUBE_ROUTES)} if real else None
    table_hits = eng.get_option(_lib.PM_OPT_BLOCK_TABLE_HITS) if real else None
    # the same step with the GPU fetching its blocks itself over its own PCIe link (route 2: no copy
    # threads; slower on one GPU, but the leg that is private to each rank when N grows - the copy
    # threads of the default route share the host's memory system)
    t_fetch = own_fetch = None
    if real and not d.shared_gpu:
        eng.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, 2)
        try:
            t_fetch, own_fetch = run(True, max(3, steps_fed // 2))
            same = same and equal()
        finally:
            eng.set_option(_lib.PM_OPT_HOST_CUBE_ROUTE, -1)
    sec.update({
        'ms_per_step_host_fed': round(t_fed * 1e3, 3),
        'ms_per_step_resident': round(t_res * 1e3, 4),
        'ms_per_step_host_fed_gpu_fetch': None if t_fetch is None else round(t_fetch * 1e3, 3),
        'rank0_step_ms_host_fed': spread(own_fed),
        'rank0_step_ms_resident': spread(own_res),
        'rank0_step_ms_host_fed_gpu_fetch': None if own_fetch is None else spread(own_fetch),
        'Mpix_s_host_fed': round(pix / t_fed / 1e6, 1),
        'Mpix_s_resident': round(pix / t_res / 1e6, 1),
        'host_feed_GBps_per_rank': round(block_bytes / t_fed / 1e9, 2),
        'host_feed': 'the route the library measured fastest on this rank (PM_OPT_HOST_CUBE_ROUTE -1: one chunk '
        'through each candidate on the first call, then the fastest): 3 = the 16-byte blocks of each plane that the map '
        'samples, collected by the copy threads into pinned staging and sent by DMA; 0 = whole planes by DMA; 2 = 128-byte '
        'blocks fetched by the GPU itself; 4 = hybrid, chunks alternately collected by the threads and fetched by the GPU; '
        'block table cached across calls by map fingerprint',
        'route_chosen': route,
        'route_ns_per_plane': route_ns,
        'copy_threads': eng.get_option(_lib.PM_OPT_HOST_COPY_THREADS_IN_USE) if real else None,
        'block_table_cache_hits': table_hits,
        'fed_equals_resident': same,
    })
    if real:
        sec['stages'] = stage_breakdown(eng, lambda st: step(True, True, st), d.sync, d)
    if real and d.world == 1 and mine_n >= 16:
        sec['shard_proxy'] = shard_proxy(eng, step_fed_n=lambda n_local, n_total, stages=None: _proxy_step(
            eng, lon_d, lat_d, n0, n1, xm, ym, cube_h, n_local, n_total, gathered, stages), planes=planes, n0=n0, n1=n1, t_n1=t_fed,
            t_n1_median=float(np.median(own_fed)))
    return sec


def stage_breakdown(eng, step_with_stages, sync, d=None, reps: int = 5) -> dict:
    """
    Where a host-fed step of this rank spends its time: `reps` extra steps (after the timed ones) with PM_OPT_TRACE bit 4
    on - the library's own stage record of the mapping call (PM_OPT_LAST_STAGE_NS: block-table look-up / build, plan,
    the first chunk's collection with the link idle, the calling thread's time inside the collections, issue of all
    chunks, drain of the last DMA and kernels, flag check; device-side sums of the H2D copies and of the kernels) and,
    around it, what `map_cube_sharded_pipelined` reports (the exchanges still in flight once the mapping is done, the
    agreement). Medians in ms. `accounted_ms` = the sequential parts (x / y map + sync, tables + plan + issue + drain +
    finish, exposed exchange, agreement): the rest of `step_ms` is Python and launch overhead.
    """
    from planetmapper_amd import _lib

    old = eng.get_option(_lib.PM_OPT_TRACE)
    eng.set_option(_lib.PM_OPT_TRACE, old | 4)
    rows = []
    try:
        for _ in range(reps):
            if d is not None:
                d.barrier()
            st: dict = {}
            t = time.perf_counter()
            step_with_stages(st)
            sync()
            st['step'] = (time.perf_counter() - t) * 1e3
            st.update({'engine_' + k: v for k, v in eng.last_stages_ms().items() if not k.startswith(('exchange', 'agreement', 'sharded'))})
            rows.append(st)
    finally:
        eng.set_option(_lib.PM_OPT_TRACE, old)
    med = {k: round(float(np.median([r.get(k, 0.0) for r in rows])), 4) for k in rows[0]}
    accounted = (med['engine_tables'] + med['engine_plan'] + med['engine_issue'] + med['engine_drain'] + med['engine_finish']
                 + med.get('exchange_exposed', 0.0) + med.get('agreement', 0.0))
    return {
        'ms': {
            'step': med['step'],
            'block_table_lookup_or_build': med['engine_tables'],
            'plan': med['engine_plan'],
            'first_chunk_fill_link_idle': med['engine_first_fill'],
            'collection_calling_thread': med['engine_collect'],
            'issue_all_chunks': med['engine_issue'],
            'drain_last_dma_and_kernels': med['engine_drain'],
            'flag_check_and_replay': med['engine_finish'],
            'dma_device_sum': med['engine_dma_device'],
            'kernels_device_sum': med['engine_kernels_device'],
            'mapping_call_total': med['engine_total'],
            'map_call_python': med.get('map_call', 0.0),
            'exchange_exposed_after_last_kernel': med.get('exchange_exposed', 0.0),
            'agreement': med.get('agreement', 0.0),
        },
        'accounted_ms': round(accounted, 4),
        'unaccounted_ms': round(med['step'] - accounted, 4),
        'reps': reps,
        'note': 'issue_all_chunks contains first_chunk_fill and collection; dma / kernels are device-clock sums that overlap the host '
        'stages; measured in extra steps with PM_OPT_TRACE bit 4 (which makes the call wait for its exchanges before the agreement)',
    }


_PROXY_SLOTS: dict = {}


def _proxy_step(eng, lon_d, lat_d, n0, n1, xm, ym, cube_h, n_local: int, n_total: int, gathered, stages: dict | None = None) -> None:
    """rank 0's step of an N-rank run on this one GPU: its block, exchange by exchange, no collective"""
    from planetmapper_amd.distributed import map_cube_sharded_pipelined

    world = -(-n_total // n_local)
    eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
    # (world * n_local >= n_total planes - more than the N = 1 buffer holds when N does not divide the cube: the
    #  proxy's slots are its own, kept between steps)
    key = (world, n_local, n0, n1)
    if _PROXY_SLOTS.get('key') != key:
        _PROXY_SLOTS['key'] = key
        _PROXY_SLOTS['slots'] = gathered.new_empty((world, n_local, n0, n1))
    slots = _PROXY_SLOTS['slots']
    map_cube_sharded_pipelined(eng, cube_h[:n_local], np.float64, n_total, xm, ym, n0, n1, slots, 0, world, host_cube=True,
                               pipeline_chunks=True, stages=stages)


def shard_proxy(eng, step_fed_n, planes: int, n0: int, n1: int, t_n1: float, t_n1_median: float) -> dict:
    """
    What the sharded host-fed cube would cost rank 0 of an N-GPU run, measured on THIS box: its block of
    ceil(P / N) planes, cut into the exchanges of the real protocol, fed with the copy threads such a rank
    gets - (a) cores / N, the library's own rule under a launcher (`LOCAL_WORLD_SIZE`) when the CPU quota
    of the job does NOT grow with the GPUs, and (b) all of this box's cores, when every rank brings its
    own quota. Each case lets the library measure its routes afresh (two warm-up steps), then times 7
    steps. The collective cannot be measured on one GPU: its EXPOSED part - the last exchange only, the
    others travel behind the mapping - is priced at an ASSUMED 50 GB/s per xGMI link (a third of the
    153 GB/s link figure of MI355X_MICROARCH.md) + 30 us. Not shared in this proxy: the host's memory
    system, which N ranks collecting at once would share - that is what the real N = 2, 4, 8 runs add.
    """
    from planetmapper_amd import _lib
    from planetmapper_amd.distributed import exchange_planes

    cores = host_cores()
    out = {'cores_this_box': cores, 'n1_ms_mean': round(t_n1 * 1e3, 3), 'n1_ms_median': round(t_n1_median * 1e3, 3),
           'assumed_xgmi_GBps_per_link': 50.0, 'cases': []}
    try:
        for n in (2, 4, 8):
            per = -(-planes // n)
            for label, threads in (('cores/N', max(2, cores // n)), ('cores', min(16, cores))):
                eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, threads)
                eng.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)  # forget what another thread count measured
                # (untimed: the call that probes the routes, then - where the probes make a hybrid a candidate - three
                #  whole calls each by the best single route and by the hybrid, after which the library has committed)
                for _ in range(8):
                    step_fed_n(per, planes)
                ts = []
                for _ in range(7):
                    t = time.perf_counter()
                    step_fed_n(per, planes)
                    eng.synchronize()
                    ts.append(time.perf_counter() - t)
                k = exchange_planes(per, n0, n1)
                exposed = k * n0 * n1 * 8 / 50e9 + 30e-6
                t_med = float(np.median(ts))
                out['cases'].append({
                    'N': n, 'planes_per_rank': per, 'copy_threads': threads, 'threads_rule': label,
                    'route_chosen': eng.get_option(_lib.PM_OPT_LAST_CUBE_ROUTE),
                    'route_ns_per_plane': {str(r): eng.get_option(_lib.PM_OPT_ROUTE_NS_PER_PLANE + r) for r in range(_lib.NUM_CUBE_ROUTES)},
                    'rank0_ms': {'median': round(t_med * 1e3, 3), 'min': round(min(ts) * 1e3, 3), 'max': round(max(ts) * 1e3, 3)},
                    'stages': stage_breakdown(eng, lambda st: step_fed_n(per, planes, st), eng.synchronize),
                    # NOT measured: a model on top of the measured rank-0 time (one assumed link rate, no host contention)
                    'model': {
                        'exposed_allgather_ms_assumed': round(exposed * 1e3, 3),
                        'step_ms': round((t_med + exposed) * 1e3, 3),
                        'scaling_vs_n1': round(t_n1_median / (t_med + exposed), 2),
                    },
                })
    finally:
        eng.set_option(_lib.PM_OPT_HOST_COPY_THREADS, 0)
        eng.set_option(_lib.PM_OPT_ROUTE_EXPLORE, 1)
    return out


def host_path_section(eng, g, sz: int) -> dict:
    """PCIe-inclusive time of the headline frame into numpy arrays (what get_*_img() callers see)."""
    import ctypes

    from planetmapper_amd import _lib
    from planetmapper_amd.engine import PLANE_INDEX, plane_mask

    x0 = (sz - 1) / 2
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    nbytes = len(HEADLINE) * sz * sz * 8
    eng.backplanes_img(HEADLINE)

    def timed(fn, reps=7):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t)
            del r
        return ts

    def ms(ts) -> dict:
        return {'median': round(float(np.median(ts)) * 1e3, 2), 'min': round(min(ts) * 1e3, 2), 'max': round(max(ts) * 1e3, 2),
                'reps': len(ts)}

    t_fresh = timed(lambda: eng.backplanes_img(HEADLINE))
    pinned = {n: eng.pinned_empty((sz, sz)) for n in HEADLINE}
    ptrs = (ctypes.c_void_p * _lib.NUM_PLANES)()
    for n, arr in pinned.items():
        ptrs[PLANE_INDEX[n]] = arr.ctypes.data

    def into_pinned():
        eng._check(eng._lib.pm_backplanes_img(eng._ctx, plane_mask(HEADLINE), 0.0, ptrs, _lib.PM_MEM_HOST))

    t_pin = timed(into_pinned)
    eng.set_option(_lib.PM_OPT_SPARSE_FRAME, 0)
    try:
        t_pin_whole = timed(into_pinned, 5)
        t_fresh_whole = timed(lambda: eng.backplanes_img(HEADLINE), 5)
    finally:
        eng.set_option(_lib.PM_OPT_SPARSE_FRAME, -1)
    med = lambda ts: float(np.median(ts))  # noqa: E731
    return {
        'frame': f'{sz}x{sz} x {len(HEADLINE)} planes = {nbytes / 1e6:.0f} MB to host',
        'ms_fresh_numpy_arrays': ms(t_fresh),
        'GBps_fresh_numpy_arrays_median': round(nbytes / med(t_fresh) / 1e9, 1),
        'ms_pinned_arrays': ms(t_pin),
        'GBps_pinned_arrays_median': round(nbytes / med(t_pin) / 1e9, 1),
        'pcie_gen5_x16_spec_GBps': 63.0,
        'Mpix_s_fresh_numpy_arrays_median': round(sz * sz / med(t_fresh) / 1e6, 1),
        'ms_fresh_numpy_arrays_whole_planes': ms(t_fresh_whole),
        'ms_pinned_arrays_whole_planes': ms(t_pin_whole),
        'note': 'default transfer: only the spans of rows inside the radius pre-mask circle cross PCIe, the copy threads write '
        'the NaN outside them; the GB/s figures count the bytes DELIVERED, so they can exceed the link rate; '
        '*_whole_planes = PM_OPT_SPARSE_FRAME 0',
    }


def api_path_section(g, sz: int, device: int) -> dict:
    """
    The drop-in surface itself, timed: what a user of the reference types -
    `BodyXY.get_{lon,lat,phase_angle,incidence_angle,emission_angle}_img()` (body_xy.py:2586-2630 and the getters
    built on `_get_lonlat_img` / `_get_illumination_gie_img`, :3281, :3658) and
    `Observation.get_mapped_data(degree_interval=1)` (observation.py:826-872) - through the Python shim
    (planetmapper_amd.BodyXY / Observation -> Engine -> ctypes -> C ABI), cold cache every repetition.
    The five getters are two families, hence two `pm_backplanes_img` calls (2 + 3 planes), into result arrays the
    engine recycles between discs (page-locked, their pages long faulted in); `host_path.ms_fresh_numpy_arrays` is the
    same 671 MB through ONE Engine call into new numpy arrays (a page fault per 4 KiB: 6 of its 16 ms).
    """
    from planetmapper_amd import Observation

    rng = np.random.default_rng(7)
    data = rng.standard_normal((1, sz, sz))
    obs = Observation(data=data, geometry=g, device=device)
    x0 = (sz - 1) / 2
    getters = ('get_lon_img', 'get_lat_img', 'get_phase_angle_img', 'get_incidence_angle_img', 'get_emission_angle_img')

    def once():
        obs.set_disc_params(x0, x0, 0.9 * x0, 0.0)  # (clears every cache, like the reference: body_xy.py:696-698)
        t0 = time.perf_counter()
        planes = [getattr(obs, name)() for name in getters]
        t1 = time.perf_counter()
        mapped = obs.get_mapped_data(degree_interval=1)
        t2 = time.perf_counter()
        assert all(p.shape == (sz, sz) for p in planes) and mapped.shape == (1, 180, 360)
        return t1 - t0, t2 - t1

    once()
    reps = [once() for _ in range(7)]
    imgs, maps = [r[0] for r in reps], [r[1] for r in reps]
    tot = [a + b for a, b in reps]
    ms = lambda ts: {'median': round(float(np.median(ts)) * 1e3, 3), 'min': round(min(ts) * 1e3, 3), 'max': round(max(ts) * 1e3, 3),  # noqa: E731
                     'reps': len(ts)}

    # The same calls with `device=True`: the planes stay in HBM as DeviceArrays (`__dlpack__` / `__cuda_array_interface__`),
    # what a GPU-side consumer takes. Timed to the point where the results are complete (engine synchronised).
    def once_device():
        obs.set_disc_params(x0, x0, 0.9 * x0, 0.0)  # (cold cache: the previous handles are invalidated, their memory pooled)
        t0 = time.perf_counter()
        planes = [getattr(obs, name)(device=True) for name in getters]
        obs._engine.synchronize()
        t1 = time.perf_counter()
        mapped = obs.get_mapped_data(degree_interval=1, device=True)
        obs._engine.synchronize()
        t2 = time.perf_counter()
        assert all(p.shape == (sz, sz) and p.valid for p in planes) and mapped.shape == (1, 180, 360)
        return t1 - t0, t2 - t1, planes, mapped

    _, _, planes_d, mapped_d = once_device()
    host_planes = [getattr(obs, name)() for name in getters]
    same = all(np.array_equal(d.numpy(), h, equal_nan=True) for d, h in zip(planes_d, host_planes)) and np.array_equal(
        mapped_d.numpy(), obs.get_mapped_data(degree_interval=1), equal_nan=True)
    del planes_d, mapped_d, host_planes
    dreps = [once_device()[:2] for _ in range(7)]
    device_leg = {
        'calls': 'the same getters with device=True: results left in HBM as DeviceArray (DLPack / __cuda_array_interface__), engine synchronised',
        'ms_five_backplane_getters': ms([r[0] for r in dreps]),
        'ms_get_mapped_data': ms([r[1] for r in dreps]),
        'equal_to_the_numpy_getters': bool(same),
    }
    return {
        'device': device_leg,
        'calls': 'Observation(data=(1, %d, %d) f64).get_{lon,lat,phase_angle,incidence_angle,emission_angle}_img() + '
        'get_mapped_data(degree_interval=1), cold cache, through the Python shim' % (sz, sz),
        'ms': ms(tot),
        'ms_five_backplane_getters': ms(imgs),
        'ms_get_mapped_data': ms(maps),
        'Mpix_s_median': round(sz * sz / float(np.median(tot)) / 1e6, 1),
        'note': 'compare ms_five_backplane_getters with host_path.ms_fresh_numpy_arrays (the same planes through one '
        'Engine call into arrays whose pages do not exist yet) and ms_pinned_arrays: the getters write into the page-locked '
        'arrays the previous disc\'s planes lived in (Engine.plane_buffer: recycled when nobody holds them any more); '
        'get_mapped_data adds the host-resident data plane (134 MB, of which the sampled blocks cross PCIe)',
    }


def interpolations_section(device: int, g, planes: int = 64, sz: int = 1024) -> dict:
    """
    Every interpolation of BodyXY.map_img (body_xy.py:1414-1904) on a cube resident in HBM: microseconds per plane of
    `sz`^2 onto the 1 deg map (180 x 360) and onto a 0.1 deg map (1800 x 3600 = 6.48 M cells), `planes` planes per call,
    and the cubic of ONE plane (a batch of splines costs what its chain of dependent steps costs). Results stay on the
    device; the smoothing-spline search is left to tools/probes/smoothing_rate.py (seconds per call).
    """
    import torch

    from planetmapper_amd.engine import Engine

    x0 = (sz - 1) / 2
    eng = Engine(device)
    eng.set_geometry(g)
    eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
    gen = torch.Generator(device='cuda').manual_seed(5)
    cube = torch.randn((planes, sz, sz), generator=gen, device='cuda', dtype=torch.float64)
    torch.cuda.synchronize()
    out = {'workload': f'{planes} planes of {sz}x{sz} f64 resident in HBM, results left on the device', 'us_per_plane': {}}
    for deg in (1.0, 0.1):
        lon_g, lat_g = rectangular_grid(g.west_positive, deg)
        n0, n1 = lon_g.shape
        lon_d, lat_d = torch.from_numpy(lon_g).cuda(), torch.from_numpy(lat_g).cuda()
        xm = torch.empty((n0, n1), dtype=torch.float64, device='cuda')
        ym = torch.empty_like(xm)
        torch.cuda.synchronize()
        eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
        res = torch.empty((planes, n0, n1), dtype=torch.float64, device='cuda')
        row = {}
        for interp in ('nearest', 'linear', 'quadratic', 'cubic', 5, 'smooth'):
            if interp == 'smooth':
                eng.set_smooth_options(5, 10_000)
            for n_pl, key in ((planes, str(interp)),) + (((1, 'cubic, one plane'),) if interp == 'cubic' else ()):
                eng.map_cube_device(cube, np.float64, n_pl, xm, ym, n0, n1, res, interp, True)
                eng.synchronize()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.map_cube_device(cube, np.float64, n_pl, xm, ym, n0, n1, res, interp, True)
                eng.synchronize()
                row[key] = round((time.perf_counter() - t0) / reps / n_pl * 1e6, 1)
        out['us_per_plane'][f'{deg} deg map ({n0}x{n1})'] = row
        del res, xm, ym, lon_d, lat_d
    eng.close()
    del cube
    torch.cuda.empty_cache()
    return out


def shared_gpu_section(args) -> dict:
    """
    The N-rank code of the sharded host-fed cube, EXECUTED on this one GPU: for N = 2 and 4 this process starts
    `torch.distributed.run` with N fresh ranks of `bench.py --workload cube-host --shared-gpu` - every rank a real
    engine on GPU 0 with its own pinned block of ceil(P / N) planes, the process group over gloo (RCCL refuses two
    ranks on one card), `LOCAL_WORLD_SIZE` = N so that each rank takes cores / N copy threads by the library's own
    rule. All ranks collect at once: the host-memory and CPU-quota contention that `shard_proxy` (one rank alone)
    leaves out is in these times. What they are NOT: a scaling figure - the N ranks share ONE PCIe link (the bytes
    over it do not shrink with N) and the collective is gloo's host-staged one, not RCCL over xGMI.
    N = 8 is not run: the pool allows six processes on a card.
    """
    out = {'what': 'N real ranks on ONE GPU (gloo group, one PCIe link and one CPU quota shared): measured, not a scaling figure',
           'runs': []}
    for n in (2, 4):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), os.path.abspath(__file__), '--gpus', str(n), '--workload', 'cube-host',
               '--shared-gpu', '--planes', str(args.planes), '--steps', '7']
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '1')
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        except subprocess.TimeoutExpired:
            out['runs'].append({'N': n, 'error': 'timed out after 420 s'})
            break
        wall = time.perf_counter() - t0
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        if p.returncode != 0 or not lines:
            out['runs'].append({'N': n, 'error': f'rc {p.returncode}: ' + (p.stdout + p.stderr)[-600:]})
            break
        sec = json.loads(lines[-1]).get('cube_host', {})
        keep = ('ranks', 'collective_backend', 'planes_per_rank', 'copy_threads', 'cpus_busy_per_rank_no_collective', 'cpus_available',
                'route_chosen', 'route_ns_per_plane',
                'ms_per_step_host_fed_no_collective', 'rank0_step_ms_host_fed_no_collective',
                'ms_per_step_host_fed', 'rank0_step_ms_host_fed', 'ms_per_step_resident',
                'ms_per_step_host_fed_plain_allgather', 'fed_equals_resident', 'fed_equals_resident_plain_allgather', 'pipelined_error')
        run = {'N': n, 'wall_s': round(wall, 1)}
        run.update({k: sec[k] for k in keep if k in sec})
        out['runs'].append(run)
    return out


# ------------------------------------------------------------------ secondary workloads
def other_workloads(args) -> None:
    """Secondary BASELINE configs (4 and 5); same JSON contract, their own metric names."""
    d = Dist(args)
    torch = d.torch
    from planetmapper_amd.distributed import map_cube_sharded_device, shard_bounds
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    eng = Engine(d.local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.workload == 'cube-host':
        g = load_scenario('jupiter_hst_2005')
        sec = cube_host_section(d, eng, g, args.planes, max(3, min(args.steps, 20)), max(20, args.steps))
        if d.rank == 0:
            print(json.dumps({
                'metric': 'Mpix/s of cube pixels map-projected from HOST memory (get_mapped_data, 1 deg rectangular map, bilinear)',
                'value': sec['Mpix_s_host_fed'], 'unit': 'Mpix/s', 'n_gpus': d.world, 'steps': max(3, min(args.steps, 20)),
                'warmup': 2, 'ms_per_step': sec['ms_per_step_host_fed'], 'higher_is_better': True, 'scaling': 'strong',
                'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'config': {'workload': sec['workload']},
                'cube_host': sec,
            }), flush=True)  # fmt: skip
        d.close()
        eng.close()
        return

    if args.workload == 'all26':
        # what save_observation asks for (observation.py:1269-1279): all 26 default backplanes of one frame
        from planetmapper_amd._lib import PLANE_NAMES

        sz = args.size
        names = list(PLANE_NAMES)
        g = load_scenario('jupiter_hst_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
        planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=d.dev) for n in names}

        def work():
            eng.backplanes_img_device(planes)

        units = sz * sz
        metric = 'Mpix/s all 26 default backplanes (save_observation set), 4096^2 frame'
        workload = (f'Jupiter/HST 2005-01-01 geometry, {sz}x{sz}, centred disc, all {len(names)} planes: 15 of the intercept '
                    '(k_disc_sph<7, false>) + 11 every pixel has (k_sky<true>)')
        alg = sz * sz * 8 * len(names)
        scaling = 'weak'
    elif args.workload == 'maps':
        # the map-space chain at throughput size (body_xy.py:3227-3300, 3419-3491, 3667-3675 and the get_*_map planes): every
        # default backplane on a 0.1 deg rectangular grid, 1800 x 3600 cells - save_mapped_observation(degree_interval=0.1)
        from planetmapper_amd._lib import PLANE_NAMES

        sz = 1024
        names = list(PLANE_NAMES)
        g = load_scenario('jupiter_hst_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
        lon_h, lat_h = rectangular_grid(bool(g.west_positive), 0.1)
        n0, n1 = lon_h.shape
        lon_d, lat_d = torch.from_numpy(lon_h).to(d.dev), torch.from_numpy(lat_h).to(d.dev)
        planes = {n: torch.empty((n0, n1), dtype=torch.float64, device=d.dev) for n in names}

        def work():
            eng.backplanes_map_device(planes, lon_d, lat_d, n0, n1)

        units = n0 * n1
        metric = 'Mcell/s all 26 default backplanes in map space, 0.1 deg rectangular grid (1800 x 3600 cells)'
        workload = (f'Jupiter/HST 2005-01-01 geometry, {n0}x{n1} lon/lat grid resident in HBM, all {len(names)} map-space planes '
                    '(k_map_b0<true, true>)')
        alg = n0 * n1 * (16 + 8 * len(names))
        scaling = 'weak'
    elif args.workload == 'saturn':
        # Saturn-like spheroid, 4096^2, r0 = 800 px, rotation 20 deg
        sz = args.size
        names = HEADLINE + ['RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE']
        g = load_scenario('saturn_earth_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 800.0 * sz / 4096, float(np.deg2rad(20.0)), sz, sz, True)
        planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=d.dev) for n in names}

        def work():
            eng.backplanes_img_device(planes)

        units = sz * sz
        metric = 'Mpix/s Saturn + rings backplane set (lon/lat/phase/inc/emi/ring radius/lon/distance), 4096^2 frame'
        workload = f'Saturn-like spheroid seen from Earth 2005-01-01, {sz}x{sz}, r0={800.0 * sz / 4096:g} px, 8 planes'
        alg = sz * sz * 8 * len(names)
        scaling = 'weak'
    else:
        # IFU cube P x 1024 x 1024 f64 resident in HBM, planes sharded over ranks
        sz = 1024
        g = load_scenario('jupiter_hst_2005')
        eng.set_geometry(g)
        x0 = (sz - 1) / 2
        eng.set_disc(x0, x0, 0.9 * x0, 0.0, sz, sz, True)
        a, b, per_rank = shard_bounds(args.planes, d.world, d.rank)
        gen = torch.Generator(device=d.dev).manual_seed(5 + d.rank)
        cube = torch.randn((per_rank, sz, sz), generator=gen, device=d.dev, dtype=torch.float64)
        cube[torch.rand((per_rank, sz, sz), generator=gen, device=d.dev) < 1e-3] = float('nan')
        lon_h, lat_h = rectangular_grid(bool(g.west_positive))
        n0, n1 = lon_h.shape
        lon_d, lat_d = torch.from_numpy(lon_h).to(d.dev), torch.from_numpy(lat_h).to(d.dev)
        xm = torch.empty((n0, n1), dtype=torch.float64, device=d.dev)
        ym = torch.empty((n0, n1), dtype=torch.float64, device=d.dev)
        gathered = torch.empty((d.world, per_rank, n0, n1), dtype=torch.float64, device=d.dev)

        def work():
            eng.xy_map_device(lon_d, lat_d, n0, n1, xm, ym)
            # (the synthetic cube holds NaNs but no +-inf: no plane needs its nanmedian; the closing
            #  eng.synchronize() would raise otherwise)
            map_cube_sharded_device(eng, cube, np.float64, per_rank, xm, ym, n0, n1, gathered, d.rank,
                                    defer_median_check=True)  # fmt: skip

        units = args.planes * sz * sz
        metric = 'Mpix/s of cube pixels map-projected (get_mapped_data, 1 deg rectangular map, bilinear)'
        workload = (
            f'synthetic IFU cube {args.planes}x{sz}x{sz} f64 resident in HBM, {per_rank} planes per GPU, '
            f'map {n0}x{n1}, RCCL all-gather of mapped planes' + ('' if d.world > 1 else ' (skipped at N=1)')
        )
        alg = n0 * n1 * (16 + 40 * per_rank)
        scaling = 'strong'
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i):
        if i is not None:
            evs[i][0].record()
        work()
        if i is not None:
            evs[i][1].record()

    dt = timed_steps(d, step, args, preheat=min(args.preheat_steps, 200))
    eng.synchronize()
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if d.rank == 0:
        n_units = units * (d.world if scaling == 'weak' else 1)
        # the step IS its kernels here (events around the step on the stream they run on); HBM traffic per step from
        # the stamped --pmc passes of tools/pmc_profile.sh for this workload
        kernels = {'saturn': ('pm::k_disc_sph<5,',), 'all26': ('pm::k_disc_sph<7,', 'pm::k_sky<true>'),
                   'maps': ('pm::k_map_b0<true, true>',), 'cube': ('pm::k_reproject_xcd<double>',)}[args.workload]
        rec, stale = profile_record(kernels, args.workload) if (args.size == 4096 or args.workload in ('cube', 'maps')) else ({}, False)
        line = {
            'metric': metric, 'value': round(n_units * args.steps / dt / 1e6, 2), 'unit': 'Mcell/s' if args.workload == 'maps' else 'Mpix/s',
            'n_gpus': d.world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload},
            'roofline': {'kernel': ' + '.join(k.rstrip(',') + ('...>' if k.endswith(',') else '') for k in kernels),
                         'bound': 'hbm', 'achieved': round(alg / (step_ms * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         'traffic': int(rec['hbm_bytes']) if 'hbm_bytes' in rec else None, 'traffic_stale': stale,
                         'library_sha256': library_sha256()[:16], 'kernel_ms': round(step_ms, 4), 'algorithmic_bytes': alg,
                         **rocprof_fields(rec, alg)},
        }
        if d.world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline_other(args.workload, g, sz, names if args.workload != 'cube' else (), host_cores(), args.planes)
            line['cpu_model'] = cpu_model()
        print(json.dumps(line), flush=True)
    d.close()
    eng.close()


# ------------------------------------------------------------------ headline
def headline(args) -> None:
    d = Dist(args)
    torch = d.torch
    from planetmapper_amd.distributed import map_cube_sharded_device
    from planetmapper_amd.engine import Engine
    from planetmapper_amd.scenarios import load_scenario

    sz = args.size
    g = load_scenario('jupiter_hst_2005')
    x0 = y0 = (sz - 1) / 2
    eng = Engine(d.local_rank)
    # launch on torch's current stream so torch events / RCCL order with our kernels
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_geometry(g)
    eng.set_disc(x0, y0, 0.9 * x0, 0.0, sz, sz, True)  # BodyXY.centre_disc body_xy.py:791
    dev = d.dev
    # The two launches of a step - the frame's backplanes and get_mapped_data of its data plane - are
    # independent of each other: with --side-stream the second one (one wave per SIMD, 10 us of
    # latency) runs on a second engine context bound to a side stream, next to the frame kernel,
    # together with the all-gather that consumes it (a context owns one stream). Measured: 0.1877 ->
    # 0.1846 ms per step, frame kernel 0.1744 -> 0.1788 ms while sharing the chip.
    if not args.side_stream:
        side, eng_map = torch.cuda.current_stream(), eng
    else:
        side = torch.cuda.Stream(device=dev)
        eng_map = Engine(d.local_rank)
        eng_map.set_stream(side.cuda_stream)
        eng_map.set_geometry(g)
        eng_map.set_disc(x0, y0, 0.9 * x0, 0.0, sz, sz, True)

    planes = {n: torch.empty((sz, sz), dtype=torch.float64, device=dev) for n in HEADLINE}
    lon_h, lat_h = rectangular_grid(bool(g.west_positive))
    n0, n1 = lon_h.shape
    lon_d = torch.from_numpy(lon_h).to(dev)
    lat_d = torch.from_numpy(lat_h).to(dev)
    xm = torch.empty((n0, n1), dtype=torch.float64, device=dev)
    ym = torch.empty((n0, n1), dtype=torch.float64, device=dev)
    # synthetic data plane: limb-darkened disc + noise (SURVEY 8d config 3 recipe)
    gen = torch.Generator(device=dev).manual_seed(20050101 + d.rank)
    yy, xx = torch.meshgrid(
        torch.arange(sz, device=dev, dtype=torch.float64),
        torch.arange(sz, device=dev, dtype=torch.float64),
        indexing='ij',
    )
    mu = torch.sqrt(torch.clamp(1 - ((xx - x0) ** 2 + (yy - y0) ** 2) / (0.9 * x0) ** 2, min=0))
    data = mu + 0.05 * torch.randn((sz, sz), generator=gen, device=dev, dtype=torch.float64)
    del yy, xx, mu

    # two result buffers used alternately: a gather may stay in flight for two frames before its
    # buffer is written again (RCCL latency at 8 ranks is not known to be below one 0.19 ms step)
    gathered = [torch.empty((d.world, 1, n0, n1), dtype=torch.float64, device=dev) for _ in range(2)]
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    pending = [None, None]  # work handles of the all-gathers still in flight, one per buffer
    counter = [0]

    def step(i) -> None:
        if i is not None:
            ev0[i].record()
        eng.backplanes_img_device(planes)
        if i is not None:
            ev1[i].record()
        # (the data plane and the lon/lat grids were written on the main stream before the first step
        #  and are only read: no ordering between the streams is needed inside the loop)
        # x/y map of the 1 deg grid + this rank's plane -> its slot (pm_mapped_data: the C form of
        # get_mapped_data, one launch). Frames are independent of each other: nothing is exchanged
        # between ranks (with --gather-mapped the N mapped planes are all-gathered over RCCL, left
        # in flight so that the collective overlaps the next frame's backplane kernel). The data plane is finite
        # by construction (no +-inf: no plane needs its nanmedian), so the engine's flag check is
        # deferred to the closing synchronize(), which raises if that assumption were violated.
        k = counter[0] & 1
        counter[0] += 1
        with torch.cuda.stream(side):
            pending[k] = map_cube_sharded_device(
                eng_map, data, np.float64, 1, xm, ym, n0, n1, gathered[k], d.rank, 'linear', True, async_op=True,
                previous=pending[k], defer_median_check=True, lonlat=(lon_d, lat_d), gather=args.gather_mapped,
            )  # fmt: skip

    def drain() -> None:
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    class DrainingDist:  # the barrier of the timed region also completes the gathers in flight
        def __getattr__(self, name):
            return getattr(d, name)

        def barrier(self):
            drain()
            d.barrier()

    side.wait_stream(torch.cuda.current_stream())  # inputs written on the main stream are ready
    dt = timed_steps(DrainingDist(), step, args, preheat=args.preheat_steps)
    eng.synchronize()  # surfaces deferred device-side errors
    eng_map.synchronize()

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    frac_on_disc = float(torch.isfinite(planes['LON-GRAPHIC']).double().mean().item())
    line = None
    if d.rank == 0:
        alg = algorithmic_bytes(sz, sz, len(HEADLINE))
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        ms_per_step = dt / args.steps * 1e3
        rec, traffic_stale = profile_record('pm::k_disc_sph<1,') if sz == 4096 else ({}, False)
        # step-level bytes: 5 planes + x/y map (16 B in + 16 B out per cell) + 1-plane reprojection
        step_bytes = alg + n0 * n1 * 32 + n0 * n1 * (16 + 40)
        line = {
            'metric': METRIC,
            'value': round(d.world * sz * sz * args.steps / dt / 1e6, 2),
            'unit': 'Mpix/s',
            'n_gpus': d.world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
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
                'parallelism': f'independent frames (and their mapped planes), 1 per GPU x{d.world}'
                + ('' if d.world == 1 else (', RCCL all-gather of the mapped planes every step' if args.gather_mapped
                                            else ', no collective (the sharded cube with its RCCL all-gather: cube_host)')),
                'gather_mapped': bool(args.gather_mapped),
                'streams': 'one' if not args.side_stream else 'frame kernel on the main stream, get_mapped_data (+ all-gather) on a side stream',
                'preheat_steps': args.preheat_steps,
            },
            'roofline': {
                'kernel': 'pm::k_disc_sph<1, 0, 0, 28675ull> (DF_ILLUM, spheroid, the headline plane set)',
                'bound': 'hbm',
                'achieved': round(achieved, 2),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5),
                'traffic': int(rec['hbm_bytes']) if 'hbm_bytes' in rec else None,
                'traffic_stale': traffic_stale,  # true: profiles/traffic.json was measured on another build of the library
                'library_sha256': library_sha256()[:16],
                'kernel_ms': round(kernel_ms, 4),
                'algorithmic_bytes': alg,
                **rocprof_fields(rec, alg),
            },
            'step_roofline': {
                'bound': 'hbm',
                'algorithmic_bytes': step_bytes,
                'achieved': round(step_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            },
        }
        if 'fp64_flop' in rec:
            tf = rec['fp64_flop'] / (kernel_ms * 1e-3) / 1e12
            line['fp64'] = {
                'flop_per_launch': int(rec['fp64_flop']),
                'source': 'rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 (profiles/), FMA = 2, x 64 lanes',
                'tflops': round(tf, 2),
                'peak_tflops': FP64_VECTOR_PEAK_TFLOPS,
                'frac': round(tf / FP64_VECTOR_PEAK_TFLOPS, 4),
            }
    if d.world == 1 and d.rank == 0 and not args.no_cpu_baseline:
        cores = host_cores()
        line['cpu_baseline'] = cpu_baseline(g, sz, cores, budget_s=12.0, max_frames=8)
        line['cpu_baseline_1thread'] = cpu_baseline(g, sz, 1, budget_s=8.0, max_frames=1)
        line['cpu_model'] = cpu_model()
    if not args.no_extras:
        if d.world > 1 and d.rank == 0:
            # The headline of an N > 1 run is out BEFORE the extra sections start: the sharded host-fed cube below
            # issues collectives from inside the engine's chunk callback, which has run on one GPU (gloo, loopback
            # transport) but never over RCCL between GPUs - if that hangs and the run is killed, this line stands.
            # The complete line (same fields + cube_host) follows when the sections are through.
            print(json.dumps(dict(line, extras='pending: the complete line, with cube_host, follows')), flush=True)
        watchdog = None
        partial_sec: dict = {}
        if d.world > 1:
            # ... and if the section HANGS (a collective some rank never joins), every rank gives up after a deadline of
            # its own: rank 0 prints the complete line with the section marked as timed out, and every rank leaves with
            # EXTRAS_TIMED_OUT_STATUS (75, EX_TEMPFAIL) - a status of its own, neither success nor a crash: the headline of
            # this N is on stdout twice by then (the `extras: pending` line before the section, the complete line now), and a
            # caller that looks at the status learns that a section was abandoned. Rank 0 leaves first; the other ranks wait
            # a little longer, so that the launcher (which tears the group down at the first non-zero exit) cannot cut rank 0
            # off before its line is out. (A rank stuck in a collective cannot tear its process group down: os._exit is all
            # that is left.)
            import threading

            deadline = float(os.environ.get('PM_BENCH_EXTRAS_TIMEOUT_S', '300'))

            def bail() -> None:
                if d.rank == 0:
                    # (with the plain form's measured numbers - map, finish, one all-gather - if the section got that far)
                    print(json.dumps(dict(line, cube_host=dict(partial_sec, error=f'timed out after {deadline:.0f} s: section abandoned',
                                                               exit_status=EXTRAS_TIMED_OUT_STATUS))), flush=True)
                    sys.stdout.flush()
                    note = os.environ.get('PM_BENCH_STATUS_FILE')  # (self_launch: the launcher turns every rank status into 1)
                    if note:
                        with open(note, 'w') as f:
                            f.write(str(EXTRAS_TIMED_OUT_STATUS))
                else:
                    time.sleep(2.0)
                os._exit(EXTRAS_TIMED_OUT_STATUS)

            watchdog = threading.Timer(deadline, bail)
            watchdog.daemon = True
            watchdog.start()
        if d.world == 1 and d.rank == 0:
            line['host_path'] = host_path_section(eng, g, sz)
            line['api_path'] = api_path_section(g, sz, d.local_rank)
            try:
                line['interpolations'] = interpolations_section(d.local_rank, g)
            except Exception as e:  # noqa: BLE001
                line['interpolations'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        try:
            sec = cube_host_section(d, eng, g, args.planes, steps_fed=7, steps_resident=50, partial=partial_sec)
        except Exception as e:  # noqa: BLE001
            sec = {'error': f'{type(e).__name__}: {e}'[:500]}
        if d.world == 1 and d.rank == 0 and not args.no_shared_gpu and 'error' not in sec:
            sec['shared_gpu'] = shared_gpu_section(args)
        if watchdog is not None:
            watchdog.cancel()
        if d.rank == 0:
            line['cube_host'] = sec
    if d.rank == 0:
        print(json.dumps(line), flush=True)
    d.close()
    if eng_map is not eng:
        eng_map.close()
    eng.close()


def main() -> None:
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    if args.rehearse:
        return rehearse(args)
    if args.workload != 'frame':
        return other_workloads(args)
    return headline(args)


if __name__ == '__main__':
    main()
