"""
N > 1 path on CPU: world_size = 2 `gloo` process group, planes of a cube sharded over
the ranks and re-assembled by one all-gather (planetmapper_amd/distributed.py). The
engine is the oracle-backed test double, so this checks the sharding arithmetic, the
padding of uneven shards and the collective - not the kernels.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from planetmapper_amd.distributed import shard_bounds


def test_shard_bounds():
    assert shard_bounds(10, 2, 0) == (0, 5, 5) and shard_bounds(10, 2, 1) == (5, 10, 5)
    assert shard_bounds(10, 4, 3) == (9, 10, 3)  # ceil(10/4) = 3: last rank gets one plane
    assert shard_bounds(3, 8, 5) == (3, 3, 1)  # more ranks than planes: empty shard
    assert shard_bounds(512, 8, 7) == (448, 512, 64)  # BASELINE config 5
    covered = []
    for r in range(8):
        a, b, _ = shard_bounds(13, 8, r)
        covered += list(range(a, b))
    assert covered == list(range(13))
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_planes: int, tmpdir: str) -> None:
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        from planetmapper_amd import Observation
        from planetmapper_amd.distributed import get_mapped_data_sharded
        from planetmapper_amd.scenarios import load_scenario

        g = load_scenario('jupiter_hst_2005')
        rng = np.random.default_rng(5)
        cube = rng.standard_normal((n_planes, 24, 20))
        cube[rng.random(cube.shape) < 0.01] = np.nan
        obs = Observation(data=cube, geometry=g, engine=OracleEngine())
        obs.set_disc_params(9.5, 12.0, 8.0, 30.0)
        full = obs.get_mapped_data(degree_interval=15)
        eng = obs._engine
        n_before = len([c for c in eng.calls if c[0] == 'cube'])
        sharded = get_mapped_data_sharded(obs, degree_interval=15)
        assert sharded.shape == full.shape == (n_planes, 12, 24)
        assert np.array_equal(sharded, full, equal_nan=True)
        # this rank only mapped its own block of planes
        cube_calls = [c for c in eng.calls if c[0] == 'cube'][n_before:]
        a, b, _ = shard_bounds(n_planes, world, rank)
        assert sum(c[1][0] for c in cube_calls) == b - a
        nearest = get_mapped_data_sharded(obs, 'nearest', degree_interval=15)
        assert np.array_equal(nearest, obs.get_mapped_data('nearest', degree_interval=15), equal_nan=True)
        # per-rank plane blocks: this rank hands over ONLY its own planes (obs.data is not read) ...
        a, b, _ = shard_bounds(n_planes, world, rank)
        obs2 = Observation(data=np.zeros((1, 24, 20)), geometry=g, engine=OracleEngine())
        obs2.set_disc_params(9.5, 12.0, 8.0, 30.0)
        blockwise = get_mapped_data_sharded(obs2, local_planes=cube[a:b].copy(), n_planes=n_planes, degree_interval=15)
        assert np.array_equal(blockwise, full, equal_nan=True)
        # ... and may keep its slice (no collective)
        mine = get_mapped_data_sharded(obs2, local_planes=cube[a:b].copy(), n_planes=n_planes, gather=False,
                                       degree_interval=15)
        assert mine.shape == (b - a, 12, 24) and np.array_equal(mine, full[a:b], equal_nan=True)
        with pytest.raises(ValueError):
            get_mapped_data_sharded(obs2, local_planes=np.zeros((b - a + 1, 24, 20)), n_planes=n_planes, degree_interval=15)
        # the other interpolation arguments of map_img are forwarded
        cubic = get_mapped_data_sharded(obs, 'cubic', propagate_nan=False, degree_interval=15)
        assert np.array_equal(cubic, obs.get_mapped_data('cubic', propagate_nan=False, degree_interval=15), equal_nan=True)
        sm = get_mapped_data_sharded(obs, 'linear', spline_smoothing=3.0, propagate_nan=False, degree_interval=15)
        assert np.allclose(sm, obs.get_mapped_data('linear', spline_smoothing=3.0, propagate_nan=False, degree_interval=15),
                           rtol=0, atol=1e-12, equal_nan=True)
        open(os.path.join(tmpdir, f'ok{rank}'), 'w').close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_planes', [6, 5, 1])
def test_sharded_mapping_world_size_2(tmp_path, n_planes):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_planes, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _pipeline_worker(rank: int, world: int, port: int, tmpdir: str) -> None:
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import torch

        from oracle import oracle
        from oracle_engine import OracleEngine
        from planetmapper_amd.distributed import map_cube_sharded_device
        from planetmapper_amd.scenarios import load_scenario

        g = load_scenario('jupiter_hst_2005')
        eng = OracleEngine()
        eng.set_geometry(g)
        eng.set_disc(9.5, 12.0, 8.0, 0.5, 20, 24, True)
        lon, lat = oracle.rectangular_grid(g, 15.0)
        out = eng.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon, lat)
        xm, ym = torch.from_numpy(out['PIXEL-X'].copy()), torch.from_numpy(out['PIXEL-Y'].copy())
        n0, n1 = lon.shape
        gathered = torch.full((world, 2, n0, n1), float('nan'), dtype=torch.float64)
        pending = None
        # what bench.py does per step: map this rank's planes into its slot, all-gather the slots
        # asynchronously, hand the work handle to the next step
        for step in range(3):
            rng = np.random.default_rng(100 * step)  # same stream on every rank
            frames = rng.standard_normal((world, 2, 24, 20))
            mine = torch.from_numpy(frames[rank].copy())
            pending = map_cube_sharded_device(eng, mine, np.float64, 2, xm, ym, n0, n1, gathered, rank,
                                              'linear', True, async_op=True, previous=pending,
                                              defer_median_check=True)  # fmt: skip
            assert pending is not None
        pending.wait()
        assert ('sync',) not in eng.calls  # the deferred form never goes back to the host
        expect = np.stack([oracle.map_cube(frames[r], xm.numpy(), ym.numpy()) for r in range(world)])
        assert np.array_equal(gathered.numpy(), expect, equal_nan=True)
        # the fused form (x/y map computed by the same call, bench.py's step) gives the same slots
        g2 = torch.full((world, 2, n0, n1), float('nan'), dtype=torch.float64)
        xm2, ym2 = torch.empty_like(xm), torch.empty_like(ym)
        h2 = map_cube_sharded_device(eng, mine, np.float64, 2, xm2, ym2, n0, n1, g2, rank, 'linear', True,
                                     lonlat=(torch.from_numpy(lon.copy()), torch.from_numpy(lat.copy())))
        if h2 is not None:
            h2.wait()
        assert np.array_equal(g2.numpy(), gathered.numpy(), equal_nan=True)
        assert np.array_equal(xm2.numpy(), xm.numpy(), equal_nan=True)
        # synchronous form returns the finished handle / None for a single rank
        h = map_cube_sharded_device(eng, mine, np.float64, 2, xm, ym, n0, n1, gathered, rank)
        assert h is None or h.is_completed()
        # ... after finishing the call (flag check / nanmedian replay) BEFORE the collective
        assert eng.calls[-1] == ('sync',) and eng.calls[-2][0] == 'cube'
        open(os.path.join(tmpdir, f'ok{rank}'), 'w').close()
    finally:
        dist.destroy_process_group()


def test_pipelined_device_all_gather_world_size_2(tmp_path):
    """bench.py's N > 1 step (slot write + in-place all-gather with the handle carried over)"""
    world = 2
    mp.spawn(_pipeline_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _rows_worker(rank: int, world: int, port: int, ny: int, tmpdir: str) -> None:
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle
        from oracle_engine import OracleEngine
        from planetmapper_amd.distributed import backplanes_img_sharded
        from planetmapper_amd.scenarios import load_scenario

        g = load_scenario('jupiter_hst_2005')
        eng = OracleEngine()
        eng.set_geometry(g)
        nx = 23
        eng.set_disc(11.0, ny / 2, 0.4 * ny, 0.3, nx, ny, True)
        names = ['LON-GRAPHIC', 'EMISSION', 'RA', 'RING-RADIUS']
        got = backplanes_img_sharded(eng, names, ny, nx)
        d = oracle.make_disc(11.0, ny / 2, 0.4 * ny, 0.0, nx, ny)
        d.rotation_rad = 0.3
        ref = oracle.backplanes_img(g, d, names)
        for n in names:
            assert got[n].shape == (ny, nx)
            assert np.array_equal(got[n], ref[n], equal_nan=True), n
        a, b, _ = shard_bounds(ny, world, rank)
        rows = [c for c in eng.calls if c[0] == 'rows']
        assert [(c[2], c[3]) for c in rows] == ([(a, b - a)] if b > a else [])
        open(os.path.join(tmpdir, f'ok{rank}'), 'w').close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ny', [20, 17, 1])
def test_row_block_sharded_backplanes_world_size_2(tmp_path, ny):
    """a frame's rows split over two ranks and all-gathered (SURVEY 8e partition 2)"""
    world = 2
    mp.spawn(_rows_worker, args=(world, _free_port(), ny, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _cube_host_worker(rank: int, world: int, port: int, planes: int, tmpdir: str) -> None:
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))  # fmt: skip
    import bench
    from oracle_engine import OracleEngine
    from planetmapper_amd.scenarios import load_scenario

    d = bench.Dist(bench.parse_args(['--gpus', str(world), '--rehearse']))  # gloo group, CPU tensors
    try:
        eng = OracleEngine()
        g = load_scenario('jupiter_hst_2005')
        sec = bench.cube_host_section(d, eng, g, planes, steps_fed=1, steps_resident=1, sz=48, degree_interval=20.0)
        a, b, per_rank = shard_bounds(planes, world, rank)
        assert sec['rccl_ranks'] == world and sec['planes_per_rank'] == per_rank
        assert sec['fed_equals_resident'] is True  # the gathered cube of the host-fed step == the resident one
        assert sec['fed_equals_resident_plain_allgather'] is True  # ... and so is the plain form's (one all-gather per step)
        assert sec['ms_per_step_host_fed_no_collective'] > 0 and 'pipelined_error' not in sec
        fed = [c for c in eng.calls if c[0] == 'host_cube']
        assert all(c[1][0] == b - a for c in fed) and (len(fed) > 0) == (b > a)
        # every step finishes this rank's planes before the collective
        kinds = [c[0] for c in eng.calls]
        assert all(kinds[i + 1] == 'sync' for i, k in enumerate(kinds) if k in ('host_cube', 'cube'))
        open(os.path.join(tmpdir, f'ok{rank}'), 'w').close()
    finally:
        d.close()


@pytest.mark.parametrize('planes', [6, 5, 1])
def test_bench_cube_host_section_world_size_2(tmp_path, planes):
    """
    bench.py's host-fed cube section - the code the driver's N = 1, 2, 4, 8 runs execute for the north
    star's scaling case - with two gloo ranks and the oracle-backed engine double: shard bounds,
    per-rank slots, the all-gather, uneven and empty shards.
    """
    world = 2
    mp.spawn(_cube_host_worker, args=(world, _free_port(), planes, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _protocol_worker(rank: int, world: int, port: int, tmpdir: str) -> None:
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle
        from oracle_engine import OracleEngine
        import planetmapper_amd.distributed as D
        from planetmapper_amd import Observation
        from planetmapper_amd.scenarios import load_scenario

        g = load_scenario('jupiter_hst_2005')
        eng = OracleEngine()
        eng.set_geometry(g)
        eng.set_disc(9.5, 12.0, 8.0, 0.5, 20, 24, True)
        lon, lat = oracle.rectangular_grid(g, 15.0)
        o = eng.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon, lat)
        xm, ym = torch.from_numpy(o['PIXEL-X'].copy()), torch.from_numpy(o['PIXEL-Y'].copy())
        n0, n1 = lon.shape
        planes = 11  # 6 + 5: the second rank's last exchange is half padding
        rng = np.random.default_rng(77)
        cube = rng.standard_normal((planes, 24, 20))
        cube[rng.random(cube.shape) < 0.02] = np.nan
        a, b, per_rank = D.shard_bounds(planes, world, rank)
        expect = oracle.map_cube(cube, xm.numpy(), ym.numpy())

        # ---- (b) several exchanges per block: the all-gathers of finished exchanges are in flight while
        # the next one is mapped; the pieces land rank-major
        real_exchange = D.exchange_planes
        D.exchange_planes = lambda per, n0_, n1_: 2  # (the real rule gives one exchange for a map this small)
        try:
            for host_cube in (True, False):
                gathered = torch.full((world, per_rank, n0, n1), -7.0, dtype=torch.float64)
                local = cube[a:b].copy() if host_cube else torch.from_numpy(cube[a:b].copy())
                eng.calls.clear()
                D.map_cube_sharded_pipelined(eng, local, np.float64, planes, xm, ym, n0, n1, gathered, rank, world,
                                             host_cube=host_cube)
                got = gathered.reshape(world * per_rank, n0, n1).numpy()
                assert np.array_equal(got[:planes], expect, equal_nan=True)
                assert np.isnan(got[planes:]).all()  # padding of the short block
                kinds = [c[0] for c in eng.calls]
                # ONE engine call per block (its own pipeline stays whole), finished before the closing agreement
                assert kinds == ['host_cube' if host_cube else 'cube', 'sync']
            # planes redone with their nanmedian after they were sent: everybody gathers once more
            eng.redo_planes = 1 if rank == 0 else 0
            gathered = torch.full((world, per_rank, n0, n1), -7.0, dtype=torch.float64)
            stages: dict = {}
            D.map_cube_sharded_pipelined(eng, cube[a:b].copy(), np.float64, planes, xm, ym, n0, n1, gathered, rank, world,
                                         host_cube=True, stages=stages)
            eng.redo_planes = 0
            assert np.array_equal(gathered.reshape(-1, n0, n1).numpy()[:planes], expect, equal_nan=True)
            # the stage record of the call (bench.py `stages`): the mapping call, what was left of the exchanges, the agreement
            assert set(stages) == {'map_call', 'exchange_exposed', 'agreement'} and all(v >= 0.0 for v in stages.values())
            assert stages['map_call'] > 0.0 and stages['agreement'] > 0.0
            # ---- (e) a rank whose mapping raises: nobody hangs, every rank raises
            class Boom(RuntimeError):
                pass

            def failing(*args, **kw):
                raise Boom('injected')

            eng_bad = OracleEngine()
            eng_bad.set_geometry(g)
            eng_bad.set_disc(9.5, 12.0, 8.0, 0.5, 20, 24, True)
            if rank == 1:
                eng_bad.map_cube_host_to_device = failing
            gathered = torch.zeros((world, per_rank, n0, n1), dtype=torch.float64)
            with pytest.raises(Boom if rank == 1 else D.PeerFailedError):
                D.map_cube_sharded_pipelined(eng_bad, cube[a:b].copy(), np.float64, planes, xm, ym, n0, n1, gathered, rank, world,
                                             host_cube=True)
            # the group is still usable afterwards
            D.map_cube_sharded_pipelined(eng, cube[a:b].copy(), np.float64, planes, xm, ym, n0, n1, gathered, rank, world,
                                         host_cube=True)
            assert np.array_equal(gathered.reshape(-1, n0, n1).numpy()[:planes], expect, equal_nan=True)
        finally:
            D.exchange_planes = real_exchange

        # ---- (e) the numpy-level API: agreement before the all-gather
        obs = Observation(data=cube, geometry=g, engine=OracleEngine())
        obs.set_disc_params(9.5, 12.0, 8.0, 30.0)
        full = obs.get_mapped_data(degree_interval=15)
        obs_bad = Observation(data=cube, geometry=g, engine=OracleEngine())
        obs_bad.set_disc_params(9.5, 12.0, 8.0, 30.0)
        if rank == 0:
            def bad_map_img(*args, **kw):
                raise ValueError('injected on rank 0')

            obs_bad.map_img = bad_map_img
        with pytest.raises(ValueError if rank == 0 else D.PeerFailedError):
            D.get_mapped_data_sharded(obs_bad, degree_interval=15)
        assert np.array_equal(D.get_mapped_data_sharded(obs, degree_interval=15), full, equal_nan=True)

        # ---- no collective at all: every rank writes its planes into ONE shared array
        shared = np.lib.format.open_memmap(os.path.join(tmpdir, 'shared.npy'), mode='r+')
        out = D.get_mapped_data_sharded(obs, gather=False, out=shared, degree_interval=15)
        assert out is shared
        shared.flush()
        dist.barrier()
        again = np.load(os.path.join(tmpdir, 'shared.npy'))
        assert np.array_equal(again, full, equal_nan=True)
        with pytest.raises(ValueError):
            D.get_mapped_data_sharded(obs, gather=False, out=np.zeros((1, 2, 3)), degree_interval=15)
        open(os.path.join(tmpdir, f'ok{rank}'), 'w').close()
    finally:
        dist.destroy_process_group()


def test_pipelined_exchange_failure_agreement_and_shared_output_world_size_2(tmp_path):
    """
    The protocol of the sharded cube (distributed.map_cube_sharded_pipelined = pm_map_cube_sharded):
    several exchanges per block with their all-gathers started behind the mapping, a failing rank that
    keeps its peers from hanging (every rank raises), and the collective-free form where every rank
    writes its planes into one shared (P, n0, n1) array.
    """
    world = 2
    np.lib.format.open_memmap(os.path.join(tmp_path, 'shared.npy'), mode='w+', dtype=np.float64, shape=(11, 12, 24))[...] = -1.0
    mp.spawn(_protocol_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def test_exchange_planes_rule_equals_the_c_one():
    import ctypes  # noqa: F401

    from planetmapper_amd import _lib
    from planetmapper_amd.distributed import exchange_planes

    lib = _lib.load()
    for per in (0, 1, 2, 7, 8, 9, 64, 65, 256, 512, 4096):
        for n0, n1 in ((180, 360), (6, 12), (1, 1), (1800, 3600), (0, 5)):
            assert lib.pm_exchange_planes(per, n0, n1) == exchange_planes(per, n0, n1), (per, n0, n1)
    assert exchange_planes(64, 180, 360) == 9 and exchange_planes(512, 180, 360) == 64
