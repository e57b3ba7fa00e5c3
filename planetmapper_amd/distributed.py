"""
Sharding of the path over the GPUs of one node: wavelength planes of
`Observation.get_mapped_data`, and row blocks of a frame's backplane images.

The planes of a cube are independent in the reference (`observation.py:892-904` maps
them one after another with the same x/y map), so the path shards with no halo and no
reduction: rank r maps a contiguous block of planes on its own GPU (the tiny x/y map is
recomputed per rank instead of broadcast), and ONE all-gather of the mapped planes
(RCCL over xGMI for the `nccl` backend) assembles the (P, n_lat, n_lon) result on every
rank. One process per GPU, `torch.distributed` for the process group.

The same code runs under the `gloo` backend on CPU tensors (used by the world_size=2
tests, where the engine is the oracle-backed test double).
"""

from __future__ import annotations

import numpy as np


def shard_bounds(n_planes: int, world_size: int, rank: int) -> tuple[int, int, int]:
    """
    Contiguous block of planes of `rank`: returns (start, stop, per_rank) with
    per_rank = ceil(n_planes / world_size); trailing ranks may get fewer (or no) planes.
    """
    if n_planes < 0 or world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError('invalid shard request')
    per_rank = -(-n_planes // world_size) if n_planes else 0
    start = min(rank * per_rank, n_planes)
    stop = min(start + per_rank, n_planes)
    return start, stop, per_rank


def exchange_planes(per_rank: int, n0: int, n1: int) -> int:
    """
    Planes per exchange of the pipelined all-gather (`pm_exchange_planes` of the C ABI): at most 8
    exchanges per block, each of at least 4 MiB of mapped planes. A function of shapes only, so every
    rank issues the same sequence of collectives whatever happens on it.
    """
    if per_rank < 0 or n0 < 0 or n1 < 0:
        raise ValueError('negative shape')
    if per_rank == 0:
        return 1
    nmap_bytes = max(n0 * n1 * 8, 1)
    by_count = -(-per_rank // 8)
    by_bytes = -(-(4 << 20) // nmap_bytes)
    return min(per_rank, max(by_count, by_bytes, 1))


# ONE class for "a sharded call failed on ANOTHER rank: the gathered result is not valid on this rank
# either", whichever layer reports it: `Engine._check` for `PM_ERR_PEER` of the C ABI, `agree_on_success` /
# `map_cube_sharded_pipelined` for the torch.distributed form
from ._lib import PeerFailedError  # noqa: E402


def agree_on_success(error: BaseException | None, group=None) -> None:
    """
    The closing agreement of a sharded call: every rank contributes whether its part worked, one tiny
    all-reduce (MIN) spreads the verdict, and EVERY rank raises if any rank failed - its own exception
    where there is one, `PeerFailedError` elsewhere. A rank must reach this call even when its own
    work failed (catch, participate, then raise): leaving early would strand its peers in the next
    collective.
    """
    import torch
    import torch.distributed as dist

    ok = torch.tensor([0 if error is None else -1], dtype=torch.int32, device=_group_device(group))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if error is not None:
        raise error
    if int(ok.item()) != 0:
        raise PeerFailedError('another rank failed in this sharded call: the gathered result is not valid')


def _group_device(group):
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    if backend == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_planes(local: np.ndarray, n_planes: int, group=None) -> np.ndarray:
    """
    All-gather equally sized (padded) plane blocks and trim to `n_planes`.
    `local`: (per_rank, n0, n1) float64, rows beyond this rank's share are padding.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = _group_device(group)
    t_local = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    out = torch.empty((world,) + tuple(t_local.shape), dtype=t_local.dtype, device=dev)
    dist.all_gather_into_tensor(out.view(-1), t_local.view(-1), group=group)
    full = out.reshape((world * local.shape[0],) + tuple(local.shape[1:]))[:n_planes]
    return full.cpu().numpy()


def _check_rank_device(obs, group) -> None:
    """Under RCCL every rank must compute on its own GPU (the one torch has made current)."""
    import torch
    import torch.distributed as dist

    if dist.get_backend(group) != 'nccl':
        return
    dev = getattr(getattr(obs, '_engine', None), 'device', None)
    cur = torch.cuda.current_device()
    if dev is not None and dev != cur:
        raise RuntimeError(
            f'this rank\'s engine runs on GPU {dev} but torch.cuda.current_device() is {cur}: build the '
            'Observation / BodyXY with device=LOCAL_RANK (and torch.cuda.set_device(LOCAL_RANK)) so that '
            'each rank maps its planes on its own GPU'
        )


def get_mapped_data_sharded(
    obs,
    interpolation='linear',
    *,
    spline_smoothing: float = 0,
    propagate_nan: bool = True,
    smooth_oversample_by: int = 5,
    smooth_max_oversampled_img_size: int = 10_000,
    group=None,
    local_planes: np.ndarray | None = None,
    n_planes: int | None = None,
    gather: bool = True,
    out: np.ndarray | None = None,
    **map_kwargs,
) -> np.ndarray:
    """
    `Observation.get_mapped_data` (observation.py:826-905) with the planes of the cube sharded over
    the ranks of `group` (default: the world): rank r maps the contiguous block
    `shard_bounds(P, world, r)` on its own GPU and one all-gather assembles the result.

    Where the planes come from:
      * default: every rank holds the same `obs.data` (P, ny, nx) and slices its block from it;
      * `local_planes` (+ `n_planes` = P, the size of the whole cube): this rank passes ONLY its own
        block, shape (stop - start, ny, nx) - e.g. a slice of a memory-mapped FITS cube or planes
        read by this rank alone - and `obs.data` is not touched, so no rank needs the full cube in
        host memory (at BASELINE config 5, 512 MiB per rank instead of 4 GiB on each of 8 ranks).

    What comes back:
      * `gather=True`: the full (P, n0, n1) float64 result on every rank (one all-gather, RCCL over
        xGMI for the nccl backend);
      * `gather=False`: this rank's (stop - start, n0, n1) block only, no collective at all (SURVEY
        8e: "otherwise none - each rank keeps / writes its slice");
      * `gather=False, out=<(P, n0, n1) float64 array>`: the block is written into `out[start:stop]` and
        `out` is returned - with ONE array in shared memory (`multiprocessing.shared_memory`, a
        memory-mapped file) handed to every rank the whole result assembles in the caller's array
        without any collective.

    A rank whose mapping raises does not leave its peers waiting: with `gather=True` every rank first
    agrees on success (`agree_on_success`: a 4-byte all-reduce) and all of them raise if one failed.

    The interpolation arguments are those of `map_img` (body_xy.py:1414-1429).
    """
    import torch.distributed as dist

    interp = dict(
        interpolation=interpolation, spline_smoothing=spline_smoothing, propagate_nan=propagate_nan,
        smooth_oversample_by=smooth_oversample_by, smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
    )  # fmt: skip
    if not dist.is_initialized():
        if local_planes is not None:
            return obs.map_img(local_planes, **interp, **map_kwargs)
        return obs.get_mapped_data(**interp, **map_kwargs)
    _check_rank_device(obs, group)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local_planes is None:
        n_planes = obs.data.shape[0]
    elif n_planes is None:
        raise ValueError('n_planes (planes of the whole cube) is required with local_planes')
    start, stop, per_rank = shard_bounds(int(n_planes), world, rank)
    if local_planes is None:
        local_planes = obs.data[start:stop]
    local_planes = np.asarray(local_planes)
    if local_planes.ndim == 2:
        local_planes = local_planes[None]
    if local_planes.shape[0] != stop - start:
        raise ValueError(
            f'rank {rank} of {world} owns planes [{start}, {stop}) of {n_planes} but was given {local_planes.shape[0]}'
        )
    error, mine, shape = None, None, None
    try:
        x_map = obs.get_x_map(**map_kwargs)
        shape = x_map.shape
        mine = obs.map_img(local_planes, **interp, **map_kwargs) if stop > start else np.empty((0,) + shape)
    except Exception as e:  # noqa: BLE001 - reported through the agreement below
        if not gather:
            raise
        error = e
    if not gather:
        if out is None:
            return mine
        if out.shape != (int(n_planes),) + shape or out.dtype != np.float64:
            raise ValueError(f'out must be a float64 array of shape {(int(n_planes),) + shape}')
        out[start:stop] = mine
        return out
    agree_on_success(error, group)
    n0, n1 = shape
    if n_planes == 0:
        return np.empty((0, n0, n1))
    local = np.full((per_rank, n0, n1), np.nan)
    local[: stop - start] = mine
    return all_gather_planes(local, int(n_planes), group)


def backplanes_img_sharded(engine, names, ny: int, nx: int, *, alt: float = 0.0, group=None) -> dict[str, np.ndarray]:
    """
    One frame's image backplanes with the ROWS sharded over the ranks of `group`: rank r computes
    the contiguous block of ceil(ny / world) rows `shard_bounds(ny, world, r)` on its GPU
    (`pm_backplanes_img_rows`; pixels are independent in the reference, body_xy.py:3155-3164, so
    there is no halo) and one all-gather assembles the (ny, nx) planes on every rank.
    `engine` must already hold the geometry and disc of the full frame on every rank.
    """
    import torch
    import torch.distributed as dist

    names = list(names)
    if not dist.is_initialized():
        return engine.backplanes_img_rows(names, 0, ny, alt=alt)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, stop, per_rank = shard_bounds(ny, world, rank)
    local = np.full((len(names), per_rank, nx), np.nan)
    if stop > start:
        block = engine.backplanes_img_rows(names, start, stop - start, alt=alt)
        for i, n in enumerate(names):
            local[i, : stop - start] = block[n]
    dev = _group_device(group)
    t_local = torch.from_numpy(local).to(dev)
    out = torch.empty((world,) + tuple(t_local.shape), dtype=t_local.dtype, device=dev)
    dist.all_gather_into_tensor(out.view(-1), t_local.view(-1), group=group)
    full = out.permute(1, 0, 2, 3).reshape(len(names), world * per_rank, nx)[:, :ny]
    full = full.cpu().numpy()
    return {n: np.ascontiguousarray(full[i]) for i, n in enumerate(names)}


def map_cube_sharded_device(engine, cube, dtype, n_planes_local: int, x_map, y_map, n0: int, n1: int,
                            gathered, rank: int, interpolation='linear', propagate_nan=True, group=None,
                            async_op: bool = False, previous=None, defer_median_check: bool = False, lonlat=None,
                            gather: bool = True):
    """
    Device-resident variant used by the benchmark: this rank's `n_planes_local` planes
    (`cube`, a device tensor / pointer) are mapped straight into its slot of `gathered`
    ((world, n_planes_local, n0, n1) float64 cuda tensor) and the slots are exchanged with
    one in-place RCCL all-gather. Nothing touches the host.

    `async_op=True` returns the collective's work handle instead of waiting for it, so the
    caller can enqueue independent kernels (the next frame's backplanes) while the gather is
    in flight over xGMI; pass that handle back as `previous` on the next call: it is waited for
    right before this rank's slot is overwritten.

    A device-mode `pm_map_cube` is final only after `pm_synchronize()`: planes in which a sampled
    pixel needs the plane's nanmedian (+-inf pixels, neighbourhoods without a finite pixel) are
    completed there. By default this function therefore synchronises the engine BEFORE the
    collective, so that peers never receive provisional values. `defer_median_check=True` skips
    that host round trip (fully asynchronous step) for data known to hold no such pixels; the
    caller's next `engine.synchronize()` still detects a violation and raises.

    `gather=False` leaves the slots where they are (independent frames: nothing to exchange).
    """
    import torch.distributed as dist

    if previous is not None:
        previous.wait()
    mine = gathered[rank]
    if lonlat is not None:
        # `lonlat` = (lon, lat) device grids: the x/y map is computed by the same call (`pm_mapped_data`,
        # one launch for up to 8 planes) into x_map / y_map instead of being read from them
        engine.mapped_data_device(cube, dtype, n_planes_local, lonlat[0], lonlat[1], n0, n1, x_map, y_map, mine,
                                  interpolation, propagate_nan)
    else:
        engine.map_cube_device(cube, dtype, n_planes_local, x_map, y_map, n0, n1, mine, interpolation, propagate_nan)
    if not defer_median_check:
        engine.synchronize()
    if gather and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), group=group, async_op=async_op)
    return None


def map_cube_sharded_pipelined(engine, local_cube, dtype, n_planes_total: int, x_map, y_map, n0: int, n1: int, gathered,
                               rank: int, world: int, *, interpolation='linear', propagate_nan=True, group=None,
                               host_cube: bool = False, gather: bool = True, pipeline_chunks: bool | None = None,
                               stages: dict | None = None) -> None:
    """
    The plane-sharded cube with the collective PIPELINED behind the mapping - the torch.distributed
    form of `pm_map_cube_sharded` (same protocol, include/planetmapper_hip.h):

      1. this rank's block (`local_cube`: its planes `shard_bounds(n_planes_total, world, rank)` only - a
         device tensor / pointer, or a numpy array in host memory with `host_cube=True`) is cut into
         exchanges of `exchange_planes(per_rank, n0, n1)` planes - a function of shapes only;
      2. ONE engine call maps the block into this rank's slot of `gathered` ((world, per_rank, n0, n1)
         float64 on the group's device), so the engine's own pipeline (collecting / copying chunk k + 1
         while chunk k is mapped) stays whole; each time the kernels of further planes are on the
         engine's stream (`Engine.set_chunk_callback`) the all-gathers of the exchanges they complete are
         started asynchronously: exchange k crosses xGMI while later planes are still on their way in;
      3. after the call has been FINISHED (`engine.synchronize()`: flag check / nanmedian replay) any
         exchange not started yet - a rank without planes, a rank whose mapping raised - is started all
         the same: no peer is left waiting in a collective;
      4. one small all-reduce closes the call: (ranks that failed, ranks that redid planes with their
         nanmedian after those planes had been sent). Any failure: EVERY rank raises (its own exception,
         `PeerFailedError` elsewhere). Any redo (rare: +-inf pixels): every rank gathers the block once more.

    `gather=False`: only this rank's slot is written, no collective, errors raise at once.
    `pipeline_chunks=True` keeps the exchange bookkeeping running even when nothing is exchanged (a
    single process timing what one rank of N would do: bench.py's shard proxy).
    `stages`: a dict that receives where this rank's call spent its time, in ms - `map_call` (the engine call and its
    finishing `synchronize()`; `Engine.last_stages_ms()` has its inside), `exchange_exposed` (what was left of the
    all-gathers once the mapping had finished), `agreement`. Asking for it makes the call wait for the device at those
    points, which it otherwise does not.
    """
    import torch
    import torch.distributed as dist

    start, stop, per_rank = shard_bounds(int(n_planes_total), int(world), int(rank))
    mine_n = stop - start
    exchange = gather and dist.is_initialized() and dist.get_world_size(group) > 1
    chunked = exchange or bool(pipeline_chunks)
    step = exchange_planes(per_rank, n0, n1) if chunked else max(per_rank, 1)
    slot = gathered[rank]
    if mine_n < per_rank:
        slot[mine_n:].fill_(float('nan'))  # planes of a short last block
    works, state = [], {'next': 0, 'error': None}
    ext = None
    if exchange and dist.get_backend(group) == 'nccl' and getattr(engine, 'stream', 0):
        # collectives are ordered behind torch's CURRENT stream: make that the engine's stream while they are queued
        ext = torch.cuda.ExternalStream(engine.stream, device=slot.device)

    own_echo = []  # where the collective puts ITS copy of this rank's planes (kept alive until the works are done)

    def issue(e0: int, e1: int) -> None:
        if not exchange:
            return
        # The collective must not write into this rank's slot: a list all-gather lands in a flat buffer and is
        # copied out entry by entry on the backend's own stream - this rank's entry included - unordered
        # against the engine stream, on which a nanmedian replay may rewrite the slot with its final planes
        # after the exchange was started. The echo of this rank's planes goes to scratch; peers' pieces go
        # where they belong (as in the C form: sends and receives with peers only).
        echo = torch.empty_like(slot[e0:e1])
        own_echo.append(echo)
        outs = [echo if r == rank else gathered[r, e0:e1] for r in range(world)]
        if ext is not None:
            with torch.cuda.stream(ext):
                works.append(dist.all_gather(outs, slot[e0:e1], group=group, async_op=True))
        else:
            works.append(dist.all_gather(outs, slot[e0:e1], group=group, async_op=True))

    def progress(done: int) -> None:
        # planes [0, done) of the block have their kernels enqueued: start what has become ready
        while state['next'] < per_rank:
            e1 = min(state['next'] + step, per_rank)
            if min(e1, mine_n) > done:
                break
            issue(state['next'], e1)
            state['next'] = e1

    def on_chunk(first: int, n: int) -> None:
        try:
            if state['error'] is None:
                progress(first + n)
        except Exception as e:  # noqa: BLE001 - must not unwind through the C frames
            state['error'] = e

    import time

    t_begin = time.perf_counter()
    redone = 0
    if mine_n > 0:
        if chunked:
            engine.set_chunk_callback(on_chunk)
        try:
            if host_cube:
                engine.map_cube_host_to_device(local_cube[:mine_n], x_map, y_map, n0, n1, slot[:mine_n], interpolation, propagate_nan)
            else:
                engine.map_cube_device(local_cube[:mine_n], dtype, mine_n, x_map, y_map, n0, n1, slot[:mine_n], interpolation,
                                       propagate_nan)
            engine.synchronize()  # final: flag check / nanmedian replay
            redone = int(engine.last_redo_planes())
        except Exception as e:  # noqa: BLE001 - reported through the agreement
            if not exchange:
                raise
            state['error'] = state['error'] or e
        finally:
            if chunked:
                engine.set_chunk_callback(None)
    t_mapped = time.perf_counter()
    if stages is not None:
        stages.update({'map_call': (t_mapped - t_begin) * 1e3, 'exchange_exposed': 0.0, 'agreement': 0.0})
    if not exchange:
        if state['error'] is not None:
            raise state['error']
        return
    error = state['error']
    try:
        progress(per_rank)  # whatever has not been started yet
    except Exception as e:  # noqa: BLE001
        error = error or e
    for w in works:
        w.wait()
    if stages is not None:
        if slot.is_cuda:
            torch.cuda.synchronize(slot.device)
        stages['exchange_exposed'] = (time.perf_counter() - t_mapped) * 1e3
    t_exchanged = time.perf_counter()
    verdict = torch.tensor([0 if error is None else 1, 1 if (error is None and redone > 0) else 0], dtype=torch.int32,
                           device=_group_device(group))
    dist.all_reduce(verdict, op=dist.ReduceOp.SUM, group=group)
    n_failed, n_redo = (int(v) for v in verdict.tolist())
    if stages is not None:
        stages['agreement'] = (time.perf_counter() - t_exchanged) * 1e3
    if n_failed == 0 and n_redo > 0:
        # somebody's planes changed after they had been sent: everybody gathers the whole block again
        dist.all_gather([torch.empty_like(slot) if r == rank else gathered[r] for r in range(world)], slot, group=group)
    own_echo.clear()
    if error is not None:
        raise error
    if n_failed > 0:
        raise PeerFailedError(f'{n_failed} other rank(s) failed in this sharded call: the gathered cube is not valid')


class Comm:
    """
    RCCL communicator behind the C ABI (`pm_comm_*`, include/planetmapper_hip.h): for callers that
    shard a cube WITHOUT torch.distributed. Rank 0 makes the 128-byte id with `Comm.unique_id()`
    and hands it to the other ranks (file, socket, MPI ...); every rank then constructs
    `Comm(engine, world, rank, id)` - a collective call - and maps with `map_cube_sharded`.
    """

    def __init__(self, engine, world: int, rank: int, unique_id: bytes) -> None:
        import ctypes

        if len(unique_id) != 128:
            raise ValueError('unique_id must be the 128 bytes of Comm.unique_id()')
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        engine._check(engine._lib.pm_comm_create(engine._ctx, self.world, self.rank, buf, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from . import _lib

        buf = ctypes.create_string_buffer(128)
        rc = _lib.load().pm_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f'pm_comm_unique_id failed with status {rc} (is RCCL installed?)')
        return buf.raw

    def close(self) -> None:
        if getattr(self, '_h', None):
            self.engine._lib.pm_comm_destroy(self._h)
            self._h = None

    def map_cube_sharded(self, local_cube, dtype, n_planes_total: int, x_map, y_map, n0: int, n1: int, out_all,
                         interpolation='linear', propagate_nan=True, host_cube: bool = False, gather: bool = True) -> None:
        """
        `pm_map_cube_sharded`: this rank's planes `shard_bounds(n_planes_total, world, rank)` (a device
        tensor / pointer, or a numpy array with `host_cube=True`) are mapped into its block of
        `out_all` (device, world * per_rank * n0 * n1 doubles) and, with `gather`, all blocks are
        exchanged by one RCCL all-gather enqueued on the engine's stream.
        """
        from . import _lib
        from .engine import _ptr, dtype_code, interpolation_code

        eng = self.engine
        eng.set_spline_smoothing(0.0)
        eng._check(
            eng._lib.pm_map_cube_sharded(
                eng._ctx, self._h, _ptr(local_cube) if local_cube is not None else None, dtype_code(dtype),
                int(n_planes_total), _ptr(x_map), _ptr(y_map), int(n0), int(n1), interpolation_code(interpolation),
                1 if propagate_nan else 0, _ptr(out_all), _lib.PM_MEM_HOST_CUBE if host_cube else _lib.PM_MEM_DEVICE,
                1 if gather else 0,
            )
        )
