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


def get_mapped_data_sharded(
    obs,
    interpolation='linear',
    *,
    propagate_nan: bool = True,
    group=None,
    **map_kwargs,
) -> np.ndarray:
    """
    `Observation.get_mapped_data` with the planes of `obs.data` sharded over the ranks
    of `group` (default: the world). Every rank must hold the same `obs` (same data,
    disc parameters and map arguments) and receives the full (P, n0, n1) float64 result.
    """
    import torch.distributed as dist

    if not dist.is_initialized():
        return obs.get_mapped_data(interpolation, propagate_nan=propagate_nan, **map_kwargs)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_planes = obs.data.shape[0]
    start, stop, per_rank = shard_bounds(n_planes, world, rank)
    x_map = obs.get_x_map(**map_kwargs)
    n0, n1 = x_map.shape
    local = np.full((per_rank, n0, n1), np.nan)
    if stop > start:
        local[: stop - start] = obs.map_img(
            obs.data[start:stop], interpolation=interpolation, propagate_nan=propagate_nan, **map_kwargs
        )
    if n_planes == 0:
        return np.empty((0, n0, n1))
    return all_gather_planes(local, n_planes, group)


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
                            async_op: bool = False, previous=None):
    """
    Device-resident variant used by the benchmark: this rank's `n_planes_local` planes
    (`cube`, a device tensor / pointer) are mapped straight into its slot of `gathered`
    ((world, n_planes_local, n0, n1) float64 cuda tensor) and the slots are exchanged with
    one in-place RCCL all-gather. Nothing touches the host.

    `async_op=True` returns the collective's work handle instead of waiting for it, so the
    caller can enqueue independent kernels (the next frame's backplanes) while the gather is
    in flight over xGMI; pass that handle back as `previous` on the next call: it is waited for
    right before this rank's slot is overwritten.
    """
    import torch.distributed as dist

    if previous is not None:
        previous.wait()
    mine = gathered[rank]
    engine.map_cube_device(cube, dtype, n_planes_local, x_map, y_map, n0, n1, mine, interpolation, propagate_nan)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), group=group, async_op=async_op)
    return None
