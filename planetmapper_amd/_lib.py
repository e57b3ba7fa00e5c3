"""
ctypes binding of libplanetmapper_hip.so (C ABI: include/planetmapper_hip.h).

The library is built in-tree by ``planetmapper_amd/csrc/Makefile`` (see
``__graft_entry__.build``). There is no fallback: if the shared object is missing, or no
gfx950 device is present when a context is created, an exception is raised.
"""

from __future__ import annotations

import ctypes
import os

from .geometry import PMDisc, PMGeometry

LIB_NAME = 'libplanetmapper_hip.so'
# (PLANETMAPPER_HIP_LIB: another build of the library, e.g. for an A/B timing run)
LIB_PATH = os.environ.get('PLANETMAPPER_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

NUM_PLANES = 26
PLANE_NAMES = (
    'LON-GRAPHIC', 'LAT-GRAPHIC', 'LON-CENTRIC', 'LAT-CENTRIC', 'RA', 'DEC',
    'PIXEL-X', 'PIXEL-Y', 'KM-X', 'KM-Y', 'ANGULAR-X', 'ANGULAR-Y',
    'PHASE', 'INCIDENCE', 'EMISSION', 'AZIMUTH', 'LOCAL-SOLAR-TIME',
    'DISTANCE', 'RADIAL-VELOCITY', 'DOPPLER',
    'LIMB-DISTANCE', 'LIMB-LON-GRAPHIC', 'LIMB-LAT-GRAPHIC',
    'RING-RADIUS', 'RING-LON-GRAPHIC', 'RING-DISTANCE',
)  # fmt: skip
PLANE_INDEX = {n: i for i, n in enumerate(PLANE_NAMES)}

PM_OK = 0
PM_ERR_INVALID_ARGUMENT = -1
PM_ERR_NO_DEVICE = -2
PM_ERR_HIP = -3
PM_ERR_STATE = -4
PM_ERR_ALLOC = -5
PM_ERR_UNSUPPORTED = -6
PM_ERR_PEER = -7

PM_MEM_HOST = 0
PM_MEM_DEVICE = 1
PM_MEM_HOST_CUBE = 2

COORDS = {'xy': 0, 'radec': 1, 'angular': 2, 'km': 3, 'lonlat': 4}
PM_TF_NOT_VISIBLE_NAN = 1
PM_TF_PLANETOCENTRIC = 2

PM_INTERP_NEAREST = 0
PM_INTERP_LINEAR = 1
PM_INTERP_SMOOTH = 2

# every symbol declared in include/planetmapper_hip.h
EXPORTS = (
    'pm_abi_version', 'pm_device_count', 'pm_create', 'pm_destroy', 'pm_last_error',
    'pm_synchronize', 'pm_stream', 'pm_set_stream', 'pm_device_malloc', 'pm_device_free',
    'pm_memcpy_h2d', 'pm_memcpy_d2h', 'pm_set_geometry', 'pm_set_disc',
    'pm_backplanes_img', 'pm_xy_map', 'pm_backplanes_map', 'pm_map_cube', 'pm_transform',
    'pm_set_smooth_options', 'pm_backplanes_img_rows', 'pm_set_spline_smoothing', 'pm_radec_query',
    'pm_set_option', 'pm_get_option', 'pm_host_alloc', 'pm_host_free', 'pm_host_register',
    'pm_host_unregister', 'pm_shard_bounds', 'pm_comm_unique_id', 'pm_comm_create', 'pm_comm_destroy',
    'pm_map_cube_sharded', 'pm_mapped_data', 'pm_exchange_planes', 'pm_set_chunk_callback',
    'pm_dlpack_hold_create', 'pm_dlpack_export', 'pm_dlpack_exports', 'pm_dlpack_release', 'pm_dlpack_delete',
)  # fmt: skip

PM_OPT_GENERAL_KERNEL = 1
PM_OPT_HOST_CHUNK_BYTES = 2
PM_OPT_HOST_COPY_THREADS = 3
PM_OPT_ZERO_COPY = 4
PM_OPT_HOST_CUBE_ROUTE = 4  # the same option under the name that says what it selects
PM_OPT_LAST_DISC_KERNEL = 5
PM_OPT_SPARSE_FRAME = 6
PM_OPT_BLOCK_TABLE_CACHE = 7
PM_OPT_BLOCK_TABLE_HITS = 8
PM_OPT_ROUTE_EXPLORE = 9
PM_OPT_LAST_CUBE_ROUTE = 10
PM_OPT_LAST_REDO_PLANES = 11
PM_OPT_FUSE_PLANES = 12
PM_OPT_HOST_COPY_THREADS_IN_USE = 13
PM_OPT_HYBRID_FETCH_PERMILLE = 14
PM_OPT_FETCH_BLOCK_BYTES = 15
PM_OPT_LT_MODE = 24
PM_OPT_TRACE = 25
PM_OPT_SM_BATCH_PLANES = 26
PM_OPT_LAST_LT_PATH = 27
PM_OPT_SPLINE_SEGMENT = 28
PM_OPT_LAST_SPLINE_SEGMENT = 29
PM_OPT_LAST_SM_KNIFE_EDGES = 30
PM_OPT_LAST_SM_ILL_CONDITIONED = 31
PM_OPT_ROUTE_NS_PER_PLANE = 16  # + route 0..4
PM_OPT_LAST_STAGE_NS = 32  # + stage 0..12 (STAGES)
STAGES = ('total', 'tables', 'plan', 'first_fill', 'collect', 'issue', 'drain', 'finish', 'dma_device', 'kernels_device',
          'exchange_exposed', 'agreement', 'sharded_total')
NUM_CUBE_ROUTES = 5


# pm_chunk_callback: void (*)(void *user, int first_plane, int n_planes)
CHUNK_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)


class LibraryNotBuiltError(ImportError):
    pass


class NoDeviceError(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


class UnsupportedError(NotImplementedError):
    pass


class PeerFailedError(EngineError):
    """A sharded call failed on ANOTHER rank: the gathered result is not valid on this one either."""


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library, declaring argument types. Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryNotBuiltError(
            f'{LIB_PATH} not found: build it with `make -C planetmapper_amd/csrc` '
            '(or `python -c "import __graft_entry__ as g; g.build()"`). '
            'planetmapper_amd has no CPU fallback.'
        )
    # PyTorch wheels carry their own libamdhip64 / libhsa-runtime64. A process that loads the
    # system's HIP runtime first (through this library) and torch's afterwards ends up with two
    # runtimes, and the second one finds no GPU ("No HIP GPUs are available"). Importing torch
    # first makes its copy the one this library binds to as well, whatever the caller's import order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    dp = ctypes.POINTER(ctypes.c_double)
    dpp = ctypes.POINTER(ctypes.c_void_p)
    c_int = ctypes.c_int
    lib.pm_abi_version.restype = c_int
    lib.pm_device_count.restype = c_int
    lib.pm_create.restype = vp
    lib.pm_create.argtypes = [c_int, ctypes.POINTER(c_int)]
    lib.pm_destroy.restype = None
    lib.pm_destroy.argtypes = [vp]
    lib.pm_last_error.restype = ctypes.c_char_p
    lib.pm_last_error.argtypes = [vp]
    lib.pm_synchronize.argtypes = [vp]
    lib.pm_stream.restype = vp
    lib.pm_stream.argtypes = [vp]
    lib.pm_set_stream.argtypes = [vp, vp]
    lib.pm_device_malloc.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp)]
    lib.pm_device_free.argtypes = [vp, vp]
    lib.pm_dlpack_hold_create.restype = vp
    lib.pm_dlpack_hold_create.argtypes = [vp, c_int]
    lib.pm_dlpack_export.restype = vp
    lib.pm_dlpack_export.argtypes = [vp, c_int, c_int, c_int, vp]
    lib.pm_dlpack_exports.restype = ctypes.c_int64
    lib.pm_dlpack_exports.argtypes = [vp]
    lib.pm_dlpack_release.restype = c_int
    lib.pm_dlpack_release.argtypes = [vp, c_int]
    lib.pm_dlpack_delete.restype = None
    lib.pm_dlpack_delete.argtypes = [vp]
    lib.pm_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_uint64]
    lib.pm_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_uint64]
    lib.pm_set_geometry.argtypes = [vp, ctypes.POINTER(PMGeometry)]
    lib.pm_set_disc.argtypes = [vp, ctypes.POINTER(PMDisc)]
    lib.pm_backplanes_img.argtypes = [vp, ctypes.c_uint64, ctypes.c_double, dpp, c_int]
    lib.pm_xy_map.argtypes = [vp, vp, vp, c_int, c_int, ctypes.c_double, vp, vp, c_int]
    lib.pm_backplanes_map.argtypes = [
        vp, ctypes.c_uint64, vp, vp, c_int, c_int, ctypes.c_double, dpp, c_int,
    ]  # fmt: skip
    lib.pm_transform.argtypes = [
        vp, c_int, c_int, ctypes.c_uint64, vp, vp, ctypes.c_double, c_int, vp, vp, c_int,
    ]  # fmt: skip
    lib.pm_radec_query.argtypes = [vp, ctypes.c_uint64, vp, vp, ctypes.c_double, c_int, vp, c_int]
    lib.pm_map_cube.argtypes = [
        vp, vp, c_int, c_int, vp, vp, c_int, c_int, c_int, c_int, vp, c_int,
    ]  # fmt: skip
    lib.pm_mapped_data.argtypes = [
        vp, vp, c_int, c_int, vp, vp, c_int, c_int, ctypes.c_double, c_int, c_int, vp, vp, vp, c_int,
    ]  # fmt: skip
    lib.pm_set_smooth_options.argtypes = [vp, c_int, c_int]
    lib.pm_set_spline_smoothing.argtypes = [vp, ctypes.c_double]
    lib.pm_backplanes_img_rows.argtypes = [vp, ctypes.c_uint64, ctypes.c_double, c_int, c_int, dpp, c_int]
    lib.pm_set_option.argtypes = [vp, c_int, ctypes.c_int64]
    lib.pm_get_option.argtypes = [vp, c_int, ctypes.POINTER(ctypes.c_int64)]
    lib.pm_host_alloc.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp)]
    lib.pm_host_free.argtypes = [vp, vp]
    lib.pm_host_register.argtypes = [vp, vp, ctypes.c_uint64]
    lib.pm_host_unregister.argtypes = [vp, vp]
    ip = ctypes.POINTER(c_int)
    lib.pm_shard_bounds.argtypes = [c_int, c_int, c_int, ip, ip, ip]
    lib.pm_exchange_planes.argtypes = [c_int, c_int, c_int]
    lib.pm_set_chunk_callback.argtypes = [vp, vp, vp]
    lib.pm_comm_unique_id.argtypes = [vp]
    lib.pm_comm_create.argtypes = [vp, c_int, c_int, vp, ctypes.POINTER(vp)]
    lib.pm_comm_destroy.argtypes = [vp]
    lib.pm_map_cube_sharded.argtypes = [
        vp, vp, vp, c_int, c_int, vp, vp, c_int, c_int, c_int, c_int, vp, c_int, c_int,
    ]  # fmt: skip
    del dp
    if lib.pm_abi_version() != 3:
        raise ImportError('libplanetmapper_hip.so ABI version mismatch')
    _lib = lib
    return lib
