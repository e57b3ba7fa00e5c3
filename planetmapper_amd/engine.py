"""
`Engine`: a thin object wrapper over one `pm_ctx` (one GPU, one HIP stream).

This is the only module that calls into libplanetmapper_hip.so. The BodyXY /
Observation classes (the reference-compatible API) are built on it.
"""

from __future__ import annotations

import ctypes
import os
from typing import Iterable, Mapping

import numpy as np

from . import _lib
from ._lib import PLANE_INDEX, PLANE_NAMES, NUM_PLANES
from .geometry import PMDisc, PMGeometry

_DTYPE_CODES = {
    np.dtype('float64'): 0,
    np.dtype('float32'): 1,
    np.dtype('int16'): 2,
    np.dtype('int32'): 3,
    np.dtype('uint8'): 4,
    np.dtype('uint16'): 5,
}

_INTERP_CODES = {'nearest': _lib.PM_INTERP_NEAREST, 'linear': _lib.PM_INTERP_LINEAR, 1: _lib.PM_INTERP_LINEAR}


def interpolation_code(interpolation) -> int:
    """
    C-ABI code of a `map_img` interpolation (body_xy.py:1598-1630): 'nearest', 'linear',
    'quadratic', 'cubic', 'smooth', an int degree or a (k_rows, k_cols) tuple of spline degrees.
    """
    names = {'linear': 1, 'quadratic': 2, 'cubic': 3}
    if isinstance(interpolation, str) and interpolation in names:
        interpolation = names[interpolation]
    if interpolation == 'nearest':
        return _lib.PM_INTERP_NEAREST
    if interpolation == 'smooth':
        return _lib.PM_INTERP_SMOOTH
    if isinstance(interpolation, bool):
        raise ValueError(f'Unknown interpolation method {interpolation!r}')
    if isinstance(interpolation, int):
        interpolation = (interpolation, interpolation)
    if (
        isinstance(interpolation, tuple)
        and len(interpolation) == 2
        and all(isinstance(k, int) and 1 <= k <= 5 for k in interpolation)
    ):
        if interpolation == (1, 1):
            return _lib.PM_INTERP_LINEAR
        return 0x100 | (interpolation[0] << 4) | interpolation[1]
    raise ValueError(f'Unknown interpolation method {interpolation!r}')


def dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt not in _DTYPE_CODES:
        raise TypeError(f'unsupported cube dtype {dt}')
    return _DTYPE_CODES[dt]


def plane_mask(names: Iterable[str]) -> int:
    m = 0
    for n in names:
        m |= 1 << PLANE_INDEX[n]
    return m


def _ptr(x) -> int:
    """Address of a numpy array (host) / torch tensor (device) / raw int pointer."""
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, 'data_ptr'):
        return int(x.data_ptr())
    raise TypeError(f'cannot take the address of {type(x)!r}')


class Engine:
    """One GPU context of the HIP engine. Raises `NoDeviceError` without a gfx950 GPU."""

    def __init__(self, device: int = 0, *, general_kernel: bool | None = None) -> None:
        """
        `general_kernel=True` routes every image-plane request through the general kernel
        (`PM_OPT_GENERAL_KERNEL`) instead of the spheroid fast path; None keeps the library default.
        """
        self._lib = _lib.load()
        status = ctypes.c_int(0)
        self._ctx = self._lib.pm_create(int(device), ctypes.byref(status))
        if not self._ctx:
            if status.value == _lib.PM_ERR_NO_DEVICE:
                raise _lib.NoDeviceError(
                    f'no usable gfx950 (MI355X) device {device}: planetmapper_amd has no CPU '
                    'fallback'
                )
            raise _lib.EngineError(f'pm_create failed with status {status.value}')
        self.device = int(device)
        self._disc: PMDisc | None = None
        self._geometry: PMGeometry | None = None
        # image planes handed to callers of the drop-in surface, recycled once nobody looks at them any more (plane_buffer)
        self._plane_pool: dict[tuple, list] = {}
        self._plane_owned: dict[int, int] = {}  # id(array) -> bytes, of the arrays plane_buffer() allocated and that are alive
        self._plane_bytes = 0
        self._plane_limit = _plane_pool_limit()
        self._free_counts = self._calibrate_recycling()
        # device memory of invalidated DeviceArrays, kept for the next ones (PLANETMAPPER_DEVICE_POOL_MB, default 8 GiB of 288)
        self._device_pool: dict[int, list] = {}
        self._device_pool_bytes = 0
        self._device_waiting: list = []
        import weakref

        self._device_arrays = weakref.WeakSet()  # every live DeviceArray of this engine: close() takes their memory with it
        self._device_pool_limit = int(float(os.environ.get('PLANETMAPPER_DEVICE_POOL_MB', '8192')) * (1 << 20))
        if general_kernel is not None:
            self.set_option(_lib.PM_OPT_GENERAL_KERNEL, 1 if general_kernel else 0)

    # ------------------------------------------------------------------ plumbing
    def close(self) -> None:
        self._plane_pool = {}
        if getattr(self, '_ctx', None):
            self._device_sweep()
            for arr in list(self._device_arrays):
                arr._orphan()  # (freed now; one a consumer still imports: by that import's own deleter, when it lets go)
            self._device_waiting = []
            for blocks in self._device_pool.values():
                for ptr in blocks:
                    self._lib.pm_device_free(self._ctx, ctypes.c_void_p(ptr))
            self._device_pool, self._device_pool_bytes = {}, 0
        if getattr(self, '_ctx', None):
            self._lib.pm_destroy(self._ctx)
            self._ctx = None

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc == _lib.PM_OK:
            return
        msg = self._lib.pm_last_error(self._ctx).decode('utf-8', 'replace')
        if rc == _lib.PM_ERR_INVALID_ARGUMENT:
            raise ValueError(msg)
        if rc == _lib.PM_ERR_UNSUPPORTED:
            raise _lib.UnsupportedError(msg)
        if rc == _lib.PM_ERR_ALLOC:
            raise MemoryError(msg)
        if rc == _lib.PM_ERR_PEER:
            raise _lib.PeerFailedError(msg)
        raise _lib.EngineError(f'{msg} (status {rc})')

    def synchronize(self) -> None:
        self._check(self._lib.pm_synchronize(self._ctx))

    def set_option(self, option: int, value: int) -> None:
        """`pm_set_option`: `_lib.PM_OPT_*` (general kernel, host-path chunking / threads / zero copy)."""
        self._check(self._lib.pm_set_option(self._ctx, int(option), int(value)))

    def get_option(self, option: int) -> int:
        v = ctypes.c_int64(0)
        self._check(self._lib.pm_get_option(self._ctx, int(option), ctypes.byref(v)))
        return int(v.value)

    def last_stages_ms(self) -> dict[str, float]:
        """`PM_OPT_LAST_STAGE_NS`: where the latest host-fed `map_cube` / sharded call of this engine spent its time, in ms."""
        return {name: self.get_option(_lib.PM_OPT_LAST_STAGE_NS + k) * 1e-6 for k, name in enumerate(_lib.STAGES)}

    def last_redo_planes(self) -> int:
        """planes of the latest finished nearest / linear map_cube that were redone with their nanmedian"""
        return self.get_option(_lib.PM_OPT_LAST_REDO_PLANES)

    def set_chunk_callback(self, fn) -> None:
        """
        `pm_set_chunk_callback`: `fn(first_plane, n_planes)` is called on the calling thread, inside a
        nearest / linear `map_cube*` call, each time the kernels of further planes have been enqueued
        on the engine's stream (planes arrive in order); None removes it. An exception raised by `fn`
        cannot cross the C frames: catch inside `fn`.
        """
        if fn is None:
            self._check(self._lib.pm_set_chunk_callback(self._ctx, None, None))
            self._chunk_cb = None
            return
        cb = _lib.CHUNK_CALLBACK(lambda _user, first, n: fn(int(first), int(n)))
        self._check(self._lib.pm_set_chunk_callback(self._ctx, ctypes.cast(cb, ctypes.c_void_p), None))
        self._chunk_cb = cb  # (the C side holds a bare pointer: keep the thunk alive)

    @property
    def stream(self) -> int:
        """The context's `hipStream_t` as an integer."""
        return int(self._lib.pm_stream(self._ctx) or 0)

    def set_stream(self, hip_stream: int) -> None:
        self._check(self._lib.pm_set_stream(self._ctx, ctypes.c_void_p(hip_stream)))

    def pinned_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """
        An uninitialised numpy array in page-locked host memory (`pm_host_alloc`): host-buffer calls
        move such arrays by DMA at the full PCIe rate, and `map_cube` reads a pinned cube in place
        (`PM_OPT_ZERO_COPY`). The memory is released when the array (and every view of it) is gone;
        it must not outlive the engine.
        """
        import weakref

        dt = np.dtype(dtype)
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        p = ctypes.c_void_p()
        self._check(self._lib.pm_host_alloc(self._ctx, max(nbytes, 1), ctypes.byref(p)))
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        weakref.finalize(buf, Engine._release_pinned, weakref.ref(self), p.value)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    @staticmethod
    def _release_pinned(engine_ref, ptr: int) -> None:
        eng = engine_ref()
        ctx = getattr(eng, '_ctx', None) if eng is not None else None
        # (an array that outlives its engine - close() / garbage collection came first - is freed without a context)
        _lib.load().pm_host_free(ctx, ctypes.c_void_p(ptr))

    # ------------------------------------------------------------------ recycled result planes
    def plane_buffer(self, shape) -> np.ndarray:
        """
        A float64 array for a result plane: one that `recycle_plane` took back if there is one of this shape, else a new
        page-locked one while the pool's budget lasts (PLANETMAPPER_PLANE_POOL_MB, default min(8 GiB, RAM / 8)), else
        `np.empty`. A fresh pageable array costs the copy threads a page fault per 4 KiB they write - 6 of the 16 ms of
        the headline frame's five planes; a recycled pinned one takes its plane by DMA, in place (9 ms).
        """
        import weakref

        shape = tuple(int(v) for v in shape)
        free = self._plane_pool.get(shape)
        if free:
            return free.pop()
        nbytes = int(np.prod(shape, dtype=np.int64)) * 8
        if nbytes < (1 << 20) or self._plane_bytes + nbytes > self._plane_limit:
            return np.empty(shape, dtype=np.float64)
        arr = self.pinned_empty(shape)
        self._plane_owned[id(arr)] = nbytes
        self._plane_bytes += nbytes
        weakref.finalize(arr, Engine._forget_plane, weakref.ref(self), id(arr))
        return arr

    @staticmethod
    def _forget_plane(engine_ref, key: int) -> None:
        eng = engine_ref()
        if eng is not None:
            eng._plane_bytes -= eng._plane_owned.pop(key, 0)

    def recycle_plane(self, arr, _calibrate: bool = False):
        """
        Take a `plane_buffer` array back - if the caller's is the ONLY reference left to it and to the memory behind it
        (a view a user still holds keeps the owner's count up: then the array is simply left to the garbage collector,
        and whoever holds it keeps data nobody will overwrite). Call with exactly one reference of your own (a local).
        """
        import sys

        base = arr.base
        counts = (sys.getrefcount(arr), sys.getrefcount(base) if base is not None else 0)
        if _calibrate:
            return counts
        if id(arr) not in self._plane_owned or counts != self._free_counts:
            return False
        self._plane_pool.setdefault(arr.shape, []).append(arr)
        return True

    def _calibrate_recycling(self):
        # the counts `recycle_plane` sees for an array nobody else refers to, measured through the same calls ...
        probe = np.frombuffer(bytearray(16), dtype=np.float64).reshape((2,))
        free = self.recycle_plane(probe, _calibrate=True)
        # ... and a NEGATIVE check: an array somebody holds (directly, or through a view) must count differently - on an
        # interpreter where it does not (free-threaded builds, borrowed-reference loads) or that is not CPython the pool
        # is switched off: recycling an array a user still looks at would overwrite their data
        import platform

        held = probe
        viewed = np.frombuffer(bytearray(16), dtype=np.float64).reshape((2,))
        view = viewed[:1]
        sound = (platform.python_implementation() == 'CPython' and self.recycle_plane(probe, _calibrate=True) != free
                 and self.recycle_plane(viewed, _calibrate=True) != free)  # fmt: skip
        del held, view
        if not sound:
            self._plane_limit = 0
        return free

    # ------------------------------------------------------------------ results that stay on the device
    def device_array(self, shape, dtype=np.float64):
        """A `DeviceArray` (planetmapper_amd/device_array.py) of this engine: the memory of a result that stays in HBM."""
        from .device_array import DeviceArray  # noqa: PLC0415

        return DeviceArray(self, shape, dtype)

    def _device_take(self, nbytes: int) -> int:
        """device memory for a DeviceArray: a block `_device_give` took back if one of this size is there (a cold getter after
        a disc change finds the planes of the frame before it: no hipMalloc - 0.1-1 ms per 134 MB plane - in the call)"""
        self._device_sweep()
        free = self._device_pool.get(int(nbytes))
        if free:
            self._device_pool_bytes -= int(nbytes)
            return free.pop()
        return self.device_malloc(max(int(nbytes), 1))

    def _device_wait(self, arr) -> None:
        """an invalidated DeviceArray a DLPack consumer still imports: its memory is taken back once the consumer lets go"""
        if arr not in self._device_waiting:
            self._device_waiting.append(arr)

    def _device_sweep(self) -> None:
        if self._device_waiting:
            waiting, self._device_waiting = self._device_waiting, []
            for arr in waiting:
                arr._settle()  # (back into the list if still imported)

    def _device_give(self, ptr: int, nbytes: int) -> None:
        if not getattr(self, '_ctx', None):
            return  # (the context took its memory with it)
        if self._device_pool_bytes + nbytes <= self._device_pool_limit:
            self._device_pool.setdefault(int(nbytes), []).append(ptr)
            self._device_pool_bytes += int(nbytes)
        else:
            self.device_free(ptr)

    def pinned_copy(self, arr) -> np.ndarray:
        """`arr` copied into a new pinned array (see `pinned_empty`)."""
        arr = np.asarray(arr)
        out = self.pinned_empty(arr.shape, arr.dtype)
        out[...] = arr
        return out

    def device_malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._check(self._lib.pm_device_malloc(self._ctx, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def device_free(self, dptr: int) -> None:
        self._check(self._lib.pm_device_free(self._ctx, ctypes.c_void_p(dptr)))

    def h2d(self, dptr: int, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr)
        self._check(self._lib.pm_memcpy_h2d(self._ctx, ctypes.c_void_p(dptr), arr.ctypes.data, arr.nbytes))

    def d2h(self, arr: np.ndarray, dptr: int) -> None:
        assert arr.flags.c_contiguous
        self._check(self._lib.pm_memcpy_d2h(self._ctx, arr.ctypes.data, ctypes.c_void_p(dptr), arr.nbytes))

    # ------------------------------------------------------------------ state
    def set_geometry(self, g: PMGeometry) -> None:
        self._check(self._lib.pm_set_geometry(self._ctx, ctypes.byref(g)))
        self._geometry = g

    def set_disc(self, x0, y0, r0, rotation_rad, nx, ny, optimize_speed=True) -> None:
        d = PMDisc()
        d.x0, d.y0, d.r0, d.rotation_rad = float(x0), float(y0), float(r0), float(rotation_rad)
        d.nx, d.ny = int(nx), int(ny)
        d.optimize_speed = 1 if optimize_speed else 0
        self._check(self._lib.pm_set_disc(self._ctx, ctypes.byref(d)))
        self._disc = d

    # ------------------------------------------------------------------ image backplanes
    def backplanes_img(self, names: Iterable[str], alt: float = 0.0, *, recycled: bool = False) -> dict[str, np.ndarray]:
        """Compute image-space backplanes into fresh host arrays of shape (ny, nx) (`recycled`: into `plane_buffer`s)."""
        names = list(names)
        assert self._disc is not None
        shape = (self._disc.ny, self._disc.nx)
        if shape[0] <= 0 or shape[1] <= 0:
            raise ValueError('nx and ny must be positive to create a backplane image')
        outs = {n: (self.plane_buffer(shape) if recycled else np.empty(shape, dtype=np.float64)) for n in names}
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = a.ctypes.data
        self._check(
            self._lib.pm_backplanes_img(self._ctx, plane_mask(names), float(alt), ptrs, _lib.PM_MEM_HOST)
        )
        return outs

    def backplanes_img_device(self, outs: Mapping[str, object], alt: float = 0.0) -> None:
        """Enqueue image backplanes into device buffers (`name -> pointer / torch tensor`)."""
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = _ptr(a)
        self._check(
            self._lib.pm_backplanes_img(self._ctx, plane_mask(outs.keys()), float(alt), ptrs, _lib.PM_MEM_DEVICE)
        )

    def backplanes_img_rows(self, names: Iterable[str], row_begin: int, n_rows: int, alt: float = 0.0):
        """Rows [row_begin, row_begin + n_rows) of the image backplanes: arrays of shape (n_rows, nx)."""
        names = list(names)
        assert self._disc is not None
        if self._disc.ny <= 0 or self._disc.nx <= 0:
            raise ValueError('nx and ny must be positive to create a backplane image')
        outs = {n: np.empty((int(n_rows), self._disc.nx), dtype=np.float64) for n in names}
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = a.ctypes.data
        self._check(
            self._lib.pm_backplanes_img_rows(
                self._ctx, plane_mask(names), float(alt), int(row_begin), int(n_rows), ptrs, _lib.PM_MEM_HOST
            )
        )
        return outs

    def backplanes_img_rows_device(self, outs: Mapping[str, object], row_begin: int, n_rows: int,
                                   alt: float = 0.0) -> None:
        """Enqueue a row block of the image backplanes into device buffers of n_rows * nx doubles."""
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = _ptr(a)
        self._check(
            self._lib.pm_backplanes_img_rows(
                self._ctx, plane_mask(outs.keys()), float(alt), int(row_begin), int(n_rows), ptrs, _lib.PM_MEM_DEVICE
            )
        )

    # ------------------------------------------------------------------ map space
    def backplanes_map(self, names, lon_deg, lat_deg, alt: float = 0.0) -> dict[str, np.ndarray]:
        names = list(names)
        lon = np.ascontiguousarray(lon_deg, dtype=np.float64)
        lat = np.ascontiguousarray(lat_deg, dtype=np.float64)
        if lon.shape != lat.shape or lon.ndim != 2:
            raise ValueError('lon/lat grids must be 2D arrays of the same shape')
        n0, n1 = lon.shape
        outs = {n: np.empty((n0, n1), dtype=np.float64) for n in names}
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = a.ctypes.data
        self._check(
            self._lib.pm_backplanes_map(
                self._ctx, plane_mask(names), lon.ctypes.data, lat.ctypes.data, n0, n1, float(alt), ptrs,
                _lib.PM_MEM_HOST,
            )
        )
        return outs

    def backplanes_map_device(self, outs: Mapping[str, object], lon, lat, n0: int, n1: int, alt: float = 0.0) -> None:
        """Enqueue map-space backplanes of a lon / lat grid resident in HBM into device buffers (`name -> pointer / tensor`)."""
        ptrs = (ctypes.c_void_p * NUM_PLANES)()
        for n, a in outs.items():
            ptrs[PLANE_INDEX[n]] = _ptr(a)
        self._check(
            self._lib.pm_backplanes_map(
                self._ctx, plane_mask(outs.keys()), _ptr(lon), _ptr(lat), int(n0), int(n1), float(alt), ptrs, _lib.PM_MEM_DEVICE
            )
        )

    def xy_map(self, lon_deg, lat_deg, alt: float = 0.0):
        o = self.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon_deg, lat_deg, alt)
        return o['PIXEL-X'], o['PIXEL-Y']

    def xy_map_device(self, lon, lat, n0: int, n1: int, x_map, y_map, alt: float = 0.0) -> None:
        self._check(
            self._lib.pm_xy_map(
                self._ctx, _ptr(lon), _ptr(lat), int(n0), int(n1), float(alt), _ptr(x_map), _ptr(y_map),
                _lib.PM_MEM_DEVICE,
            )
        )

    # ------------------------------------------------------------------ point transforms
    def transform(self, src: str, dst: str, a, b, *, alt: float = 0.0, not_visible_nan=False, planetocentric=False):
        """
        Array-valued coordinate transform between 'xy', 'radec', 'angular', 'km' and 'lonlat'
        (inputs are broadcast together; returns two arrays of the broadcast shape).
        """
        a, b = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
        shape = a.shape
        a = np.ascontiguousarray(a).ravel()
        b = np.ascontiguousarray(b).ravel()
        oa = np.empty_like(a)
        ob = np.empty_like(b)
        flags = (_lib.PM_TF_NOT_VISIBLE_NAN if not_visible_nan else 0) | (
            _lib.PM_TF_PLANETOCENTRIC if planetocentric else 0
        )
        self._check(
            self._lib.pm_transform(
                self._ctx, _lib.COORDS[src], _lib.COORDS[dst], a.size, a.ctypes.data, b.ctypes.data, float(alt),
                flags, oa.ctypes.data, ob.ctypes.data, _lib.PM_MEM_HOST,
            )
        )
        return oa.reshape(shape), ob.reshape(shape)

    def radec_query(self, ra, dec, *, alt: float = 0.0, ring_only_visible: bool = True) -> np.ndarray:
        """
        For sky points (broadcast together): an (8,) + shape array of planetographic lon / lat of
        the intercept, ring-plane radius / longitude / distance and limb longitude / latitude /
        distance (Body.radec2lonlat, ring_plane_coordinates, limb_coordinates_from_radec).
        """
        ra, dec = np.broadcast_arrays(np.asarray(ra, dtype=np.float64), np.asarray(dec, dtype=np.float64))
        shape = ra.shape
        ra = np.ascontiguousarray(ra).ravel()
        dec = np.ascontiguousarray(dec).ravel()
        out = np.empty((8, ra.size), dtype=np.float64)
        self._check(
            self._lib.pm_radec_query(
                self._ctx, ra.size, ra.ctypes.data, dec.ctypes.data, float(alt), 1 if ring_only_visible else 0,
                out.ctypes.data, _lib.PM_MEM_HOST,
            )
        )
        return out.reshape((8,) + shape)

    # ------------------------------------------------------------------ reprojection
    def set_smooth_options(self, oversample_by: int = 5, max_oversampled_img_size: int = 10_000) -> None:
        """`smooth_oversample_by` / `smooth_max_oversampled_img_size` of map_img (body_xy.py:1427-1428)"""
        clip = lambda v: int(max(-(2**31), min(2**31 - 1, int(v))))  # noqa: E731
        self._check(self._lib.pm_set_smooth_options(self._ctx, clip(oversample_by), clip(max_oversampled_img_size)))

    def set_spline_smoothing(self, s: float = 0.0) -> None:
        """`spline_smoothing` of map_img (FITPACK `s`), used by 'linear' and the spline degrees"""
        self._check(self._lib.pm_set_spline_smoothing(self._ctx, float(s)))

    def map_cube(self, cube: np.ndarray, x_map, y_map, interpolation='linear', propagate_nan=True, *,
                 smooth_oversample_by: int = 5, smooth_max_oversampled_img_size: int = 10_000,
                 spline_smoothing: float = 0.0) -> np.ndarray:
        """Reproject host cube (P, ny, nx) [or one (ny, nx) image] -> (P, n0, n1) float64."""
        code = interpolation_code(interpolation)
        self.set_spline_smoothing(spline_smoothing)
        if code == _lib.PM_INTERP_SMOOTH:
            self.set_smooth_options(smooth_oversample_by, smooth_max_oversampled_img_size)
        cube = np.asarray(cube)
        if cube.dtype.byteorder not in ('=', '|') and cube.dtype.byteorder != ('<' if np.little_endian else '>'):
            cube = cube.astype(cube.dtype.newbyteorder('='))
        if cube.dtype not in _DTYPE_CODES:
            cube = cube.astype(np.float64)
        cube = np.ascontiguousarray(cube)
        if cube.ndim == 2:
            cube = cube[None]
        assert self._disc is not None
        if cube.shape[1:] != (self._disc.ny, self._disc.nx):
            raise ValueError(
                f'The input `img` shape {cube.shape[1:]!r} is inconsistent with the body\'s image size '
                f'(ny={self._disc.ny}, nx={self._disc.nx})'
            )
        xm = np.ascontiguousarray(x_map, dtype=np.float64)
        ym = np.ascontiguousarray(y_map, dtype=np.float64)
        n0, n1 = xm.shape
        out = np.empty((cube.shape[0], n0, n1), dtype=np.float64)
        self._check(
            self._lib.pm_map_cube(
                self._ctx, cube.ctypes.data, dtype_code(cube.dtype), cube.shape[0], xm.ctypes.data,
                ym.ctypes.data, n0, n1, code, 1 if propagate_nan else 0,
                out.ctypes.data, _lib.PM_MEM_HOST,
            )
        )
        return out

    def map_cube_device(
        self, cube, dtype, n_planes: int, x_map, y_map, n0: int, n1: int, out,
        interpolation='linear', propagate_nan=True, spline_smoothing: float = 0.0,
    ) -> None:  # fmt: skip
        """
        Enqueue the reprojection of a device-resident cube into a device output. 'nearest' / 'linear' return with the work
        on the stream (finish with `synchronize()`). The spline degrees BLOCK the calling thread once per 2 GiB chunk of
        planes (the read-back that says whether any plane needs its nanmedian), 'smooth' blocks on the map-limits read-back,
        `spline_smoothing > 0` once per round of the knot search: host work placed behind such a call does not overlap it.
        """
        self.set_spline_smoothing(spline_smoothing)
        self._check(
            self._lib.pm_map_cube(
                self._ctx, _ptr(cube), dtype_code(dtype), int(n_planes), _ptr(x_map), _ptr(y_map), int(n0),
                int(n1), interpolation_code(interpolation), 1 if propagate_nan else 0, _ptr(out), _lib.PM_MEM_DEVICE,
            )
        )


    def map_cube_host_to_device(
        self, cube: np.ndarray, x_map, y_map, n0: int, n1: int, out, interpolation='linear', propagate_nan=True,
    ) -> None:  # fmt: skip
        """
        Reproject a HOST cube (numpy; pinned arrays from `pinned_empty` are gathered in place, others
        are copied through the pipelined path) into a DEVICE output with device x/y maps
        (`PM_MEM_HOST_CUBE`): the per-rank step of a plane-sharded cube, whose mapped planes feed an
        all-gather. Asynchronous for pinned cubes (finish with `synchronize()`).
        """
        self.set_spline_smoothing(0.0)
        assert cube.flags.c_contiguous and cube.ndim == 3
        self._check(
            self._lib.pm_map_cube(
                self._ctx, cube.ctypes.data, dtype_code(cube.dtype), int(cube.shape[0]), _ptr(x_map), _ptr(y_map),
                int(n0), int(n1), interpolation_code(interpolation), 1 if propagate_nan else 0, _ptr(out),
                _lib.PM_MEM_HOST_CUBE,
            )
        )


    def mapped_data_device(self, cube, dtype, n_planes: int, lon, lat, n0: int, n1: int, x_map, y_map, out,
                           interpolation='linear', propagate_nan=True, alt: float = 0.0) -> None:
        """
        `pm_mapped_data` on device buffers: x/y map of the lon/lat grid and reprojection of the planes
        in one call (one launch for up to 8 planes, nearest / linear) - Observation.get_mapped_data.
        """
        self.set_spline_smoothing(0.0)
        self._check(
            self._lib.pm_mapped_data(
                self._ctx, _ptr(cube), dtype_code(dtype), int(n_planes), _ptr(lon), _ptr(lat), int(n0), int(n1),
                float(alt), interpolation_code(interpolation), 1 if propagate_nan else 0, _ptr(x_map), _ptr(y_map),
                _ptr(out), _lib.PM_MEM_DEVICE,
            )
        )


def _plane_pool_limit() -> int:
    import os

    env = os.environ.get('PLANETMAPPER_PLANE_POOL_MB')
    if env is not None:
        return max(0, int(float(env) * (1 << 20)))
    try:
        ram = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')
    except (ValueError, OSError):
        ram = 8 << 30
    # (every rank of a node keeps a pool of its own: the default is the node's budget divided among them)
    ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))
    return int(min(8 << 30, ram // 8) // ranks)


def device_count() -> int:
    return int(_lib.load().pm_device_count())


__all__ = ['Engine', 'device_count', 'plane_mask', 'PLANE_NAMES', 'PLANE_INDEX']
