"""
`DeviceArray`: a result of the engine that stays where it was computed - a C-contiguous array in the GPU's HBM, handed
to GPU-side consumers through the two interchange protocols they read: `__dlpack__` / `__dlpack_device__` (torch.from_dlpack,
cupy.from_dlpack, jax ...) and `__cuda_array_interface__` (CuPy, Numba; version 3, the read-only flag set).

What `BodyXY.get_*_img(device=True)`, `get_*_map(device=True, **map_kwargs)`, `get_backplane_img / _map(name, device=True)`
and `Observation.get_mapped_data(..., device=True)` return. The numpy forms of those getters move every plane over PCIe (9.4 ms for the five planes of a
4096^2 frame whose kernel is 0.13 ms); a consumer that works on the GPU takes the planes where they are instead.

Semantics (the device form of the reference's read-only cached arrays, base.py:115-138 / body_xy.py:2586-2630):
  * read-only by contract: the memory is the cache entry itself - the next getter returns the same array;
  * valid until the cache entry goes (`_clear_cache`: a new disc, a new image size): `invalidate()` then makes every later
    access raise `ValueError`. Memory a consumer already imported through DLPack is not pulled from under it - it is kept
    until the consumer's deleter has run, then goes back to the engine's pool of device arrays.
"""

from __future__ import annotations

import ctypes
import sys

import numpy as np

from . import _lib

_KDLROCM = 10  # DLDeviceType::kDLROCM

# The DLManagedTensor of an export and its deleter are C (pm_dlpack_export, planetmapper_amd/csrc/pm_capi.hip): a consumer
# that lets go of an import - on any thread, at interpreter shutdown - calls into no Python. What is Python here is the
# capsule's destructor, which matters only for a capsule NOBODY consumed; its thunk is made immortal (a capsule may be
# collected after this module's globals).
_CAPSULE_DESTRUCTOR = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
_api = ctypes.pythonapi
_api.PyCapsule_New.restype = ctypes.py_object
_api.PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, _CAPSULE_DESTRUCTOR]
_api.PyCapsule_IsValid.restype = ctypes.c_int
_api.PyCapsule_IsValid.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
_api.PyCapsule_GetPointer.restype = ctypes.c_void_p
_api.PyCapsule_GetPointer.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
_NAME = ctypes.create_string_buffer(b'dltensor')  # (PyCapsule_New keeps the POINTER to the name)


def _capsule_destructor_py(capsule) -> None:
    try:
        if sys.is_finalizing():
            return
        # a capsule nobody consumed still carries the name 'dltensor' (consumers rename it): its tensor is ours to delete
        if _api.PyCapsule_IsValid(capsule, _NAME):
            _lib.load().pm_dlpack_delete(_api.PyCapsule_GetPointer(capsule, _NAME))
    except Exception:  # noqa: BLE001
        pass


_capsule_destructor = _CAPSULE_DESTRUCTOR(_capsule_destructor_py)
_api.Py_IncRef.argtypes = [ctypes.py_object]
_api.Py_IncRef(_capsule_destructor)  # immortal: see above

_DL_CODES = {'f': 2, 'i': 0, 'u': 1}


class DeviceArray:
    """A C-contiguous array in an engine's HBM (see the module docstring). Made by `Engine.device_array`."""

    def __init__(self, engine, shape, dtype=np.float64) -> None:
        self._engine = engine
        self.shape = tuple(int(v) for v in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._ptr = engine._device_take(self.nbytes)
        self._hold = _lib.load().pm_dlpack_hold_create(ctypes.c_void_p(self._ptr), int(engine.device))
        self._valid = True
        engine._device_arrays.add(self)

    # ------------------------------------------------------------------ lifetime
    @property
    def valid(self) -> bool:
        return self._valid

    @property
    def _exports(self) -> int:
        """DLPack imports of this array that are alive (counted by the C side)"""
        return int(_lib.load().pm_dlpack_exports(ctypes.c_void_p(self._hold))) if self._hold else 0

    def _check(self) -> None:
        if not self._valid:
            raise ValueError('this device array belonged to a cache entry that has been cleared (the disc or the image size '
                             'changed): ask the getter again')  # fmt: skip

    @property
    def ptr(self) -> int:
        """the device address (valid arrays only)"""
        self._check()
        return self._ptr

    def data_ptr(self) -> int:  # (what Engine's own entry points take: planetmapper_amd.engine._ptr)
        return self.ptr

    def invalidate(self) -> None:
        """the cache entry is gone: later accesses raise; the memory returns to the engine once no DLPack import holds it"""
        if self._valid:
            self._valid = False
            self._settle()

    def _settle(self) -> bool:
        """give the memory back to the engine's pool if nobody imports it any more; else the engine asks again later"""
        if not self._ptr:
            return True
        if self._exports == 0:
            ptr, hold, self._ptr, self._hold = self._ptr, self._hold, 0, None
            _lib.load().pm_dlpack_release(ctypes.c_void_p(hold), 0)
            self._engine._device_give(ptr, self.nbytes)
            return True
        self._engine._device_wait(self)
        return False

    def _orphan(self) -> None:
        """the engine is closing: an import that is still alive frees the memory itself when its consumer lets go"""
        if self._ptr:
            ptr, hold, self._ptr, self._hold = self._ptr, self._hold, 0, None
            self._valid = False
            if _lib.load().pm_dlpack_release(ctypes.c_void_p(hold), 1) == 0:
                return  # (the last import's deleter runs hipFree)
            # (no import left: pm_dlpack_release freed the block)

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            if self._ptr and not sys.is_finalizing():
                self._valid = False
                if self._exports == 0:
                    self._settle()
                else:
                    self._orphan()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ interchange
    @property
    def __cuda_array_interface__(self) -> dict:
        self._check()
        stream = self._engine.stream
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self._ptr, True), 'version': 3, 'strides': None,
                'stream': stream if stream else 1}  # fmt: skip

    def __dlpack_device__(self) -> tuple:
        return (_KDLROCM, self._engine.device)

    def __dlpack__(self, stream=None, **_ignored):
        """
        A 'dltensor' capsule over the array (no copy). The engine's stream is synchronised first: whatever stream the
        consumer then works on sees finished data.
        """
        self._check()
        self._engine.synchronize()
        shape = (ctypes.c_int64 * max(len(self.shape), 1))(*self.shape)
        managed = _lib.load().pm_dlpack_export(ctypes.c_void_p(self._hold), _DL_CODES[self.dtype.kind], self.dtype.itemsize * 8,
                                              len(self.shape), ctypes.cast(shape, ctypes.c_void_p))
        if not managed:
            raise MemoryError('pm_dlpack_export failed')
        return _api.PyCapsule_New(managed, _NAME, _capsule_destructor)

    # ------------------------------------------------------------------ convenience
    def torch(self):
        """`torch.from_dlpack(self)`: a tensor on this GPU over the same memory (do not write to it)"""
        import torch

        return torch.from_dlpack(self)

    def numpy(self) -> np.ndarray:
        """a host copy (what the numpy form of the getter would have returned)"""
        self._check()
        out = np.empty(self.shape, dtype=self.dtype)
        self._engine.synchronize()
        self._engine.d2h(out, self._ptr)
        return out

    def __repr__(self) -> str:
        return f'DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self._engine.device}, {"valid" if self._valid else "invalidated"})'
