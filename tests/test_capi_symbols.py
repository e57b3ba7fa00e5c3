"""
CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol declared in include/planetmapper_hip.h, struct layouts agree between the header
(as compiled into the oracle) and the ctypes mirrors, and the product fails loudly
when there is no GPU (no CPU fallback).
"""

import ctypes
import os
import re

import pytest

from conftest import REPO
from planetmapper_amd import _lib
from planetmapper_amd.geometry import PMDisc, PMGeometry


def _declared_functions():
    hdr = open(os.path.join(REPO, 'include', 'planetmapper_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(pm_[a-z0-9_]+)\s*\(', hdr)))


def test_header_declares_what_python_binds():
    assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared_functions():
        assert hasattr(lib, name), f'{name} missing from libplanetmapper_hip.so'
    assert lib.pm_abi_version() == 3


def test_struct_layouts_match_the_header():
    from oracle import oracle

    o = oracle.lib()
    assert o.pmo_sizeof_geometry() == ctypes.sizeof(PMGeometry)
    assert o.pmo_sizeof_disc() == ctypes.sizeof(PMDisc)


def test_no_gpu_means_loud_failure():
    """On a box without a gfx950 device the engine must refuse to work (no fallback)."""
    from planetmapper_amd.engine import Engine, device_count

    if device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.NoDeviceError):
        Engine(0)


def test_product_does_not_import_the_oracle():
    """Nothing under planetmapper_amd/ may reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(REPO, 'planetmapper_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(root, f), encoding='utf-8').read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert 'pm_oracle' not in text and 'libpm_oracle' not in text, f


def test_c_shard_bounds_equal_the_python_ones():
    """pm_shard_bounds (pure arithmetic: callable without a GPU) == distributed.shard_bounds"""
    from planetmapper_amd.distributed import shard_bounds

    lib = _lib.load()
    for n in (0, 1, 3, 10, 13, 512, 513):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                assert lib.pm_shard_bounds(n, world, rank, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
                assert (a.value, b.value, c.value) == shard_bounds(n, world, rank)
    assert lib.pm_shard_bounds(4, 2, 2, None, None, None) == _lib.PM_ERR_INVALID_ARGUMENT


def test_missing_rccl_is_an_error_code_not_a_crash():
    """
    A box without RCCL: pm_comm_unique_id must return PM_ERR_UNSUPPORTED (the path that used to call
    dlerror() twice and build a std::string from NULL). Forced with PM_RCCL_LIBRARY, in a child
    process because the binding is resolved once per process.
    """
    import subprocess
    import sys

    code = (
        'import ctypes, sys\n'
        f'sys.path.insert(0, {REPO!r})\n'
        'from planetmapper_amd import _lib\n'
        'lib = _lib.load()\n'
        'buf = ctypes.create_string_buffer(128)\n'
        'print("rc", lib.pm_comm_unique_id(buf))\n'
    )
    env = dict(os.environ, PM_RCCL_LIBRARY='/nonexistent/librccl-missing.so')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f'rc {_lib.PM_ERR_UNSUPPORTED}' in r.stdout


def test_hand_declared_rccl_abi_matches_the_installed_header(tmp_path):
    """
    pm_comm.hip binds RCCL at run time through hand declarations (planetmapper_amd/csrc/pm_rccl_abi.h: no build- or
    link-time dependency). tests/rccl_abi_check.cpp includes them NEXT TO the installed <rccl/rccl.h> and static_asserts
    sizeof / alignof(ncclUniqueId), the enum values and sizes, and every bound prototype: compiled here (the compiler, not
    a regular expression, reads the header) - and once more with a wrong prototype and a wrong constant, which must not compile.
    """
    import re
    import shutil
    import subprocess

    header = '/opt/rocm/include/rccl/rccl.h'
    if not os.path.exists(header) or not shutil.which('g++'):
        pytest.skip('no rccl.h / g++ on this machine')
    csrc = os.path.join(REPO, 'planetmapper_amd', 'csrc')
    cmd = ['g++', '-std=c++17', '-fsyntax-only', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include']
    check = os.path.join(REPO, 'tests', 'rccl_abi_check.cpp')
    ok = subprocess.run(cmd + ['-I' + csrc, check], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-2000:]
    ours = open(os.path.join(csrc, 'pm_rccl_abi.h')).read()
    for wrong in (ours.replace('using Send_t = int (*)(const void *, size_t,', 'using Send_t = int (*)(const void *, int,'),
                  ours.replace('Float64 = 8', 'Float64 = 7'), ours.replace('char internal[128]', 'char internal[64]')):
        assert wrong != ours
        (tmp_path / 'pm_rccl_abi.h').write_text(wrong)
        bad = subprocess.run(cmd + ['-I' + str(tmp_path), check], capture_output=True, text=True)
        assert bad.returncode != 0 and 'static assertion failed' in bad.stderr
    src = open(os.path.join(csrc, 'pm_comm.hip')).read()
    assert '#include "pm_rccl_abi.h"' in src and 'PM_RCCL_SYMBOLS(PM_RCCL_BIND)' in src
    # (nothing in the library declares an nccl type or prototype of its own any more)
    assert not re.search(r'struct\s+ncclUniqueId\s*\{|\(int \(\*\)\([^)]*\)\)sym\("nccl', src)
    symbols = re.findall(r'X\(\w+, "(nccl\w+)"\)', ours)
    assert len(symbols) == 11
    # ... and the loopback transport of the tests exports exactly those symbols
    lb = open(os.path.join(REPO, 'tests', 'loopback', 'loopback_nccl.cpp')).read()
    for sym in symbols:
        assert re.search(r'\b%s\s*\(' % sym, lb), sym


def test_dlpack_export_layout_and_hold_bookkeeping_without_a_gpu():
    """
    pm_dlpack_*: the DLManagedTensor a `DeviceArray` hands to torch / cupy is built and released by C code of the library
    (no Python in a consumer's release). Its layout - dlpack.h's legacy struct, device type kDLROCM - and the hold's counting
    are host-side code: checked here on a made-up device address (nothing dereferences it; `free_memory` is never asked for).
    """
    import ctypes

    from planetmapper_amd import _lib

    lib = _lib.load()

    class DLDevice(ctypes.Structure):
        _fields_ = [('device_type', ctypes.c_int32), ('device_id', ctypes.c_int32)]

    class DLDataType(ctypes.Structure):
        _fields_ = [('code', ctypes.c_uint8), ('bits', ctypes.c_uint8), ('lanes', ctypes.c_uint16)]

    class DLTensor(ctypes.Structure):
        _fields_ = [('data', ctypes.c_void_p), ('device', DLDevice), ('ndim', ctypes.c_int32), ('dtype', DLDataType),
                    ('shape', ctypes.POINTER(ctypes.c_int64)), ('strides', ctypes.POINTER(ctypes.c_int64)), ('byte_offset', ctypes.c_uint64)]  # fmt: skip

    class DLManagedTensor(ctypes.Structure):
        _fields_ = [('dl_tensor', DLTensor), ('manager_ctx', ctypes.c_void_p), ('deleter', ctypes.c_void_p)]

    hold = lib.pm_dlpack_hold_create(ctypes.c_void_p(0x7F0000001000), 3)
    assert hold and lib.pm_dlpack_exports(ctypes.c_void_p(hold)) == 0
    shape = (ctypes.c_int64 * 3)(5, 180, 360)
    exports = [lib.pm_dlpack_export(ctypes.c_void_p(hold), 2, 64, 3, ctypes.cast(shape, ctypes.c_void_p)) for _ in range(2)]
    assert all(exports) and exports[0] != exports[1] and lib.pm_dlpack_exports(ctypes.c_void_p(hold)) == 2
    m = DLManagedTensor.from_address(exports[0])
    t = m.dl_tensor
    assert t.data == 0x7F0000001000 and (t.device.device_type, t.device.device_id) == (10, 3) and t.ndim == 3
    assert (t.dtype.code, t.dtype.bits, t.dtype.lanes) == (2, 64, 1) and [t.shape[i] for i in range(3)] == [5, 180, 360]
    assert not t.strides and t.byte_offset == 0 and m.manager_ctx == hold and m.deleter
    shape[0] = 99  # the export carries its own copy of the shape
    assert t.shape[0] == 5
    assert lib.pm_dlpack_export(ctypes.c_void_p(hold), 2, 64, 9, ctypes.cast(shape, ctypes.c_void_p)) is None  # ndim > 8
    # the owner lets go while two consumers still import: nothing is freed under them, the hold lives until the last deleter
    assert lib.pm_dlpack_release(ctypes.c_void_p(hold), 0) == 0
    assert lib.pm_dlpack_export(ctypes.c_void_p(hold), 2, 64, 3, ctypes.cast(shape, ctypes.c_void_p)) is None  # no new exports of an orphan
    lib.pm_dlpack_delete(ctypes.c_void_p(exports[0]))
    assert lib.pm_dlpack_exports(ctypes.c_void_p(hold)) == 1
    lib.pm_dlpack_delete(ctypes.c_void_p(exports[1]))  # (the last one: deletes the hold too)
    # an owner with no consumer left: released at once
    hold2 = lib.pm_dlpack_hold_create(ctypes.c_void_p(0x7F0000002000), 0)
    e = lib.pm_dlpack_export(ctypes.c_void_p(hold2), 1, 8, 1, ctypes.cast((ctypes.c_int64 * 1)(7), ctypes.c_void_p))
    lib.pm_dlpack_delete(ctypes.c_void_p(e))
    assert lib.pm_dlpack_exports(ctypes.c_void_p(hold2)) == 0 and lib.pm_dlpack_release(ctypes.c_void_p(hold2), 0) == 1
    assert lib.pm_dlpack_release(None, 0) == 1 and lib.pm_dlpack_exports(None) == 0
    lib.pm_dlpack_delete(None)
