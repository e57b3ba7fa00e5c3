"""
The loopback transport of tests/loopback/ (test infrastructure behind PM_RCCL_LIBRARY) on the CPU, in its
host-only mode: grouped send / recv between 2 ... 8 ranks as threads and as processes, all-gather, all-reduce,
abort releasing blocked peers, the no-progress timeout and the injected send failure. The GPU tests
(tests/test_multirank_one_gpu.py) run pm_map_cube_sharded over the same library.
"""

import ctypes
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'loopback', 'libpm_loopback_nccl.so')
F64, I32 = 8, 2


def _load():
    if not os.path.exists(LIB):
        subprocess.run(['make', '-C', os.path.join(HERE, 'loopback')], check=True)
    os.environ['PM_LOOPBACK_HOST_ONLY'] = '1'
    lib = ctypes.CDLL(LIB)
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    return lib


class UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _rank_body(lib, uid, world, rank, n, results, abort_rank=None):
    comm = ctypes.c_void_p()
    rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    assert rc == 0, lib.ncclGetErrorString(rc)
    mine = np.full(n, float(rank + 1))
    got = np.zeros((world, n))
    got[rank] = mine
    if abort_rank is not None:
        if rank == abort_rank:
            lib.ncclCommAbort(comm)
            results[rank] = 'aborted'
            return
        # the others block in a group the aborting rank never joins: the flag must release them
        lib.ncclGroupStart()
        for p in range(world):
            if p != rank:
                lib.ncclSend(_ptr(mine), ctypes.c_size_t(n), F64, p, comm, None)
                lib.ncclRecv(_ptr(got[p]), ctypes.c_size_t(n), F64, p, comm, None)
        results[rank] = lib.ncclGroupEnd()
        lib.ncclCommAbort(comm)
        return
    for _ in range(3):  # several groups in a row over the same rings
        lib.ncclGroupStart()
        for p in range(world):
            if p != rank:
                assert lib.ncclSend(_ptr(mine), ctypes.c_size_t(n), F64, p, comm, None) == 0
                assert lib.ncclRecv(_ptr(got[p]), ctypes.c_size_t(n), F64, p, comm, None) == 0
        assert lib.ncclGroupEnd() == 0
    assert all(np.all(got[p] == p + 1) for p in range(world))
    status = np.array([rank, 1], dtype=np.int32)
    total = np.zeros(2, dtype=np.int32)
    assert lib.ncclAllReduce(_ptr(status), _ptr(total), ctypes.c_size_t(2), I32, 0, comm, None) == 0
    assert total.tolist() == [world * (world - 1) // 2, world]
    ag = np.zeros((world, 5))
    assert lib.ncclAllGather(_ptr(np.full(5, rank + 0.5)), _ptr(ag), ctypes.c_size_t(5), F64, comm, None) == 0
    assert np.array_equal(ag[:, 0], np.arange(world) + 0.5)
    assert lib.ncclCommDestroy(comm) == 0
    results[rank] = 'ok'


def _run_threads(world, n, **kw):
    lib = _load()
    uid = UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    results = [None] * world
    errors = []

    def body(r):
        try:
            _rank_body(lib, uid, world, r, n, results, **kw)
        except BaseException as e:  # noqa: BLE001
            errors.append((r, e))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts), 'a rank is still blocked'
    assert not errors, errors
    return results


@pytest.mark.parametrize('world,n', [(2, 7), (3, 100_000), (8, 70_000)])
def test_groups_allreduce_allgather_between_threads(world, n):
    # (n * 8 bytes beyond the 256 KiB ring: both directions of a pair must progress in turn)
    assert _run_threads(world, n) == ['ok'] * world


def test_abort_releases_ranks_blocked_in_a_group():
    res = _run_threads(3, 1000, abort_rank=1)
    assert res[1] == 'aborted' and res[0] != 0 and res[2] != 0


_CHILD = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, {here!r})
import test_loopback_transport as t
lib = t._load()
uid = t.UniqueId()
uid.internal = bytes.fromhex(sys.argv[3])
res = [None] * int(sys.argv[1])
t._rank_body(lib, uid, int(sys.argv[1]), int(sys.argv[2]), 50_000, res)
assert res[int(sys.argv[2])] == 'ok'
'''


def test_ranks_as_processes():
    lib = _load()
    uid = UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    world = 4
    env = dict(os.environ, PM_LOOPBACK_HOST_ONLY='1')
    procs = [subprocess.Popen([sys.executable, '-c', _CHILD.format(here=HERE), str(world), str(r), bytes(uid.internal).hex()],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_timeout_and_injected_send_failure():
    code = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %r)
import test_loopback_transport as t
lib = t._load()
uid = t.UniqueId(); lib.ncclGetUniqueId(ctypes.byref(uid))
comm = ctypes.c_void_p()
assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0   # world 1: nobody to talk to
a = np.zeros(4)
assert lib.ncclSend(t._ptr(a), ctypes.c_size_t(4), t.F64, 0, comm, None) != 0   # peer == self: invalid
lib.ncclCommDestroy(comm)
# a second rank that never comes: init times out
uid2 = t.UniqueId(); lib.ncclGetUniqueId(ctypes.byref(uid2))
rc = lib.ncclCommInitRank(ctypes.byref(comm), 2, uid2, 0)
assert rc != 0 and b'timed out' in lib.ncclGetErrorString(rc), rc
print('ok')
''' % HERE
    env = dict(os.environ, PM_LOOPBACK_HOST_ONLY='1', PM_LOOPBACK_TIMEOUT_S='1')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stdout + p.stderr
    # the nth send of a rank fails: the group reports it
    code2 = r'''
import ctypes, os, sys, threading
import numpy as np
sys.path.insert(0, %r)
import test_loopback_transport as t
lib = t._load()
uid = t.UniqueId(); lib.ncclGetUniqueId(ctypes.byref(uid))
rcs = [None, None]
def body(r):
    comm = ctypes.c_void_p()
    assert lib.ncclCommInitRank(ctypes.byref(comm), 2, uid, r) == 0
    a, b = np.ones(10), np.zeros(10)
    lib.ncclGroupStart()
    lib.ncclSend(t._ptr(a), ctypes.c_size_t(10), t.F64, 1 - r, comm, None)
    lib.ncclRecv(t._ptr(b), ctypes.c_size_t(10), t.F64, 1 - r, comm, None)
    rcs[r] = lib.ncclGroupEnd()
    lib.ncclCommAbort(comm)
ts = [threading.Thread(target=body, args=(r,)) for r in range(2)]
[x.start() for x in ts]; [x.join(60) for x in ts]
assert rcs[0] == 3 and rcs[1] not in (0, None), rcs   # rank 0: injected; rank 1: released by the abort (or its timeout)
print('ok')
''' % HERE
    env = dict(os.environ, PM_LOOPBACK_HOST_ONLY='1', PM_LOOPBACK_TIMEOUT_S='5', PM_LOOPBACK_FAIL='0:1')
    p = subprocess.run([sys.executable, '-c', code2], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stdout + p.stderr
