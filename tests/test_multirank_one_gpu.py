"""
The multi-rank code on ONE MI355X (`-m gpu`): the boxes of this pool have one GPU and RCCL refuses two ranks
on one card ("Duplicate GPU detected", profiles/r04_multiproc_probe.jsonl), so

  * the C ABI's sharded cube (`pm_comm_*`, `pm_map_cube_sharded`, planetmapper_amd/csrc/pm_comm.hip) runs at
    world sizes 2, 3 and 8 over the loopback transport of tests/loopback, bound through the library's own
    PM_RCCL_LIBRARY override: its Send / Recv groups issued from the chunk callback, the closing all-reduce, a
    rank whose mapping fails, the redo round after a nanmedian replay, a failing transport;
  * the torch.distributed form (`distributed.map_cube_sharded_pipelined`) runs with 2 and 4 REAL engines on
    device 0 - one process each, pinned host blocks, `host_cube=True` - over a gloo group.

Each rank compares the gathered cube, bit for bit, with the whole cube mapped by itself alone
(tests/multirank_worker.py). The children are fresh processes that initialise the GPU themselves; this
process only starts them and reads their reports (at most 4 at a time: the pool allows 6 processes on a card).
Reference: the plane loop of Observation._get_mapped_data, observation.py:876-905.
"""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'multirank_worker.py')
LOOPBACK = os.path.join(HERE, 'loopback', 'libpm_loopback_nccl.so')


def _env(**extra):
    env = dict(os.environ, PM_RCCL_LIBRARY=LOOPBACK, PM_LOOPBACK_TIMEOUT_S='60', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('PM_LOOPBACK_HOST_ONLY', None)
    env.update(extra)
    return env


def _reports(procs, timeout=600):
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += '\n[test] killed after timeout'
        outs.append(o)
    reps = []
    for p, o in zip(procs, outs):
        lines = [ln for ln in o.splitlines() if ln.startswith('{')]
        assert p.returncode == 0 and lines, o[-3000:]
        reps.append(json.loads(lines[-1]))
    return reps


def _build_loopback():
    if not os.path.exists(LOOPBACK):
        subprocess.run(['make', '-C', os.path.join(HERE, 'loopback')], check=True)


@pytest.mark.parametrize('world', [2, 3, 8])
def test_c_abi_sharded_cube_over_the_loopback_transport_threads(world):
    """ranks = threads of one child process, each with its own pm_ctx and pm_comm"""
    _build_loopback()
    p = subprocess.Popen([sys.executable, WORKER, 'capi', '--threads', '--world', str(world)], env=_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    (rep,) = _reports([p])
    assert rep['ok'] and not rep['hung'], rep
    assert set(rep['checks']) == {str(r) for r in range(world)}
    for checks in rep['checks'].values():
        assert set(checks) == {'device', 'host', 'pageable', 'uneven', 'redo', 'fail', 'fail_then_ok'}, checks
    groups, sends, recvs, nbytes, allreduces, aborts = rep['loopback_stats[groups,sends,recvs,bytes,allreduces,aborts]']
    # every call closes with one status all-reduce per rank; the exchanges are groups of (world - 1) sends and receives
    assert allreduces >= 7 * world and sends == recvs and sends >= groups * (world - 1) > 0 and aborts == 0
    assert nbytes > 0


def test_c_abi_sharded_cube_at_config5_shapes_world_8():
    """
    BASELINE config 5 as the north star shards it, every number real but the transport: 512 planes of 1024^2 f64 (4 GiB),
    8 ranks (threads: the pool allows 6 PROCESSES on a card) x 64 planes, the 1 deg map, 33.2 MB per rank in the exchange
    sequence of pm_exchange_planes (8 exchanges per rank - 7 of 9 planes, one of 1 - each a group of 7 sends and 7
    receives), from pinned host blocks and from device blocks; every rank's gathered (512, 180, 360) cube bit-equal to the
    cube mapped by one rank alone.
    """
    _build_loopback()
    world = 8
    p = subprocess.Popen([sys.executable, WORKER, 'capi', '--threads', '--world', str(world), '--planes', '512', '--size', '1024',
                          '--cases', 'host,device', '--deadline', '900'], env=_env(PM_LOOPBACK_TIMEOUT_S='300'), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    (rep,) = _reports([p], timeout=1000)
    assert rep['ok'] and not rep['hung'], rep
    assert all(set(c) == {'host', 'device'} for c in rep['checks'].values()) and len(rep['checks']) == world
    groups, sends, recvs, nbytes, allreduces, aborts = rep['loopback_stats[groups,sends,recvs,bytes,allreduces,aborts]']
    assert aborts == 0 and sends == recvs == 2 * world * 8 * (world - 1)  # 2 cases x 8 ranks x 8 exchanges x 7 peers
    assert nbytes == 2 * world * (world - 1) * 64 * 180 * 360 * 8  # every rank's 64 mapped planes to each of its 7 peers, twice


def test_c_abi_sharded_cube_survives_a_failing_transport():
    """the third send of rank 1 fails inside the exchange: every rank returns an error, nobody waits for ever"""
    _build_loopback()
    world = 4
    p = subprocess.Popen([sys.executable, WORKER, 'capi', '--threads', '--world', str(world), '--cases', 'send_fault', '--deadline', '240'],
                         env=_env(PM_LOOPBACK_FAIL='1:3', PM_LOOPBACK_TIMEOUT_S='30'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    (rep,) = _reports([p])
    assert rep['ok'] and not rep['hung'], rep
    assert rep['loopback_stats[groups,sends,recvs,bytes,allreduces,aborts]'][5] >= 1


def test_a_failing_transport_with_an_abort_that_releases_nobody_but_the_caller():
    """
    The same failure with PM_LOOPBACK_LOCAL_ABORT=1: the failing rank's ncclCommAbort is local, as RCCL's is. Its peers are
    in receives from it; what ends their wait is the TRANSPORT's own failure detection (here PM_LOOPBACK_TIMEOUT_S; on real
    RCCL its watchdog / NCCL_TIMEOUT, if configured - pm_comm adds none of its own) - every rank then returns an error.
    The protocol guarantees that no rank returns success with an invalid cube, not that a dead link is noticed quickly.
    """
    _build_loopback()
    world = 4
    p = subprocess.Popen([sys.executable, WORKER, 'capi', '--threads', '--world', str(world), '--cases', 'send_fault', '--deadline', '240'],
                         env=_env(PM_LOOPBACK_FAIL='1:3', PM_LOOPBACK_TIMEOUT_S='4', PM_LOOPBACK_LOCAL_ABORT='1'), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    (rep,) = _reports([p])
    assert rep['ok'] and not rep['hung'], rep


@pytest.mark.parametrize('world', [2, 4])
def test_c_abi_sharded_cube_over_the_loopback_transport_processes(world, tmp_path):
    """ranks = processes sharing device 0 (the shape of a real launch: one process per rank)"""
    _build_loopback()
    uid = str(tmp_path / 'uid')
    procs = [subprocess.Popen([sys.executable, WORKER, 'capi', '--world', str(world), '--rank', str(r), '--uid-file', uid,
                               '--cases', 'device,host,uneven,redo,fail'], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    reps = _reports(procs)
    assert all(r['ok'] for r in reps), reps
    assert sorted(r['rank'] for r in reps) == list(range(world))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 4])
def test_torch_distributed_protocol_with_real_engines_sharing_one_gpu(world):
    """map_cube_sharded_pipelined(host_cube=True): real engines, pinned blocks, gloo group, device 0 for every rank"""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, WORKER, 'pyproto', '--cases', 'host,uneven,redo,fail'],
                              env=_env(RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    reps = _reports(procs)
    assert all(r['ok'] for r in reps), reps
    for r in reps:
        assert set(r['checks']) == {'host', 'uneven', 'redo', 'fail', 'fail_then_ok'}, r


def test_bench_headline_and_sharded_cube_at_two_ranks_on_one_gpu():
    """
    `bench.py --gpus 2 --shared-gpu`: the code the driver's N > 1 runs execute - launcher, process group, barriers,
    the headline line printed BEFORE the extra sections, the sharded host-fed cube (plain all-gather form, then the
    pipelined protocol with its exchanges issued from the engine's chunk callback, then every rank feeding at once
    without a collective) and the complete line - with two real engines on device 0 and a gloo group in place of
    RCCL (which refuses two ranks on one card). Sizes cut down; the numbers mean nothing here, the flow does.
    """
    repo = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--shared-gpu', '--size', '1024', '--steps', '10',
                        '--warmup', '2', '--preheat-steps', '5', '--planes', '48'], env=_env(), capture_output=True, text=True, timeout=900)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 0 and len(lines) == 2, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    first, last = lines
    assert first['n_gpus'] == 2 and 'pending' in first['extras'] and 'cube_host' not in first
    assert last['value'] == first['value'] and last['ms_per_step'] == first['ms_per_step'] and 'extras' not in last
    sec = last['cube_host']
    assert 'error' not in sec and 'pipelined_error' not in sec, sec
    assert sec['ranks'] == 2 and sec['planes_per_rank'] == 24 and sec['rccl_ranks'] == 0 and 'gloo' in sec['collective_backend']
    assert sec['fed_equals_resident'] is True and sec['fed_equals_resident_plain_allgather'] is True
    assert sec['ms_per_step_host_fed'] > 0 and sec['ms_per_step_host_fed_no_collective'] > 0


def test_bench_gives_up_on_a_hanging_extra_section_and_keeps_its_headline():
    """
    The optional sharded-cube section of an N > 1 run is given a deadline (PM_BENCH_EXTRAS_TIMEOUT_S): when it passes -
    here at once - rank 0 has printed the complete headline line with the section marked as timed out, and the run leaves
    with a status of its own (75: not success, not a crash); the measured headline of that N is never lost to a
    collective that hangs, and a caller that reads the status learns that a section was abandoned.
    """
    repo = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--shared-gpu', '--size', '1024', '--steps', '10',
                        '--warmup', '2', '--preheat-steps', '5', '--planes', '48'], env=_env(PM_BENCH_EXTRAS_TIMEOUT_S='0.05'),
                       capture_output=True, text=True, timeout=600)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 75 and len(lines) == 2, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert 'pending' in lines[0]['extras'] and lines[1]['value'] == lines[0]['value']
    assert 'timed out' in lines[1]['cube_host']['error'] and lines[1]['cube_host']['exit_status'] == 75
