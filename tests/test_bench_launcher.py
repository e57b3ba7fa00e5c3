"""
`python bench.py --gpus N` typed directly must launch itself (one process per GPU through
torch.distributed.run, rendezvous on 127.0.0.1), print ONE JSON line from rank 0 and pass a
child's failure on as its exit status. Rehearsed here on CPU with the gloo backend and an empty
step (`--rehearse`: launcher, process group, barriers, max-over-ranks timing, all-gather, report).
"""

import json
import os
import subprocess
import sys

from conftest import REPO


def _run(extra, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--rehearse', '--steps', '3', '--warmup', '1'] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)


def _json_lines(stdout):
    return [json.loads(line) for line in stdout.splitlines() if line.startswith('{')]


def test_self_launch_two_ranks():
    r = _run(['--gpus', '2'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 only
    line = lines[0]
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1 and line['rehearsal'] is True
    for key in ('metric', 'value', 'unit', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config'):  # fmt: skip
        assert key in line


def test_single_process_needs_no_launcher():
    r = _run(['--gpus', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 1


def test_child_failure_is_the_exit_status():
    r = _run(['--gpus', '2'], {'PM_BENCH_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_runs_under_an_external_torchrun_too():
    """the driver's form: python -m torch.distributed.run ... bench.py --gpus N"""
    env = dict(os.environ)
    cmd = [
        sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
        '--master-port', '29717', os.path.join(REPO, 'bench.py'), '--gpus', '2', '--rehearse', '--steps', '2', '--warmup', '1',
    ]  # fmt: skip
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 2


def test_pmc_figures_are_reported_only_for_the_build_they_were_measured_on(tmp_path, monkeypatch):
    """
    `roofline.traffic` / `fp64` of the bench line come from profiles/traffic.json (PMC counters cannot be read
    inside the timed process). The file is stamped with the sha256 of the library it was measured on
    (tools/pmc_summary.py); with another build loaded the figures are withheld and the line says
    `traffic_stale: true`.
    """
    import json

    import bench

    (tmp_path / 'profiles').mkdir()
    rec = {'pm::k_disc_sph<1, false>': {'hbm_bytes': 671213229.0, 'fp64_flop': 4.5e9}}
    monkeypatch.setattr(bench, 'REPO', str(tmp_path))
    (tmp_path / 'profiles' / 'traffic.json').write_text(json.dumps(dict(rec, _library_sha256=bench.library_sha256())))
    got, stale = bench.profile_record('pm::k_disc_sph<1,')
    assert not stale and got['hbm_bytes'] == 671213229.0
    (tmp_path / 'profiles' / 'traffic.json').write_text(json.dumps(dict(rec, _library_sha256='0' * 64)))
    assert bench.profile_record('pm::k_disc_sph<1,') == ({}, True)
    (tmp_path / 'profiles' / 'traffic.json').write_text(json.dumps(rec))  # a file from before the stamp existed
    assert bench.profile_record('pm::k_disc_sph<1,') == ({}, True)
    (tmp_path / 'profiles' / 'traffic.json').unlink()
    assert bench.profile_record('pm::k_disc_sph<1,') == ({}, False)
