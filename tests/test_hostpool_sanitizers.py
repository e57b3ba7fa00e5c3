"""
The CPU side of the host <-> HBM leg under ThreadSanitizer and AddressSanitizer + UBSan, without a GPU.

`planetmapper_amd/csrc/pm_hostpool.h` - the pool of copy threads (a queue of jobs cut into parts), the ring of
pinned staging buffers, its retire thread with out-of-order slot release, the disc-span and block-gather jobs -
is HIP-free: the library plugs the HIP runtime in behind `pmh::CopyBackend` (pm_hostpipe.hip), the harness
`tests/hostpool/harness.cpp` a thread that plays the DMA engine. The harness replays the patterns of the GPU
soak (`tests/soak_hostpath.py`): 1 ... 16 threads, staging buffers of 0.25 ... 16 MiB, planes of every size,
disc planes into pageable and pinned destinations, gathers, backend failures in the middle of a call, pool
resizes between calls - and compares every destination byte. Here: a third of the (threads x staging) grid per
sanitizer with a seed of this run (logged; `PM_FUZZ_SEED` replays it); `PM_HOSTPOOL_FULL=1` runs the whole grid
(TSan: ~3 min).
"""

import os
import shutil
import subprocess

import pytest

from conftest import REPO, fresh_seed

HARNESS = os.path.join(REPO, 'tests', 'hostpool', 'harness.cpp')
INCLUDE = os.path.join(REPO, 'planetmapper_amd', 'csrc')


@pytest.mark.parametrize('name,flags', [
    ('tsan', ['-fsanitize=thread']),
    ('asan_ubsan', ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined']),
])  # fmt: skip
def test_hostpool_is_clean_under_the_sanitizers(tmp_path, name, flags):
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = str(tmp_path / f'hostpool_{name}')
    cc = subprocess.run(['g++', '-std=c++17', '-O1', '-g', f'-I{INCLUDE}', *flags, '-o', exe, HARNESS, '-lpthread'],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-3000:]
    seed = fresh_seed(f'test_hostpool_is_clean_under_the_sanitizers[{name}]')
    quick = '0' if os.environ.get('PM_HOSTPOOL_FULL') else '1'
    env = dict(os.environ, TSAN_OPTIONS='halt_on_error=1 second_deadlock_stack=1', ASAN_OPTIONS='detect_leaks=1',
               UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe, str(seed), quick], capture_output=True, text=True, timeout=1500, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert 'checks passed' in r.stdout
    assert 'ThreadSanitizer' not in out and 'AddressSanitizer' not in out and 'runtime error' not in out, out[-4000:]
