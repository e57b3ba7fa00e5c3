import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """
    The fresh-seed legs of the fuzz tests (a seed of the run, logged: `fresh_seed` below) go to the END of the session:
    they exist to find what the fixed seeds cannot, so one of them may fail on a box for a corner nobody has met
    before - under `pytest -x` that must not hide the tests that would otherwise have run after it.
    """
    late = [it for it in items if 'fresh_seed' in it.name]
    if late:
        items[:] = [it for it in items if 'fresh_seed' not in it.name] + late


@pytest.fixture(scope='session')
def jupiter():
    """Geometry block of Body('Jupiter', observer='HST', utc='2005-01-01T00:00:00')."""
    from planetmapper_amd.scenarios import load_scenario

    return load_scenario('jupiter_hst_2005')


@pytest.fixture(scope='session')
def saturn():
    from planetmapper_amd.scenarios import load_scenario

    return load_scenario('saturn_earth_2005')


@pytest.fixture(scope='session')
def jupiter_info():
    from planetmapper_amd.scenarios import scenario_info

    return scenario_info('jupiter_hst_2005')


def fresh_seed(test_name: str) -> int:
    """
    Seed of the per-run leg of a fuzz test: from PM_FUZZ_SEED if set (to replay a failure), else from the
    clock. Printed (pytest -s / the failure report shows it) and appended to gpurun_out/fuzz_seeds.log.
    """
    import time

    env = os.environ.get('PM_FUZZ_SEED')
    seed = int(env) if env else int(time.time_ns() % (2**32))
    line = f'{test_name}: PM_FUZZ_SEED={seed}'
    print('\n[fuzz] ' + line)
    out = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'fuzz_seeds.log'), 'a') as f:
            f.write(line + '\n')
    return seed
