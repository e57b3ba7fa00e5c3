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


_ALL_SEEDS: dict = {}     # every seed drawn in this session, and those of tests that failed: the LAST lines of the run
_FAILED_SEEDS: dict = {}  # (pytest_terminal_summary) - a truncated tail of the log still names them


def pytest_terminal_summary(terminalreporter):
    if _ALL_SEEDS:
        terminalreporter.write_line('fuzz seeds of this run: ' + ', '.join(f'{n}={s}' for n, s in _ALL_SEEDS.items()))
    if _FAILED_SEEDS:
        terminalreporter.write_line('FAILED with a fresh seed - replay: ' + '; '.join(f'PM_FUZZ_SEED={s} python -m pytest tests -m gpu -k {n}' for n, s in _FAILED_SEEDS.items()))


_FRESH_SEEDS: dict = {}  # seeds drawn by the test that is running (pytest_runtest_call below puts them into a failure's message)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """
    A fresh-seed leg that fails must NAME its seed where every reader of the run sees it - the one-line summary
    (`FAILED ...::test[fresh_seed] - AssertionError: [replay: PM_FUZZ_SEED=...] ...`), not only captured stdout or a log
    file the driver does not collect: the seed is put in front of the exception's own message.
    """
    _FRESH_SEEDS.clear()
    outcome = yield
    if outcome.excinfo is not None and _FRESH_SEEDS:
        exc = outcome.excinfo[1]
        tag = '[' + ', '.join(f'PM_FUZZ_SEED={s}' for s in _FRESH_SEEDS.values()) + '] '
        _FAILED_SEEDS.update(_FRESH_SEEDS)
        first = exc.args[0] if exc.args else ''
        exc.args = (tag + (first if isinstance(first, str) else repr(first)),) + tuple(exc.args[1:])


def fresh_seed(test_name: str) -> int:
    """
    Seed of the per-run leg of a fuzz test: from PM_FUZZ_SEED if set (to replay a failure), else from the
    clock. Printed, appended to gpurun_out/fuzz_seeds.log, and - if the test fails - put in front of the failure's own
    message (pytest_runtest_call above).
    """
    import time

    env = os.environ.get('PM_FUZZ_SEED')
    seed = int(env) if env else int(time.time_ns() % (2**32))
    _FRESH_SEEDS[test_name] = seed
    _ALL_SEEDS[test_name] = seed
    line = f'{test_name}: PM_FUZZ_SEED={seed}'
    print('\n[fuzz] ' + line)
    out = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'fuzz_seeds.log'), 'a') as f:
            f.write(line + '\n')
    return seed
