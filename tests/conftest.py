import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')
RELSTR = {0: None, 1: '<=', 2: '=='}


ROCSOLVER = '/opt/rocm/lib/librocsolver.so'
_warm = {'thread': None, 'bytes': 0}


def _warm_file(path):
    """Read a file once so that a later dlopen finds it in the page cache (no dl lock is held meanwhile)."""
    try:
        with open(path, 'rb', buffering=0) as f:
            while True:
                chunk = f.read(1 << 24)
                if not chunk:
                    break
                _warm['bytes'] += len(chunk)
    except OSError:
        pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # GPU sessions: librocsolver.so is 0.9 GB and a fresh box pages it in at a few MB/s; start reading it now, in the
    # background, so that the one test that needs it (device-side eigendecompositions) does not wait minutes for dlopen
    expr = config.getoption('-m', default='') or ''
    if 'gpu' in expr and 'not gpu' not in expr and os.path.exists(ROCSOLVER) and _warm['thread'] is None:
        import threading
        _warm['thread'] = threading.Thread(target=_warm_file, args=(ROCSOLVER,), daemon=True)
        _warm['thread'].start()


def pytest_collection_modifyitems(config, items):
    """Tests that start PROCESSES (bench.py under a launcher, two engines in two processes) go last: on a fresh GPU box every new
    process pages python, numpy and the HIP runtime in from the image, and while the background read of librocsolver.so (above)
    saturates the disk each of those start-ups takes 40 s instead of 1 s (measured: the two-process test 105 s inside a session,
    1.7 s for the same commands outside one).  By the end of the session the read has long finished."""
    late = ('test_bench_under_the_drivers_launcher_on_one_gpu', 'test_two_real_engines_two_processes_share_one_gpu')
    first = [it for it in items if not any(name in it.nodeid for name in late)]
    last = [it for it in items if any(name in it.nodeid for name in late)]
    items[:] = first + last


def rocsolver_warm(timeout):
    """Wait for the background read of librocsolver.so; True if it finished (or was never started)."""
    t = _warm['thread']
    if t is None:
        return True
    t.join(timeout)
    return not t.is_alive()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def funcs_from_npz(z):
    """Rebuild the raw-array problem [(P, q, r, relop)] (objective first) stored in a fixture."""
    return [(z['P'][k], z['q'][k], float(z['r'][k]), RELSTR[int(z['relop'][k])])
            for k in range(z['P'].shape[0])]


def oracle_map(fn, items):
    """Independent oracle runs side by side (the C oracle runs without the GIL; the GPU box offers 16 host cores)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=max(1, min(16, cores, len(items)))) as ex:
        return list(ex.map(fn, items))


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
