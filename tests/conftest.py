import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')
RELSTR = {0: None, 1: '<=', 2: '=='}


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def funcs_from_npz(z):
    """Rebuild the raw-array problem [(P, q, r, relop)] (objective first) stored in a fixture."""
    return [(z['P'][k], z['q'][k], float(z['r'][k]), RELSTR[int(z['relop'][k])])
            for k in range(z['P'].shape[0])]


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
