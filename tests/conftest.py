import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """Load tests/golden/<name>.npz into {key: array}; 'params/<k>' style keys become nested dicts."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    out = {}
    for k in z.files:
        if '/' in k:
            grp, sub = k.split('/', 1)
            out.setdefault(grp, {})[sub] = z[k]
        else:
            out[k] = z[k]
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden
