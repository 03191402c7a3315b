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
    return np.load(os.path.join(GOLDEN, name))


def coo_to_dense_like(r, c, d, shape):
    from scipy import sparse
    return sparse.csr_matrix((d, (r, c)), shape=shape)


@pytest.fixture(scope='session')
def fb():
    """The HIP library handle; GPU tests fail loudly if it cannot be loaded."""
    import feabas_amd
    return feabas_amd
