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


def pytest_sessionstart(session):
    """every test run is under the resident-set watchdog (feabas_amd/_watchdog.py): a runaway host allocation -- round 5's
    suspected region raster of a mesh thrown far away by a diverged solve -- ends THIS process with exit code 3 and a stack
    dump instead of taking the box down.  A thread (stack dump) at the limit, a child process (SIGKILL; works while a C call
    holds the GIL) at 1.15 x the limit.  FEABAS_RSS_LIMIT_GB (default 24; 0 = off)."""
    from feabas_amd import _watchdog
    if _watchdog.start() > 0:
        session.config._fb_backstop = _watchdog.start_backstop()


def pytest_sessionfinish(session, exitstatus):
    p = getattr(session.config, '_fb_backstop', None)
    if p is not None:
        p.kill()
        p.wait()


def pytest_collection_modifyitems(config, items):
    """every GPU test runs under a timeout of its own (pytest-timeout, 300 s unless the test sets one): a test that hangs
    -- a kernel that never finishes, a host loop fed by a runaway mesh -- fails there, with a stack dump, instead of holding the
    box until the caller's limit (a box held to that limit counts against the pool)"""
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for it in items:
        if it.get_closest_marker('gpu') is not None and it.get_closest_marker('timeout') is None:
            it.add_marker(pytest.mark.timeout(300, method='thread'))      # 'thread': ends the process even when the main thread sits in a HIP call


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
