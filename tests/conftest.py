import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_memory_between_tests():
    """Engines, solvers and their device buffers are freed by refcount, but a few hold reference cycles (ctypes callbacks, closures):
    collect after every test so that the GPU suite — one process, 260 tests, several of them tens of GiB — never carries the
    previous tests' allocations into the next one."""
    yield
    import gc
    gc.collect()
