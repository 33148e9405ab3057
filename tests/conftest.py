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
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


class _LibOptions(object):
    """Library switches (include/cpg_hip.h: cpg_set_option) for the duration of one test; the library reads the environment only
    once, when it is loaded, so tests flip switches through the C ABI."""

    def __init__(self):
        self._old = {}

    def set(self, name, value):
        from cpg_amd import _lib
        if name not in self._old:
            self._old[name] = _lib.get_option(name)
        _lib.set_option(name, value)

    def restore(self):
        from cpg_amd import _lib
        for name, v in self._old.items():
            _lib.set_option(name, v)


@pytest.fixture
def libopt():
    o = _LibOptions()
    yield o
    o.restore()
