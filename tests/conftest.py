import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

if os.environ.get('UNFLOW_LIB_PATH'):
    # a variant build of the library under test (python -m unopticalflow_amd.build --tuning with UNFLOW_TUNING_TAG; tools/gpu_r5.sh
    # loss_pending): test infrastructure only -- the product always loads libunflow_hip.so next to the package
    from unopticalflow_amd import _lib as _unflow_lib
    _unflow_lib.LIB_PATH = os.environ['UNFLOW_LIB_PATH']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]
    return load
