import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

if os.environ.get('UNFLOW_LIB_PATH'):
    # a variant build of the library under test (python -m unopticalflow_amd.build --tuning with UNFLOW_TUNING_TAG; tools/gpu_r6.sh
    # loss_pending): test infrastructure only -- the product always loads libunflow_hip.so next to the package
    from unopticalflow_amd import _lib as _unflow_lib
    _unflow_lib.LIB_PATH = os.environ['UNFLOW_LIB_PATH']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'own_process: a -m gpu test that runs its body in a child pytest (tests/test_zz_round5_gpu.py)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]
    return load


if os.environ.get('UNFLOW_TESTS_ON_HOST') == '1':
    # the -m gpu tests rehearsed without a GPU (tests/rehearsal.py: what it is for and what it cannot answer); with UNFLOW_REHEARSAL_TIMES=file
    # one line per test: wall seconds and the seconds spent outside the host-executed kernels (oracle + torch)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rehearsal
    rehearsal.install()

    @pytest.fixture(autouse=True)
    def _over_host_executed_kernels():
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import hostexec
        from unopticalflow_amd import ops
        spent = [0.0]
        call = ops._call

        def timed_call(*a, **k):                                    # time inside the host-executed kernels: what a GPU would NOT spend
            t0 = time.perf_counter()
            try:
                return call(*a, **k)
            finally:
                spent[0] += time.perf_counter() - t0
        ops._call = timed_call
        t0 = time.perf_counter()
        try:
            with hostexec.patched(ops):
                yield
        finally:
            ops._call = call
            if os.environ.get('UNFLOW_REHEARSAL_TIMES'):             # one line per test: wall seconds, seconds outside the kernels (oracle + torch)
                wall = time.perf_counter() - t0
                with open(os.environ['UNFLOW_REHEARSAL_TIMES'], 'a') as f:
                    f.write('%s\t%.2f\t%.2f\n' % (os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0], wall, wall - spent[0]))
