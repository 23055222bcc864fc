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
    # a variant build of the library under test (python -m unopticalflow_amd.build --tuning with UNFLOW_TUNING_TAG; tools/gpu_r5.sh
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
    # A REHEARSAL of the -m gpu tests without a GPU (test infrastructure, off unless asked for):
    #     UNFLOW_TESTS_ON_HOST=1 python -m pytest tests/test_zz_round5_gpu.py -m gpu -k "..."
    # runs the GPU tests' own code on CPU tensors over the host-executed build of the kernel sources (tests/hostexec.py) -- `.cuda()` and
    # `.to('cuda')` keep the tensor where it is, every test body sits inside hostexec.patched(ops).  It finds mistakes in TEST code (shapes,
    # expectations, argument order) before a GPU session pays for them, and re-checks the kernels' arithmetic; what only a device has --
    # MIOpen's convolutions (torch's CPU convolutions stand in), the kernel timer's events, hipGraphs, a second process's GPU -- it
    # cannot answer, and tests that need those fail here by design.  Nothing in the default runs (`-m "not gpu"`, `-m gpu`) sees any of this.
    import torch

    torch.Tensor.cuda = lambda self, *a, **k: self.clone()          # (a copy, differentiable like the real host-to-device copy: no aliasing)
    _to = torch.Tensor.to

    def _to_host(self, *a, **k):
        n = len(a) + len(k)
        a = tuple(x for x in a if not (isinstance(x, (str, torch.device)) and str(x).startswith('cuda')))
        if str(k.get('device', '')).startswith('cuda'):
            k.pop('device')
        moved = len(a) + len(k) < n
        out = _to(self, *a, **k) if (a or k) else self
        return out.clone() if (moved and out is self) else out
    torch.Tensor.to = _to_host
    for _name in ('zeros', 'ones', 'empty', 'full', 'rand', 'randn', 'randint', 'arange', 'tensor', 'as_tensor', 'zeros_like', 'empty_like', 'ones_like'):
        def _on_host(*a, _f=getattr(torch, _name), **k):              # torch.zeros(..., device='cuda') -> the same tensor in host memory
            if str(k.get('device', '')).startswith('cuda'):
                k.pop('device')
            return _f(*a, **k)
        setattr(torch, _name, _on_host)
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.is_current_stream_capturing = lambda: False
    torch.nn.Module.cuda = lambda self, *a, **k: self

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
