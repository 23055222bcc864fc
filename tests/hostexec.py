"""ops.py on CPU tensors over the HOST-EXECUTED kernel library -- TEST INFRASTRUCTURE.

``with hostexec.patched(ops):`` points the binding at tests/host_check/_build/libunflow_hostexec.so (tests/host_check/build_hostexec.py: the
kernel source files of csrc/ compiled for the build host, lanes as fibers) and lets the autograd wrappers take CPU tensors: the product's
Python and the REAL kernel sources, end to end, without a GPU.  Outside the block everything is as it was -- the product has no CPU path
and never learns of this library (tests/test_abi.py::test_no_cpu_fallback still holds).  The host library holds every kernel file of csrc/
(the fp32 cost-volume kernels with their LDS-DMA rings, the fused warp + cost volume, bf16 epilogues and Adam included)."""
import contextlib
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_check'))
import build_hostexec  # noqa: E402

_cache = {}


def library():
    """The host-executed library with the binding's argument types (None without the ROCm clang++)."""
    if 'lib' not in _cache:
        path = build_hostexec.build()
        lib = None
        if path is not None:
            from unopticalflow_amd import _lib
            lib = ctypes.CDLL(path)
            for name, argtypes in _lib.SIGNATURES.items():
                if hasattr(lib, name):
                    fn = getattr(lib, name)
                    fn.argtypes = argtypes
                    fn.restype = ctypes.c_int
            assert lib.unflow_abi_version() == _lib.ABI_VERSION
        _cache['lib'] = lib
    return _cache['lib']


@contextlib.contextmanager
def patched(ops):
    from unopticalflow_amd import _lib
    lib = library()
    if lib is None:
        import pytest
        pytest.skip('the host-executed library needs the ROCm clang++ (vector extensions, __bf16)')
    old = (_lib._lib, ops._dev, ops._stream, ops._on, ops.on_device)
    capturing = torch.cuda.is_current_stream_capturing

    def dev(*tensors):
        for t in tensors:
            if t is not None and t.dtype != torch.float32:
                raise TypeError('unopticalflow_amd ops compute in fp32; got %s' % t.dtype)
        return torch.device('cpu')
    _lib._lib, ops._dev, ops._stream, ops._on, ops.on_device = lib, dev, (lambda: None), (lambda d: contextlib.nullcontext()), (lambda t: True)
    torch.cuda.is_current_stream_capturing = lambda: False          # (the trainer asks before it checks what a backward pass left; no device here)
    try:
        yield lib
    finally:
        _lib._lib, ops._dev, ops._stream, ops._on, ops.on_device = old
        torch.cuda.is_current_stream_capturing = capturing
