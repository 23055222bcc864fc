"""A REHEARSAL of code written for the GPU, without one -- TEST INFRASTRUCTURE, off unless asked for.

``install()`` makes ``.cuda()`` / ``.to('cuda')`` / ``device='cuda'`` keep tensors in host memory (as copies, differentiable like the real
host-to-device copy) and answers the few ``torch.cuda`` questions GPU code asks; together with ``tests/hostexec.py`` (ops.py over the kernel
sources executed on the build host) the GPU tests' OWN code, ``__graft_entry__.smoke()`` and the product's train step run on CPU tensors:
    UNFLOW_TESTS_ON_HOST=1 python -m pytest tests/test_zz_round5_gpu.py -m gpu -k "..."          (tests/conftest.py calls install())
It finds mistakes in TEST code (shapes, expectations, argument order) before a GPU session pays for them and re-checks the kernels'
arithmetic; what only a device has -- MIOpen's convolutions (torch's CPU convolutions stand in), the kernel timer's events, hipGraphs, a
second process's GPU -- it cannot answer, and tests that need those fail here by design.  Nothing in the default runs (`-m "not gpu"`,
`-m gpu`) sees any of this."""
import torch

_installed = []


def install():
    if _installed:
        return
    _installed.append(True)
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
    _mto = torch.nn.Module.to

    def _module_to(self, *a, **k):                                     # model.to('cuda:0') / .to(device=...): stays where it is; dtypes etc. still apply
        a = tuple(x for x in a if not (isinstance(x, (str, torch.device)) and str(x).startswith('cuda')))
        if str(k.get('device', '')).startswith('cuda'):
            k.pop('device')
        return _mto(self, *a, **k) if (a or k) else self
    torch.nn.Module.to = _module_to
