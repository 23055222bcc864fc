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
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.empty_cache = lambda: None
    torch.cuda.get_device_name = lambda *a, **k: 'host rehearsal'
    _Generator = torch.Generator

    class _HostGenerator(_Generator):                                  # torch.Generator(device='cuda') -> the host generator (still a type: annotations use it)
        def __new__(cls, *a, **k):
            return _Generator.__new__(cls)
    torch.Generator = _HostGenerator
    import types
    torch.cuda.get_device_properties = lambda *a, **k: types.SimpleNamespace(gcnArchName='host:rehearsal', multi_processor_count=0, name='host rehearsal', total_memory=0)

    class _Event:                                                       # everything is in order on the host: events and streams have nothing to do
        def __init__(self, *a, **k): pass
        def record(self, *a, **k): pass
        def synchronize(self): pass
        def wait(self, *a, **k): pass
        def query(self): return True
        def elapsed_time(self, other): return 1.0                       # (a made-up millisecond: rates computed from it must not divide by zero)

    class _Stream:
        cuda_stream = 0
        def __init__(self, *a, **k): pass
        def wait_stream(self, *a, **k): pass
        def wait_event(self, *a, **k): pass
        def record_event(self, e=None): return e or _Event()
        def synchronize(self): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    import contextlib
    torch.cuda.Event, torch.cuda.Stream = _Event, _Stream
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.default_stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()

    class _Device:                                                      # (a class: torch.serialization asks isinstance(x, torch.cuda.device))
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    torch.cuda.device = _Device
    _load = torch.load

    def _load_on_host(f, *a, **k):
        if a and str(a[0]).startswith('cuda'):
            a = ('cpu',) + tuple(a[1:])
        if str(k.get('map_location', '')).startswith('cuda'):
            k['map_location'] = 'cpu'
        return _load(f, *a, **k)
    torch.load = _load_on_host
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.Tensor.record_stream = lambda self, *a, **k: None
    # hipGraphs cannot be rehearsed (a capture records device work): a trainer asked to replay its step steps eagerly here
    from unopticalflow_amd import trainer as _trainer
    _init = _trainer.FlowTrainer.__init__

    def _eager_init(self, *a, **k):
        k['use_graph'] = False
        _init(self, *a, **k)
    _trainer.FlowTrainer.__init__ = _eager_init
    torch.cuda.memory_allocated = lambda *a, **k: 0
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.cuda.reset_peak_memory_stats = lambda *a, **k: None
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
