"""The oracle's cost volume and its autograd for one seeded case, evaluated ONCE per test session -- TEST INFRASTRUCTURE.

`corr_naive` at d = 8 is 289 shifted products with an autograd graph behind them (several GB and tens of seconds on the host for a
[16,32,64,208] pair); the same case is wanted by several `-m gpu` tests (the fp32 kernels, the matrix-core backward, its pixel-pair form in a
child process).  ``corr_case`` keeps what the oracle returned in memory and in a file under the system's temporary directory (keyed by the
case AND by the bytes of oracle/ref_cpu.py, so an edited oracle never meets an old answer); inputs are regenerated from their seeds."""
import hashlib
import os
import tempfile

import numpy as np
import torch

from oracle import ref_cpu as R

_mem = {}
_MEM_CAP = 3 << 30          # bytes kept in memory (oldest entries go first); the files stay


def rnd(seed, shape, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * np.float32(scale))


def _oracle_tag():
    if 'tag' not in _mem:
        with open(R.__file__, 'rb') as f:
            _mem['tag'] = hashlib.sha1(f.read()).hexdigest()[:12]
    return _mem['tag']


def _dir():
    d = os.path.join(tempfile.gettempdir(), 'unflow_oracle_cache_%d' % os.getuid())
    os.makedirs(d, exist_ok=True)
    return d


def corr_case(d, B, C, h, w, seeds=(17, 18, 19), gscale=1.0):
    """-> dict(f1, f2, gout, cv, gf1, gf2): seeded inputs, `corr_naive(f1, f2, d)` and the gradients of `cv.backward(gout)` (pwc_tf.py:97-106
    through the oracle)."""
    key = 'corr_d%d_%dx%dx%dx%d_s%d_%d_%d_g%g_%s' % (d, B, C, h, w, seeds[0], seeds[1], seeds[2], gscale, _oracle_tag())
    DD = 2 * d + 1
    f1, f2 = rnd(seeds[0], (B, C, h, w)), rnd(seeds[1], (B, C, h, w))
    gout = rnd(seeds[2], (B, DD * DD, h, w), gscale)
    if key not in _mem:
        path = os.path.join(_dir(), key + '.npz')
        got = None
        if os.path.exists(path):
            try:
                z = np.load(path)
                got = tuple(torch.from_numpy(z[k]) for k in ('cv', 'gf1', 'gf2'))
            except Exception:                                       # a file cut short by a killed run: evaluate again
                got = None
        if got is None:
            a, b = f1.clone().requires_grad_(), f2.clone().requires_grad_()
            cv = R.corr_naive(a, b, d)
            cv.backward(gout)
            got = (cv.detach(), a.grad, b.grad)
            tmp = '%s.%d.tmp.npz' % (path, os.getpid())
            np.savez(tmp, cv=got[0].numpy(), gf1=got[1].numpy(), gf2=got[2].numpy())
            os.replace(tmp, path)
        _mem[key] = got
        held = [(k, sum(t.numel() * 4 for t in v)) for k, v in _mem.items() if k != 'tag']
        while sum(n for _, n in held) > _MEM_CAP and len(held) > 1:
            _mem.pop(held.pop(0)[0])
    cv, gf1, gf2 = _mem[key]
    return dict(f1=f1, f2=f2, gout=gout, cv=cv, gf1=gf1, gf2=gf2)
