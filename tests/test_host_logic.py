"""CPU tests of the host-side logic around the HIP operators (no kernel runs here): the bf16 option's weight shadows, the argument
checks the operators make before they touch a device, the torch forms the model falls back to off the GPU."""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from unopticalflow_amd import ops
from unopticalflow_amd.core.networks.structures import PWC_tf
from unopticalflow_amd.core.networks.structures.net_utils import HeadConv2d, WeightShadows, conv, conv_weight, weights_to_channels_last


def test_weight_shadows_cast_once_and_route_gradients():
    """net_utils.WeightShadows: inside the context every convolution contracts with a bf16 copy of its weight (heads also of their
    bias) made by ONE autograd node; outside nothing is left on the modules; the fp32 parameters get fp32 gradients in their own
    (channels_last) layout, equal to what per-call casts (autocast's arithmetic) give."""
    torch.manual_seed(0)
    net = nn.Sequential(conv(3, 8), conv(8, 8), HeadConv2d(8, 2, 3, padding=1))
    weights_to_channels_last(net)
    x = torch.randn(2, 3, 9, 7).bfloat16()

    def run(shadows):
        for p in net.parameters():
            p.grad = None
        if shadows:
            with WeightShadows((net,)):
                assert all(conv_weight(m).dtype == torch.bfloat16 for m in net.modules() if isinstance(m, nn.Conv2d))
                nodes = {conv_weight(m).grad_fn for m in net.modules() if isinstance(m, nn.Conv2d)}
                assert len(nodes) == 1                                  # one cast node for all weights
                t = x
                for blk in net[:2]:
                    c = blk[0]
                    t = F.leaky_relu(F.conv2d(t, conv_weight(c), c.bias.bfloat16(), padding=1), 0.1)
                y = net[2](t)
        else:
            t = x
            for blk in net[:2]:
                c = blk[0]
                t = F.leaky_relu(F.conv2d(t, c.weight.bfloat16(), c.bias.bfloat16(), padding=1), 0.1)
            y = F.conv2d(t, net[2].weight.bfloat16(), net[2].bias.bfloat16(), padding=1)
        y.float().square().sum().backward()
        return y.detach().float(), [p.grad.clone() for p in net.parameters()]

    ya, ga = run(True)
    assert not any('_w_half' in m.__dict__ or '_b_half' in m.__dict__ for m in net.modules())
    yb, gb = run(False)
    assert torch.equal(ya, yb)
    for p, a, b in zip(net.parameters(), ga, gb):
        assert a.dtype == torch.float32 and a.stride() == p.stride()
        assert torch.equal(a, b)
    with WeightShadows((net,), enabled=False):
        assert conv_weight(net[0][0]) is net[0][0].weight
    for p in net.parameters():
        p.grad = None
    with WeightShadows((net,)):                     # a shadow nobody reads (the head's bias when ops.flow_head adds it in fp32): no gradient, no error
        h = net[2]
        F.conv2d(x[:, :1].repeat(1, 8, 1, 1), conv_weight(h), None, padding=1).float().sum().backward()
    assert h.weight.grad is not None and h.bias.grad is None and net[0][0].weight.grad is None


def test_flow_heads_keep_the_reference_checkpoint_keys():
    """HeadConv2d replaces nn.Conv2d for predict_flow{6..2} / dc_conv7 (pwc_tf.py:93-94): same parameters, same state-dict keys, and
    without shadows the same forward."""
    pw = PWC_tf()
    keys = set(pw.state_dict().keys())
    for lvl in (6, 5, 4, 3, 2):
        assert {'predict_flow%d.weight' % lvl, 'predict_flow%d.bias' % lvl} <= keys
    assert {'dc_conv7.weight', 'dc_conv7.bias'} <= keys
    m = pw.predict_flow6
    x = torch.randn(1, m.in_channels, 4, 5)
    assert torch.equal(m(x), F.conv2d(x, m.weight, m.bias, padding=1))
    up = torch.randn(1, 2, 4, 5)
    assert torch.equal(pw._head(m, x, up), F.conv2d(x, m.weight, m.bias, padding=1) + up)      # the torch form off the GPU


def test_flow_upsampling_torch_form_is_the_reference_expression():
    """PWC_tf._up on tensors the HIP kernel does not take (here: host tensors of the module-level tests) is literally the reference's
    expression; Model_flow as a whole has no CPU path (its cost volume / warp / loss operators refuse host tensors)."""
    pw = PWC_tf()
    flow = torch.randn(2, 2, 4, 13)
    assert torch.equal(pw._up(flow, (8, 26), 2.0), F.interpolate(flow, scale_factor=2.0, mode='bilinear') * 2.0)      # pwc_tf.py:119
    assert torch.equal(pw._up(flow, (16, 52), 4.0), F.interpolate(flow * 4.0, [16, 52], mode='bilinear'))             # pwc_tf.py:174


def test_operator_argument_checks_come_before_the_device():
    """Shape errors are ValueErrors whatever the device; a well-formed CPU call is refused loudly (no CPU fallback)."""
    x = torch.randn(1, 2, 4, 6)
    with pytest.raises(ValueError):
        ops.upsample_bilinear_scaled(x, (6, 12), 2.0)                   # 6 is not a multiple of 4
    with pytest.raises(ValueError):
        ops.upsample_bilinear_scaled(x, (2, 6), 1.0)                    # down-sampling
    with pytest.raises(RuntimeError, match='MI355X'):
        ops.upsample_bilinear_scaled(x, (8, 12), 2.0)
    B = 3
    two, one = [torch.rand(2 * B)] * 2, [torch.rand(B)] * 2
    with pytest.raises(ValueError):
        ops.loss_combine(two, two, two, two)                            # consistency terms are [B]
    with pytest.raises(ValueError):
        ops.loss_combine(two, two, two[:1], one)                        # one term per scale and loss
    with pytest.raises(RuntimeError, match='MI355X'):
        ops.loss_combine(two, two, two, one)
    with pytest.raises(ValueError):
        ops.weighted_mean_sum(one, [1.0])
    with pytest.raises(RuntimeError, match='MI355X'):
        ops.weighted_mean_sum(one, [1.0, 2.0])
    with pytest.raises(ValueError):
        ops.flow_head(torch.zeros(1, 3, 4, 4), torch.zeros(2))
    with pytest.raises(ValueError):
        ops.to_nchw(torch.zeros(2, 4, 3, 3), dup_tail=3)
    y = torch.arange(2 * 4 * 3 * 3, dtype=torch.float32).view(2, 4, 3, 3)
    assert torch.equal(ops.to_nchw(y, dup_tail=1), torch.cat((y, y[1:]), 0))    # plain NCHW input: nothing to re-lay out


def test_total_loss_off_the_gpu_is_the_reference_formula():
    """FlowTrainer.total_loss (train.py:147-150): sum_k w_k * mean(loss_k); the fused operator is only taken for HIP tensors."""
    import types
    from unopticalflow_amd.trainer import FlowTrainer
    cfg = types.SimpleNamespace(w_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, lr=1e-4)
    tr = FlowTrainer(cfg, nn.Linear(2, 2), fused_adam=False)
    pack = {'loss_pixel': torch.rand(4), 'loss_ssim': torch.rand(4), 'loss_flow_smooth': torch.rand(4), 'loss_flow_consis': torch.rand(4)}
    want = 0.15 * pack['loss_pixel'].mean() + 0.85 * pack['loss_ssim'].mean() + 10.0 * pack['loss_flow_smooth'].mean() + 0.01 * pack['loss_flow_consis'].mean()
    torch.testing.assert_close(tr.total_loss(pack), want, rtol=1e-6, atol=0)


def test_configs_and_loss_weights_are_the_references():
    """tests/golden/g8_config.json: the reference's config/kitti.yaml and config/sintel.yaml as yaml reads them and what its own
    generate_loss_weights_dict (core/config/config_utils.py:3-9, imported unmodified by gen_golden.py) makes of them.  The package's yaml files carry
    the same value under every key they share (the flow stage's keys; machine paths aside), and the package's and the oracle's weighting give the
    reference's four numbers bit for bit (train.py:147-150 multiplies the batch means by exactly these)."""
    import json
    import types
    import yaml
    from oracle import ref_cpu as R
    from unopticalflow_amd import generate_loss_weights_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, 'tests', 'golden', 'g8_config.json')))
    for name in ('kitti', 'sintel'):
        ours = yaml.safe_load(open(os.path.join(root, 'unopticalflow_amd', 'config', name + '.yaml')))
        ref = doc[name]['yaml']
        shared = [k for k in ours if k in ref]
        assert {'dataset', 'img_hw', 'num_scales', 'num_iterations', 'w_ssim', 'w_flow_smooth', 'w_flow_consis', 'h_flow_consist_alpha',
                'h_flow_consist_beta'} <= set(shared), (name, shared)
        for k in shared:
            assert ours[k] == ref[k], (name, k, ours[k], ref[k])
        cfg = types.SimpleNamespace(**ours)
        assert generate_loss_weights_dict(cfg) == doc[name]['loss_weights'] == R.generate_loss_weights_dict(cfg)
        assert list(generate_loss_weights_dict(cfg)) == ['loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis']      # (insertion order: the loss pack's)


def test_weight_shadow_groups_split_the_cast_nodes():
    """WeightShadows(groups=k): k cast nodes over consecutive convolutions, so the eager data-parallel step's all-reduce pieces do
    not all wait for the end of backward; same values and gradients as one node."""
    torch.manual_seed(0)
    net = nn.Sequential(conv(3, 4), conv(4, 4), conv(4, 4), HeadConv2d(4, 2, 3, padding=1))
    with WeightShadows((net,), groups=3):
        nodes = [conv_weight(m).grad_fn for m in net.modules() if isinstance(m, nn.Conv2d)]
        assert len(set(nodes)) == 3 and nodes[0] is not nodes[-1]
        assert net[3].__dict__['_b_half'].dtype == torch.bfloat16
    with WeightShadows((net,), groups=99):
        assert len({conv_weight(m).grad_fn for m in net.modules() if isinstance(m, nn.Conv2d)}) == 4
    assert not any('_w_half' in m.__dict__ for m in net.modules())


def test_deferred_loss_sums_job_table(monkeypatch):
    """ops.deferred_loss_sums (round 5): the jobs a forward pass registers are finished by ONE ``unflow_loss_finalize_batch`` call whose
    arrays describe them as the entries' own second stages would: partial sums per sample from ``unflow_loss_partial_blocks`` (the
    kernels' tilings), the divisors in fp32 exactly as photo.hip / ssim.hip compute them, kind 1 only for the smoothness term, the
    partial-sum tensors kept alive until the flush.  No kernel runs: the C call is intercepted."""
    import ctypes
    calls = []
    monkeypatch.setattr(ops, '_call', lambda name, *a, **k: calls.append((name, a)))
    monkeypatch.setattr(ops, '_stream', lambda: ctypes.c_void_p(0))
    d = ops._DeferredLossSums()
    B, H, W = 4, 64, 208
    mk = lambda *shape: torch.zeros(*shape)
    with d:
        assert d.enabled
        d.add(mk(B * 64), mk(B), mk(B, 2), 0, H, W, 0, ops._f32(float(H) * W), ops._f32(float(H) * W))                   # masked mean
        d.add(mk(B * 64), mk(B), mk(B, 2), 1, H, W, 0, ops._f32(3.0 * H * W), ops._f32(float(H) * W), True)             # SSIM loss
        d.add(mk(B * 64), mk(B), None, 2, H, W, 1, ops._f32(2.0 * H * (W - 2)), ops._f32(2.0 * (H - 2) * W))            # smoothness
        d.add(mk(2 * 64), mk(2), mk(2, 2), 3, H, W, 0, ops._f32(2.0 * H * W), ops._f32(float(H) * W))                    # consistency, B / 2 samples
        assert len(d.jobs) == 4 and not calls
        d.flush()
        assert not d.jobs and len(calls) == 1 and d.launches == 1
    assert not d.enabled and len(calls) == 1                                  # (leaving the block: nothing left to finish)
    name, (P, L, S, N_, B_, K_, A_, C_, n, stream) = calls[0]
    assert name == 'unflow_loss_finalize_batch' and n == 4
    assert list(N_) == [7, 16, 32, 7]                                         # ceil(64 * 208 / 2048), 2 strips x 8 chunks of 8 rows, 4 x 8 tiles of 64 x 8
    assert list(B_) == [B, B, B, 2] and list(K_) == [0, 0, 1, 0]
    assert list(A_) == [13312.0, 39936.0, 26368.0, 26624.0] and list(C_) == [13312.0, 13312.0, 25792.0, 13312.0]
    assert S[2] is None and all(S[k] for k in (0, 1, 3)) and all(P[k] and L[k] for k in range(4))
    with pytest.raises(RuntimeError):
        d.add(mk(8), mk(1), None, 9, H, W, 0, 1.0, 1.0)                       # an operator the library does not know


def test_conv_split_emulation_helpers():
    """tests/conv_split_emulation.py (what profiles/r5_conv_split_emulation.md was measured with): the parts are bf16 numbers that add up
    to the value, the product sets are hi.hi + hi.lo + lo.hi / the six products of three parts, and a split convolution is a convolution
    to 2^-16 (two parts) / fp32 rounding (three parts), gradients included."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('conv_split_emulation', os.path.join(os.path.dirname(__file__), 'conv_split_emulation.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4096, generator=g)
    hi, lo = m.split_parts(x, 2)
    assert torch.equal(hi, hi.to(torch.bfloat16).float()) and torch.equal(lo, lo.to(torch.bfloat16).float())
    assert float((x - hi - lo).abs().max() / x.abs().max()) < 2.0 ** -16
    assert m.product_pairs(2) == [(0, 0), (0, 1), (1, 0)] and len(m.product_pairs(3)) == 6
    conv = torch.nn.Conv2d(5, 7, 3, padding=1)
    inp = torch.randn(2, 5, 9, 11, generator=g, requires_grad=True)
    ref = conv(inp)
    gr = torch.autograd.grad(ref.square().sum(), [inp, conv.weight, conv.bias])
    for parts, tol in ((2, 3e-5), (3, 2e-6)):
        with m.patched_convolutions(parts):
            out = conv(inp)
            gs = torch.autograd.grad(out.square().sum(), [inp, conv.weight, conv.bias])
        assert float((out - ref).abs().max() / ref.abs().max()) < tol
        for a, b in zip(gs, gr):
            assert float((a - b).abs().max() / b.abs().max()) < 4 * tol
    assert torch.equal(conv(inp), ref)                                # the patch is gone after the block


def test_fused_warp_corr_levels_are_a_config_key():
    """cfg.fused_warp_corr_levels picks the decoder levels whose warp + cost volume run as the fused kernel (bench.py --fused-levels);
    cfg.fused_warp_corr keeps meaning all four, the default none."""
    from oracle import ref_cpu as R
    from unopticalflow_amd import get_model
    for kw, want in ((dict(fused_warp_corr_levels='4,5'), {4, 5}), (dict(fused_warp_corr_levels=5), {5}), (dict(fused_warp_corr_levels=(3, 5)), {3, 5}),
                     (dict(fused_warp_corr=True), {2, 3, 4, 5}), (dict(), set())):
        assert get_model('flow')(R.default_cfg(**kw)).pwc_model.fused_levels == want, kw


def test_no_undefined_names_in_python_sources():
    """A typo in a branch only a GPU run takes (bench.py's roofline legs, a recipe's probe, a `-m gpu` test) would cost a GPU session to find.
    Every name LOADED inside a function of the product, bench.py, tools/ and tests/ is a builtin, a module-level name, or bound somewhere in
    the function (arguments, assignments, imports, nested definitions, enclosing class attributes) -- a crude scope model (no flow analysis),
    enough for misspelt and forgotten names."""
    import ast
    import builtins
    import glob
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def check(path):
        tree = ast.parse(open(path).read())
        mod = set(dir(builtins)) | {'__file__', '__name__', '__doc__'}
        for n in tree.body:
            for x in (ast.walk(n) if isinstance(n, (ast.If, ast.Try, ast.With, ast.For)) else [n]):
                if isinstance(x, (ast.Import, ast.ImportFrom)):
                    mod.update((a.asname or a.name).split('.')[0] for a in x.names)
                elif isinstance(x, (ast.FunctionDef, ast.ClassDef)):
                    mod.add(x.name)
                elif isinstance(x, (ast.Assign, ast.AugAssign, ast.AnnAssign, ast.For, ast.With)):
                    mod.update(y.id for y in ast.walk(x) if isinstance(y, ast.Name) and isinstance(y.ctx, ast.Store))
        for x in ast.walk(tree):
            if isinstance(x, ast.Global):
                mod.update(x.names)

        def bound(fn):
            s = set()
            for x in ast.walk(fn):
                if isinstance(x, ast.Name) and isinstance(x.ctx, (ast.Store, ast.Del)):
                    s.add(x.id)
                elif isinstance(x, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
                    s.add(x.name)
                elif isinstance(x, (ast.Import, ast.ImportFrom)):
                    s.update((a.asname or a.name).split('.')[0] for a in x.names)
                elif isinstance(x, ast.ExceptHandler) and x.name:
                    s.add(x.name)
                elif isinstance(x, ast.arg):
                    s.add(x.arg)
            return s
        bad = set()

        def visit(node, outer):
            for n in ast.iter_child_nodes(node):
                if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                    local = outer | bound(n)
                    bad.update((x.id, x.lineno) for x in ast.walk(n) if isinstance(x, ast.Name) and isinstance(x.ctx, ast.Load) and x.id not in local and x.id not in mod)
                elif isinstance(n, ast.ClassDef):
                    cls = {y.name for y in n.body if isinstance(y, (ast.FunctionDef, ast.ClassDef))}
                    for y in n.body:
                        if isinstance(y, (ast.Assign, ast.AnnAssign)):
                            cls.update(z.id for z in ast.walk(y) if isinstance(z, ast.Name) and isinstance(z.ctx, ast.Store))
                    visit(n, outer | cls)
                else:
                    visit(n, outer)
        visit(tree, set())
        return sorted(bad)
    files = (glob.glob(os.path.join(ROOT, 'unopticalflow_amd', '**', '*.py'), recursive=True) + glob.glob(os.path.join(ROOT, 'tools', '**', '*.py'), recursive=True)
             + glob.glob(os.path.join(ROOT, 'tests', '*.py')) + glob.glob(os.path.join(ROOT, 'tests', 'host_check', '*.py')) + glob.glob(os.path.join(ROOT, 'oracle', '*.py'))
             + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')])
    assert len(files) > 60
    findings = {os.path.relpath(f, ROOT): check(f) for f in files}
    assert not any(findings.values()), {f: b for f, b in findings.items() if b}


def test_gpu_recipes_parse_and_name_existing_files():
    """tools/*.sh are what a GPU session runs first: each parses (`bash -n`) and every repository path a recipe names -- tools/..., tests/...,
    bench.py, csrc files -- exists (a renamed probe would otherwise surface as a lost recipe on the GPU box)."""
    import glob
    import re
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(ROOT, 'tools', '*.sh')))
    assert scripts
    for sh in scripts:
        r = subprocess.run(['bash', '-n', sh], capture_output=True, text=True)
        assert r.returncode == 0, (sh, r.stderr)
    text = open(os.path.join(ROOT, 'tools', 'gpu_r6.sh')).read()
    named = set(re.findall(r'(?<![\w/$}])((?:tools|tests|unopticalflow_amd|oracle)/[\w/.]+\.(?:py|sh|cpp|hip|h))\b', text)) | set(re.findall(r'(?<![\w/])(bench\.py|__graft_entry__\.py)\b', text))
    assert len(named) >= 10
    missing = sorted(p for p in named if not os.path.exists(os.path.join(ROOT, p)))
    assert not missing, missing
