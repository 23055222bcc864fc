"""torch.ops.unflow_hip.* (TORCH_LIBRARY registration of the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R


@pytest.fixture(scope='module')
def tops():
    import __graft_entry__ as ge
    ge.build()
    from unopticalflow_amd import torch_ops
    torch_ops.load()
    return torch.ops.unflow_hip


def test_schemas_and_shape_functions(tops):
    """Every operator of SURVEY 8b(1) is registered with a schema and a Meta / fake kernel: FakeTensor tracing gets the
    output shapes and dtypes without a GPU."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ('corr_fwd', 'corr_bwd', 'warp_fwd', 'warp_bwd', 'warp_corr_fwd', 'warp_corr_bwd', 'occ_weight', 'absdiff_bwd',
                 'masked_l1_fwd', 'masked_l1_bwd', 'ssim_loss_fwd', 'ssim_loss_bwd', 'smooth2_fwd', 'smooth2_bwd',
                 'consis_fwd', 'consis_bwd', 'corr', 'warp', 'warp_corr'):
        assert hasattr(tops, name), name
    with FakeTensorMode():
        B, C, H, W, d = 2, 5, 8, 12, 4
        f = torch.empty(B, C, H, W, device='cuda'); fl = torch.empty(B, 2, H, W, device='cuda')
        img = torch.empty(B, 3, H, W, device='cuda'); w = torch.empty(B, 1, H, W, device='cuda')
        assert tops.corr_fwd(f, f, d).shape == (B, 81, H, W)
        g1, g2 = tops.corr_bwd(f, f, torch.empty(B, 81, H, W, device='cuda'), d)
        assert g1.shape == f.shape and g2.shape == f.shape
        out, mask = tops.warp_fwd(img, fl, False, True)
        assert out.shape == img.shape and mask.shape == (B, 1, H, W) and mask.dtype == torch.uint8
        gs, gf = tops.warp_bwd(f, fl, f, None, False, True)
        assert gs.shape == f.shape and gf.shape == fl.shape
        assert tops.warp_corr_fwd(f, f, fl, d, False).shape == (B, 81, H, W)
        assert [t.shape for t in tops.warp_corr_bwd(f, f, fl, torch.empty(B, 81, H, W, device='cuda'), d, False)] == [f.shape, f.shape, fl.shape]
        o = tops.occ_weight(img, img, img)
        assert [t.dtype for t in o] == [torch.float32] * 4 + [torch.uint8] * 2 and all(t.shape == (B, 1, H, W) for t in o)
        loss, sums = tops.ssim_loss_fwd(img, torch.empty(2 * B, 3, H, W, device='cuda'), torch.empty(2 * B, 1, H, W, device='cuda'))
        assert loss.shape == (2 * B,) and sums.shape == (2 * B, 2)                # both directions over B centre images
        assert tops.smooth2_fwd(fl, img).shape == (B,) and tops.smooth2_bwd(fl, img, torch.empty(B, device='cuda')).shape == fl.shape
        loss, sums = tops.consis_fwd(fl, fl, w)
        assert loss.shape == (B,) and tops.consis_bwd(fl, fl, w, sums, loss).shape == fl.shape
        loss, sums = tops.masked_l1_fwd(w, w)
        assert tops.masked_l1_bwd(w, sums, loss).shape == w.shape
        # the differentiable composites trace too
        fr = torch.empty(B, C, H, W, device='cuda', requires_grad=True)
        assert tops.corr(fr, f, d).requires_grad and tops.warp_corr(fr, f, fl, d, False).shape == (B, 81, H, W)
    with pytest.raises(Exception):                       # CPU tensors: no kernel registered, no fallback
        tops.corr_fwd(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), 4)


@pytest.mark.gpu
def test_dispatcher_ops_match_the_ctypes_path(tops):
    """Same kernels behind both bindings: bit-equal forward values, equal gradients; and against the oracle."""
    from unopticalflow_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 32, 16, 52
    f1c = torch.randn(B, C, H, W, generator=g); f2c = torch.randn(B, C, H, W, generator=g)
    flc = torch.randn(B, 2, H, W, generator=g) * 1.5
    gcv = torch.randn(B, 81, H, W, generator=g).cuda()
    res = {}
    for name, fn in (('ctypes', lambda a, b, f: ops.corr(a, ops.warp_flow(b, f), 4)),
                     ('dispatcher', lambda a, b, f: tops.corr(a, tops.warp(b, f, False), 4)),
                     ('dispatcher fused', lambda a, b, f: tops.warp_corr(a, b, f, 4, False))):
        a, b, f = f1c.cuda().requires_grad_(), f2c.cuda().requires_grad_(), flc.cuda().requires_grad_()
        cv = fn(a, b, f)
        cv.backward(gcv)
        res[name] = (cv.detach(), a.grad, b.grad, f.grad)
    for name in ('dispatcher', 'dispatcher fused'):
        assert torch.equal(res[name][0], res['ctypes'][0]), name
        for x, y in zip(res[name][1:], res['ctypes'][1:]):
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-4, atol=1e-5 * y.abs().max().item())
    a, b, f = f1c.clone().requires_grad_(), f2c.clone().requires_grad_(), flc.clone().requires_grad_()
    ref = R.corr_naive(a, R.warp_flow(b, f), 4)
    np.testing.assert_allclose(res['dispatcher'][0].cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    # losses through the dispatcher: stacked directions over B centre images
    img = torch.rand(B, 3, H, W, generator=g).cuda()
    wl = (img + 0.1 * torch.randn(B, 3, H, W, generator=g).cuda()).clamp(0, 1)
    wr = (img + 0.1 * torch.randn(B, 3, H, W, generator=g).cuda()).clamp(0, 1)
    d_l, d_r, w_b, w_f, v_b, v_f = tops.occ_weight(img, wl, wr)
    e = ops.occ_weight(img, wl, wr)
    for x, y in zip((d_l, d_r, w_b, w_f, v_b, v_f), e):
        assert torch.equal(x, y)
    loss, _ = tops.ssim_loss_fwd(img, torch.cat((wl, wr)), torch.cat((w_b, w_f)))
    assert torch.equal(loss, torch.cat((ops.ssim_loss(img, wl, w_b), ops.ssim_loss(img, wr, w_f))))
