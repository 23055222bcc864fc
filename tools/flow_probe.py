"""What do the flows of the bench's random-initialised network look like per pyramid level, and how long does the feature-warp
backward take on exactly those flows (back-to-back launches) against the microbench's smooth synthetic field?"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopticalflow_amd import get_model, ops, _lib   # noqa: E402
from tools.microbench import timeit, _smooth_flow   # noqa: E402

cfg = types.SimpleNamespace(mode='flow', dataset='kitti_depth', num_scales=3, h_flow_consist_alpha=3.0, h_flow_consist_beta=0.05,
                            w_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, lr=1e-4, align_corners=False, precision='fp32')
torch.manual_seed(1234)
model = get_model('flow')(cfg).cuda()
x = torch.rand(8, 3, 768, 832, device='cuda')
seen = []
orig = ops.warp_flow


def spy(src, flow, use_mask=False, align_corners=False):
    if not use_mask and src.shape[1] >= 8:
        seen.append((src.detach().clone(), flow.detach().clone()))
    return orig(src, flow, use_mask=use_mask, align_corners=align_corners)


ops.warp_flow = spy
import unopticalflow_amd.core.networks.structures.net_utils as nu   # noqa: E402
nu.ops.warp_flow = spy
with torch.no_grad():
    model(x)
lib = _lib.load(); P = ops._ptr
for src, fl in seen:
    B, C, h, w = src.shape
    g = torch.randn_like(src); gsrc = torch.empty_like(src); gfl = torch.empty_like(fl)
    dx = (fl[:, :, :, 1:] - fl[:, :, :, :-1]).abs().mean().item()
    t_net = timeit(lambda: lib.unflow_warp_bwd(P(src), P(fl), P(g), None, P(gsrc), P(gfl), B, C, h, w, 0, ops._stream()))
    sm = _smooth_flow(B, h, w)
    t_sm = timeit(lambda: lib.unflow_warp_bwd(P(src), P(sm), P(g), None, P(gsrc), P(gfl), B, C, h, w, 0, ops._stream()))
    zf = torch.zeros_like(fl)
    t_z = timeit(lambda: lib.unflow_warp_bwd(P(src), P(zf), P(g), None, P(gsrc), P(gfl), B, C, h, w, 0, ops._stream()))
    print('[%d,%d,%d,%d] flow |u| mean %.3f max %.2f, mean |du/dx| %.3f : warp_bwd %.1f us on the network flow, %.1f on the smooth synthetic one, %.1f on zero flow'
          % (B, C, h, w, fl.abs().mean().item(), fl.abs().max().item(), dx, t_net, t_sm, t_z), flush=True)
