"""Where the HOST time of an eager train step goes (cProfile over 10 steps after warm-up).  python tools/probes/host_profile.py [bf16]"""
import cProfile
import os
import pstats
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unopticalflow_amd import get_model, tuning          # noqa: E402
from unopticalflow_amd.trainer import FlowTrainer        # noqa: E402

prec = 'bf16' if 'bf16' in sys.argv else 'fp32'
tuning.enable_miopen_tuning()
cfg = types.SimpleNamespace(mode='flow', dataset='kitti_depth', num_scales=3, h_flow_consist_alpha=3.0, h_flow_consist_beta=0.05,
                            w_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, lr=1e-4, align_corners=False, precision=prec,
                            channels_last=tuning.default_channels_last())
torch.manual_seed(0)
model = get_model('flow')(cfg).cuda()
tr = FlowTrainer(cfg, model)
x = torch.rand(8, 3, 768, 832, device='cuda')
for _ in range(5):
    tr.step(x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(x)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host enqueue %.2f ms/step (profiled, inflated by cProfile)' % ((t1 - t0) * 100))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
