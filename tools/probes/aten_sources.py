"""Which Python lines (forward) and autograd nodes (backward) launch the ATen / runtime kernels of an eager train step:
torch.profiler over 3 steps, kernels grouped by (kernel name, launching op, innermost repo frame).
python3 tools/probes/aten_sources.py [bf16]  ->  a table sorted by device time per step."""
import collections
import os
import re
import sys
import types

import torch
from torch.profiler import profile, ProfilerActivity

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
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        tr.step(x)
    torch.cuda.synchronize()

events = prof.events()
# the CPU op that launched each kernel: FunctionEvent.kernels lists them per op; keep the innermost op (the one with kernels and
# no child that also has them)
rows = collections.defaultdict(lambda: [0, 0.0])
for ev in events:
    if not getattr(ev, 'kernels', None):
        continue
    if any(getattr(c, 'kernels', None) for c in ev.cpu_children):
        continue
    frame = ''
    for fr in (ev.stack or []):
        if 'unopticalflow_amd' in fr and 'torch/' not in fr:
            frame = re.sub(r'.*unopticalflow_amd/', '', fr)
            break
    scope = ev.cpu_parent
    node = ''
    while scope is not None:
        if scope.name.startswith('autograd::engine::evaluate_function'):
            node = scope.name.split(': ')[-1]
            break
        scope = scope.cpu_parent
    for k in ev.kernels:
        kn = re.sub(r'void |at::native::|\(anonymous namespace\)::', '', k.name)[:70]
        if not re.search(r'elementwise|Fill|fill|Cat|copy|reduce|upsample|multi_tensor', k.name):
            continue
        key = (kn, ev.name[:28], node[:36] or frame[:60])
        rows[key][0] += 1
        rows[key][1] += k.duration
print('%-72s %-28s %-60s %8s %9s' % ('kernel', 'op', 'autograd node / frame', 'calls', 'us/step'))
tot = 0.0
for key, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tot += us / STEPS
    if us / STEPS >= 3.0:
        print('%-72s %-28s %-60s %8.1f %9.1f' % (key[0], key[1], key[2], n / STEPS, us / STEPS))
print('total ATen-ish device time %.1f us/step' % tot)
