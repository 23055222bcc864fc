"""Run ONE C entry point at one pyramid-level shape a few times (target of a rocprofv3 --pmc pass: every kernel the
process launches after start-up belongs to that entry point).  python tools/pmc_entry.py <entry> <level> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopticalflow_amd import ops, _lib   # noqa: E402
from microbench import LEVELS, _smooth_flow   # noqa: E402

entry, lvl = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = 16
C, h, w = LEVELS[lvl]
lib = _lib.load()
P = ops._ptr
f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
gc = torch.randn(B, 81, h, w, device='cuda'); cv = torch.empty_like(gc)
o1, o2 = torch.empty_like(f1), torch.empty_like(f2)
fl = _smooth_flow(B, h, w); gfl = torch.empty_like(fl)
torch.cuda.synchronize()
for _ in range(reps):
    if entry == 'unflow_corr_fwd':
        lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, 4, ops._stream())
    elif entry == 'unflow_corr_bwd':
        lib.unflow_corr_bwd(P(f1), P(f2), P(gc), P(o1), P(o2), B, C, h, w, 4, ops._stream())
    elif entry == 'unflow_warp_fwd':
        lib.unflow_warp_fwd(P(f1), P(fl), P(o1), None, B, C, h, w, 0, ops._stream())
    elif entry == 'unflow_warp_bwd':
        lib.unflow_warp_bwd(P(f1), P(fl), P(f2), None, P(o1), P(gfl), B, C, h, w, 0, ops._stream())
    elif entry == 'unflow_warp_bwd_det':
        lib.unflow_warp_bwd_det(P(f1), P(fl), P(f2), None, P(o1), P(gfl), B, C, h, w, 0, ops._stream())
    elif entry == 'unflow_warp_corr_fwd':
        lib.unflow_warp_corr_fwd(P(f1), P(f2), P(fl), P(cv), B, C, h, w, 4, 0, ops._stream())
    else:
        raise SystemExit('unknown entry ' + entry)
torch.cuda.synchronize()
print('PMC_ENTRY %s [%d, %d, %d, %d] reps %d' % (entry, B, C, h, w, reps))
