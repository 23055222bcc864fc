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
C, h, w = LEVELS['L2'] if lvl.startswith('S') else LEVELS[lvl]
lib = _lib.load()
P = ops._ptr
f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
gc = torch.randn(B, 81, h, w, device='cuda'); cv = torch.empty_like(gc)
o1, o2 = torch.empty_like(f1), torch.empty_like(f2)
fl = _smooth_flow(B, h, w); gfl = torch.empty_like(fl)
if lvl.startswith('S'):                                  # loss entries at image scale S0 / S1 / S2 (16 directed warps, 8 centre images)
    sc = int(lvl[1:])
    H, W = 256 >> sc, 832 >> sc
    img = torch.rand(8, 3, H, W, device='cuda'); wp = torch.rand(16, 3, H, W, device='cuda'); wt = torch.rand(16, 1, H, W, device='cuda')
    loss = torch.empty(16, device='cuda'); sums = torch.ones(16, 2, device='cuda') * (H * W / 2); gl = torch.ones(16, device='cuda'); gw = torch.empty_like(wp)
    part = ops._partials(16, H, W, img.device)
    torch.cuda.synchronize()
    for _ in range(reps):
        if entry == 'unflow_ssim_loss_bwd':                  # (sums: any positive weight sum -- only this entry's kernels may run here)
            lib.unflow_ssim_loss_bwd(P(img), P(wp), P(wt), P(sums), P(gl), P(gw), 16, H, W, 8, ops._stream())
        else:
            lib.unflow_ssim_loss_fwd(P(img), P(wp), P(wt), P(loss), P(sums), P(part), 16, H, W, 8, ops._stream())
    torch.cuda.synchronize()
    print('PMC_ENTRY %s [16, 3, %d, %d] reps %d' % (entry, H, W, reps))
    raise SystemExit(0)
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
    elif entry == 'unflow_warp_bwd_fused':
        tab = torch.empty(lib.unflow_warp_bwd_table_bytes(B, C, h, w), dtype=torch.uint8, device='cuda')
        lib.unflow_warp_bwd_fused(P(f1), P(fl), P(f2), P(o1), P(gfl), P(tab), 0, B, C, h, w, 0, ops._stream())
    elif entry == 'unflow_warp_corr_fwd':
        lib.unflow_warp_corr_fwd(P(f1), P(f2), P(fl), P(cv), B, C, h, w, 4, 0, ops._stream())
    else:
        raise SystemExit('unknown entry ' + entry)
torch.cuda.synchronize()
print('PMC_ENTRY %s [%d, %d, %d, %d] reps %d' % (entry, B, C, h, w, reps))
