"""Probe: conv trunk (pyramid + level-2 decoder + context shapes) forward+backward in NCHW vs channels_last with a plain
torch epilogue, MIOpen immediate mode and find mode.  Measures only what the layout does to MIOpen."""
import os, sys, time, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.backends.cudnn.benchmark = os.environ.get('FIND', '0') == '1'
if torch.backends.cudnn.benchmark:
    os.environ.setdefault('MIOPEN_USER_DB_PATH', '/tmp/clprobe_db'); os.makedirs('/tmp/clprobe_db', exist_ok=True)

def block(ci, co, dil=1, stride=1):
    return nn.Sequential(nn.Conv2d(ci, co, 3, stride, dil, dil), nn.LeakyReLU(0.1))

class Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        ch = [3, 16, 16, 32, 32]
        self.pyr = nn.Sequential(block(3, 16, stride=2), block(16, 16), block(16, 32, stride=2), block(32, 32))
        dd = (128, 128, 96, 64, 32)
        cins = (115, 128, 256, 224, 160)
        self.dec = nn.ModuleList([block(ci, co) for ci, co in zip(cins, dd)])
        self.ctx = nn.Sequential(block(34, 128, 1), block(128, 128, 2), block(128, 128, 4), block(128, 96, 8), block(96, 64, 16), block(64, 32, 1))
    def forward(self, img, x):
        p = self.pyr(img)
        x0 = self.dec[0](x); x1 = self.dec[1](x0)
        x2 = self.dec[2](torch.cat((x0, x1), 1)); x3 = self.dec[3](torch.cat((x1, x2), 1)); x4 = self.dec[4](torch.cat((x2, x3), 1))
        c = self.ctx(torch.cat((x4, x4[:, :2]), 1))
        return p.mean() + c.mean() + x4.mean()

def run(cl):
    torch.manual_seed(0)
    m = Trunk().cuda()
    img = torch.rand(24, 3, 256, 832, device='cuda'); x = torch.randn(16, 115, 64, 208, device='cuda')
    if cl:
        m = m.to(memory_format=torch.channels_last); img = img.contiguous(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_()
    amp = os.environ.get('AMP', '0') == '1'            # AMP=1: the bf16 conv-stack option (torch.autocast)

    def step():
        m.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            loss = m(img, x)
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3

for cl in (False, True, False, True):
    print('channels_last=%s find=%s bf16=%s: %.2f ms' % (cl, torch.backends.cudnn.benchmark, os.environ.get('AMP', '0'), run(cl)), flush=True)
