"""ResNet-50 stem input-gradient (7x7 / stride 2 / pad 3, 64 -> 3 at 1024^2 batch 8): the one-launch small-output kernel against the four
per-parity launches (GPU box; timing only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv

w = torch.randn(64, 3, 7, 7) / (3 * 49) ** 0.5
fc = conv.FrozenConv2d(w, 2, 3, device='cuda')
gy = torch.randn(8, 64, 512, 512, device='cuda')
a0 = torch.randn(8, 64, 512, 512, device='cuda')


def t(fused):
    conv.USE_FUSED_TRANSPOSED = fused
    for _ in range(3):
        out = fc.dgrad(gy, (1024, 1024), in_mask=a0, mask=(1.0, 0.0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = fc.dgrad(gy, (1024, 1024), in_mask=a0, mask=(1.0, 0.0))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10, out


a, ya = t(True)
b, yb = t(False)
print('one launch %.3f ms   four per-parity launches %.3f ms   max |diff| %.2e' % (a, b, float((ya - yb).abs().max())))
