"""Max-pool kernels at the shapes of the 1024^2 batch-8 step (GPU box; timing only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import kernels as K


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


x = torch.randn(8, 64, 1024, 1024, device='cuda')
print('VGG pool1 2x2/2 forward  [8,64,1024^2]: %.3f ms' % t(lambda: K.maxpool2d_fwd(x, 2, 2, 0)))
x = torch.randn(8, 64, 512, 512, device='cuda')
y, idx = K.maxpool2d_fwd(x, 3, 2, 1)
print('ResNet stem pool 3x3/2 forward [8,64,512^2]: %.3f ms' % t(lambda: K.maxpool2d_fwd(x, 3, 2, 1)))
gy = torch.randn_like(y)
print('ResNet stem pool 3x3/2 backward: %.3f ms' % t(lambda: K.maxpool2d_bwd(gy, idx, (512, 512), 3, 2, 1)))
