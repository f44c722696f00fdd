"""Timing of the 4x4 FIR fast path of l2i_upfirdn2d_f32 on the generator's blur shapes (GPU box).  L2I_LIB_PATH selects the build."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import kernels, _lib
if os.environ.get('L2I_LIB_PATH'):
    _lib.LIB_PATH = os.environ['L2I_LIB_PATH']
k1 = torch.tensor([1., 3., 3., 1.])
k = (k1[:, None] * k1[None, :]); k = (k / k.sum() * 4).cuda()
out = []
for c, res in ((32, 1025), (64, 513), (128, 257), (256, 129), (32, 1024), (64, 512), (32, 1028), (64, 516), (128, 260), (256, 132)):
    pad = (1, 1, 1, 1) if res % 2 else ((2, 1, 2, 1) if res % 8 == 0 else (1, -2, 1, -2))     # (2H+1)^2 natural / blur before a stride-2 conv / (2H+4)^2 padded up-layer map
    x = torch.randn(8, c, res, res, device='cuda')
    y = kernels.upfirdn2d(x, k, pad=pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        kernels.upfirdn2d(x, k, pad=pad, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out.append('%dch@%d %.3fms %.2fTB/s' % (c, res, ms, (x.numel() + y.numel()) * 4 / ms / 1e9))
print(' | '.join(out))
