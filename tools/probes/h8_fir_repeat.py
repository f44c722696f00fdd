"""upfirdn2d_h8 (separable / generic) and torgb_fwd_h8 repeated many times on identical inputs, checksums on the device, one host sync at the end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
from latent2im_amd import kernels16 as K16
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b, dev = 4, 'cuda'
torch.manual_seed(0)
def h8(c, hh, ww): return (torch.randn(b, c // 8, hh, ww, 8, device=dev) * 0.7).to(BF)
kk = torch.tensor([1., 3., 3., 1.]); k2 = (kk[:, None] * kk[None, :]); k2 = (k2 / k2.sum() * 4).to(dev); sep = K16.separable(k2)
cases = {}
for c, r in ((512, 16), (256, 64), (64, 256), (32, 512)):
    x = h8(c, r + 1, r + 1); bias = torch.randn(c, device=dev)
    cases['fir sep %dch @%d' % (c, r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5, sep=sep)
    cases['fir sep plain %dch @%d' % (c, r)] = lambda x=x: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), sep=sep)
    cases['fir gen %dch @%d' % (c, r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5)
    y = h8(c, r, r); wm = torch.randn(b, 3, c, device=dev); z3 = torch.zeros(3, device=dev)
    cases['torgb %dch @%d' % (c, r)] = lambda y=y, wm=wm, z3=z3: K16.torgb_fwd(y, wm, z3)
for name, f in cases.items():
    sums = torch.zeros(reps, dtype=torch.float64, device=dev)
    for i in range(reps):
        sums[i] = f().float().double().abs().sum()
    torch.cuda.synchronize()
    s = sums.cpu().numpy()
    vals, counts = np.unique(s, return_counts=True)
    print('%-28s %3d distinct checksums in %d  %s' % (name, len(vals), reps, sorted(counts.tolist(), reverse=True)[:4]), flush=True)
