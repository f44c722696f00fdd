"""ONE process, two streams: a bf16-MFMA kernel (the 3-term split conv of --precision bf16x3, or conv_h8) keeps stream A busy while fp32 streaming
kernels (upfirdn2d, ToRGB, fused_bias_act, dot_reduce) repeat on stream B.  Were round 2's bf16x3 steps exposed to the packed-fp32 effect?
usage: [L2I_LIB=<a build with l2i_stream.hip compiled WITH the SLP vectorizer>] python tools/probes/f32_two_streams.py [repeats] [x3|h8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
from latent2im_amd import kernels as K
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
agg = sys.argv[2] if len(sys.argv) > 2 else 'x3'
b, dev = 4, 'cuda'
torch.manual_seed(0)
w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
if agg == 'x3':
    conv.PRECISION = 'bf16x3'
    fc = conv.FrozenConv2d(w, 1, 1, device=dev)
    xa = torch.randn(8, 128, 64, 64, device=dev); ya = torch.empty_like(xa); ba = torch.randn(128, device=dev)
    aggress = lambda: fc.forward(xa, out=ya, bias=ba)
else:
    hc = conv.H8Conv(w, 1, 1, device=dev)
    xa = (torch.randn(8, 16, 64, 64, 8, device=dev)).to(torch.bfloat16); ya = torch.empty_like(xa); ba = torch.randn(128, device=dev)
    aggress = lambda: hc.forward(xa, out=ya, bias=ba, act=conv.ACT_RELU)
kk = torch.tensor([1., 3., 3., 1.]); k2 = (kk[:, None] * kk[None, :]); k2 = (k2 / k2.sum() * 4).to(dev)
cases = {}
for c, r in ((256, 64), (64, 256), (32, 512)):
    x = torch.randn(b, c, r + 4, r + 4, device=dev); bias = torch.randn(c, device=dev); nz = torch.randn(b, 1, r, r, device=dev)
    cases['upfirdn2d blur+epi %dch @%d' % (c, r)] = lambda x=x, bias=bias, nz=nz: K.upfirdn2d(x, k2, pad=(1, -2, 1, -2), noise=nz, noise_w=0.1, bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5)
    y = torch.randn(b, c, r, r, device=dev); wm = torch.randn(b, 3, c, device=dev); z3 = torch.zeros(3, device=dev)
    cases['torgb %dch @%d' % (c, r)] = lambda y=y, wm=wm, z3=z3: K.torgb_fwd(y, wm, z3)
    cases['up-2 FIR (skip) 3ch @%d' % r] = lambda r=r: K.upfirdn2d(torch.ones(b, 3, r // 2, r // 2, device=dev) * 0.37, k2, up=(2, 2), pad=(2, 1, 2, 1))
    cases['dot_reduce %dch @%d' % (c, r)] = lambda y=y: K.dot_reduce(y, y)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for name, f in cases.items():
    sums = torch.zeros(reps, dtype=torch.float64, device=dev)
    for i in range(reps):
        with torch.cuda.stream(sa):
            for _ in range(4):
                aggress()
        with torch.cuda.stream(sb):
            sums[i] = f().float().double().abs().sum()
    torch.cuda.synchronize()
    vals, counts = np.unique(sums.cpu().numpy(), return_counts=True)
    print('%-34s %3d distinct checksums in %d  %s' % (name, len(vals), reps, sorted(counts.tolist(), reverse=True)[:4]), flush=True)
