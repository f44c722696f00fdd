"""ONE process, two streams: conv_h8 launches keep stream A busy while upfirdn2d_h8 / torgb_fwd_h8 repeat on stream B with device-side checksums.
Does a co-resident bf16-MFMA kernel of the SAME process disturb the streaming kernels (it does from ANOTHER process: h8_fir_repeat.py beside h8_spin.py)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
if os.environ.get('L2I_ALT_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['L2I_ALT_LIB'])
from latent2im_amd import kernels16 as K16
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b, dev = 4, 'cuda'
torch.manual_seed(0)
def h8(c, hh, ww, bb=b): return (torch.randn(bb, c // 8, hh, ww, 8, device=dev) * 0.7).to(BF)
kk = torch.tensor([1., 3., 3., 1.]); k2 = (kk[:, None] * kk[None, :]); k2 = (k2 / k2.sum() * 4).to(dev); sep = K16.separable(k2)
w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
hc = conv.H8Conv(w, 1, 1, device=dev)
xa = h8(128, 64, 64, 8); ya = torch.empty_like(xa); bias_a = torch.randn(128, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
cases = {}
for c, r in ((256, 64), (64, 256), (32, 512)):
    x = h8(c, r + 1, r + 1); bias = torch.randn(c, device=dev)
    cases['fir sep %dch @%d' % (c, r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5, sep=sep)
    cases['fir gen %dch @%d' % (c, r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5)
    y = h8(c, r, r); wm = torch.randn(b, 3, c, device=dev); z3 = torch.zeros(3, device=dev)
    cases['torgb %dch @%d' % (c, r)] = lambda y=y, wm=wm, z3=z3: K16.torgb_fwd(y, wm, z3)
# conv_h8 itself as the victim: lean epilogue (packed scale + bias), general epilogue (residual + mask), transposed, 1x1
for cin, cout, k, st, pad, tr, res in ((64, 64, 3, 1, 1, False, 128), (256, 256, 3, 1, 1, False, 32), (64, 256, 1, 1, 0, False, 64), (128, 64, 3, 2, 0, True, 64)):
    wv = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    hv = conv.H8Conv(wv, st, pad, transposed=tr, device=dev)
    xv = h8(cin, res, res); bv = torch.randn(cout, device=dev); sc = torch.rand(b, cout, device=dev) + 0.5
    cases['conv lean %d->%d k%d%s @%d' % (cin, cout, k, 'T' if tr else '', res)] = lambda hv=hv, xv=xv, bv=bv, sc=sc, tr=tr: hv.forward(xv, out_scale=sc, bias=None if tr else bv, act=conv.ACT_LRELU, gain=2 ** 0.5)
    if not tr:
        oh, ow = hv.out_hw(res, res)
        rr = h8(cout, oh, ow); mk = h8(cout, oh, ow)
        cases['conv general %d->%d k%d @%d' % (cin, cout, k, res)] = lambda hv=hv, xv=xv, bv=bv, rr=rr, mk=mk: hv.forward(xv, bias=bv, residual=rr, out_mask=mk, mask=(1.0, 0.0), act=conv.ACT_RELU)
torch.cuda.synchronize()
for name, f in cases.items():
    sums = torch.zeros(reps, dtype=torch.float64, device=dev)
    for i in range(reps):
        with torch.cuda.stream(sa):
            for _ in range(4):
                hc.forward(xa, out=ya, bias=bias_a, act=conv.ACT_RELU)
        with torch.cuda.stream(sb):
            sums[i] = f().float().double().abs().sum()
    torch.cuda.synchronize()
    vals, counts = np.unique(sums.cpu().numpy(), return_counts=True)
    print('%-24s %3d distinct checksums in %d  %s' % (name, len(vals), reps, sorted(counts.tolist(), reverse=True)[:4]), flush=True)
