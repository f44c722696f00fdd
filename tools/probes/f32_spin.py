"""Keeps one fp32 conv shape (FrozenConv2d: Winograd / implicit GEMM / 1x1 GEMM / transposed) running for a few seconds, for clock / power sampling
beside it: python f32_spin.py cin cout k stride res seconds [T]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, k, stride, res = (int(v) for v in sys.argv[1:6])
secs = float(sys.argv[6])
tr = len(sys.argv) > 7 and sys.argv[7] == 'T'
b = 8
pad = 0 if (k == 1 or stride == 2) else 1
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
fc = conv.FrozenConv2d(w, stride, pad, transposed=tr, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
oh, ow = fc.out_hw(res, res)
y = torch.empty(b, cout, oh, ow, device='cuda')
bias = torch.randn(cout, device='cuda')
kw = {} if tr else dict(bias=bias)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        fc.forward(x, out=y, **kw)
    torch.cuda.synchronize(); n += 20
print('launches %d  avg %.4f ms' % (n, (time.time() - t0) / n * 1e3))
