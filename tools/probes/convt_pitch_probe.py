"""Transposed 3x3 conv of the generator's last up layer (64 -> 32, 512^2 -> 1025^2): natural (2H+1)^2 output against the (2H+4)^2 padded one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv


def t(fn, n=5):
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


for cin, cout, res in ((64, 32, 512), (128, 64, 256), (512, 512, 32)):
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(w, 2, 0, transposed=True, device='cuda')
    x = torch.randn(8, cin, res, res, device='cuda')
    s, d = torch.rand(8, cin, device='cuda') + 0.5, torch.rand(8, cout, device='cuda') + 0.5
    for extra in (1, 4):
        y = torch.empty(8, cout, 2 * res + extra, 2 * res + extra, device='cuda')
        a = t(lambda: fc.forward(x, out=y))
        b = t(lambda: fc.forward(x, out=y, in_scale=s, out_scale=d))
        print('%d->%d @%d out %d: plain %.3f ms, scaled %.3f ms' % (cin, cout, res, 2 * res + extra, a, b))
