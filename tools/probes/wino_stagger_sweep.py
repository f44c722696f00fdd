"""Sweep of the Winograd kernel's first-round stagger (L2I_WINO_STAGGER, units of 4096 cycles; -1 = the built-in rule, 0 = off) on the heavy
3x3 launches of the c3 step, interleaved rounds in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from latent2im_amd import conv

CASES = [(64, 64, 1024, 'relu_in'), (128, 128, 512, 'relu_in'), (64, 64, 1024, 'res'), (32, 32, 1024, 'style'), (64, 64, 512, 'style'), (128, 128, 256, 'style'),
         (256, 256, 128, 'style'), (512, 512, 64, 'style'), (256, 256, 64, 'plain'), (64, 64, 256, 'plain'), (64, 64, 512, 'mask')]
VALUES = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '-1', '2', '4', '8', '16'])]
b = 8
for cin, cout, res, kind in CASES:
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
    x = torch.randn(b, cin, res, res, device='cuda')
    y = torch.empty(b, cout, res, res, device='cuda')
    if kind == 'relu_in':
        kw = dict(in_mask=x, mask=(1.0, 0.0), bias=torch.randn(cout, device='cuda'))
    elif kind == 'res':
        r = torch.randn_like(y)
        kw = dict(residual=r, out_mask=torch.randn_like(y), res_sub=torch.randn_like(y), res_coef=0.5)
    elif kind == 'style':
        kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5, out_scale=torch.rand(b, cout, device='cuda') + 0.5, noise=torch.randn(b, 1, res, res, device='cuda'),
                  noise_w=0.1, bias=torch.randn(cout, device='cuda'), act=conv.ACT_LRELU, gain=2 ** 0.5)
    elif kind == 'mask':
        kw = dict(in_mask=torch.randn_like(x), mask=(1.0, 0.0))
    else:
        kw = dict(bias=torch.randn(cout, device='cuda'))
    times = {v: [] for v in VALUES}
    for rd in range(6):
        for v in (VALUES if rd % 2 == 0 else VALUES[::-1]):
            os.environ['L2I_WINO_STAGGER'] = str(v)
            fc.forward(x, out=y, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fc.forward(x, out=y, **kw)
            e1.record()
            torch.cuda.synchronize()
            if rd:
                times[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * b * cout * cin * 9 * res * res
    print('%4d->%-4d @%-4d %-8s ' % (cin, cout, res, kind) + '  '.join('s=%d: %.4f ms (%.0f TF)' % (v, np.median(times[v]), fl / np.median(times[v]) / 1e9) for v in VALUES), flush=True)
    del x, y, fc, kw
    torch.cuda.empty_cache()
