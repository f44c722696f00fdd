"""The vendor stack on the same MI355X: torch.nn.functional.conv2d / conv_transpose2d (MIOpen, fp32, benchmark mode so that it picks its
fastest algorithm — Winograd included) against the l2i kernels on the heavy layer shapes of the step.  Plain convolutions only (MIOpen would need
separate passes for the fused prologue / epilogue terms).  GPU box; timing only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from latent2im_amd import conv

torch.backends.cudnn.benchmark = True
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
SH = [  # cin, cout, k, stride, pad, transposed, res
    (64, 64, 3, 1, 1, False, 1024), (128, 128, 3, 1, 1, False, 512), (512, 512, 3, 1, 1, False, 64), (32, 32, 3, 1, 1, False, 1024),
    (256, 256, 3, 1, 1, False, 128), (64, 128, 3, 2, 0, False, 513), (256, 512, 3, 2, 0, False, 129), (128, 128, 3, 2, 1, False, 256),
    (512, 256, 3, 2, 0, True, 64), (128, 64, 3, 2, 0, True, 256),
]


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


for cin, cout, k, s, pad, tr, res in SH:
    B = 8
    x = torch.randn(B, cin, res, res, device='cuda')
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    fc = conv.FrozenConv2d(w, s, pad, transposed=tr, device='cuda')
    oh, ow = fc.out_hw(res, res)
    y = torch.empty(B, cout, oh, ow, device='cuda')
    ours = t(lambda: fc.forward(x, out=y))
    wd = w.cuda()
    if tr:
        wt = wd.transpose(0, 1).contiguous()
        vend = t(lambda: F.conv_transpose2d(x, wt, stride=2, padding=pad))
        macs = B * cout * cin * k * k * res * res
    else:
        vend = t(lambda: F.conv2d(x, wd, stride=s, padding=pad))
        macs = B * cout * cin * k * k * oh * ow
    print('%3d -> %3d %dx%d s%d %s @%4d^2   l2i %.3f ms %6.1f TF   MIOpen %.3f ms %6.1f TF' % (cin, cout, k, k, s, 'T' if tr else ' ', res, ours, 2 * macs / ours / 1e9, vend, 2 * macs / vend / 1e9))
