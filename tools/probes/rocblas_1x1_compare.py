"""How fast does the vendor fp32 GEMM (torch.bmm -> rocBLAS / hipBLASLt) run the 1x1 layers of ResNet-50 at the step's shapes, against
l2i_conv2d_f32's DMA-fed GEMM (plain GEMM only: no bias / residual / ReLU epilogue on the vendor side)?  GPU box; timing only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv

SH = [(256, 1024, 64), (64, 256, 256), (128, 512, 128), (1024, 256, 64), (512, 128, 128), (512, 2048, 32), (2048, 512, 32), (256, 64, 256)]


def t(fn, n=20):
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


for cin, cout, res in SH:
    B = 8
    x = torch.randn(B, cin, res, res, device='cuda')
    w = torch.randn(cout, cin, 1, 1) / cin ** 0.5
    fc = conv.FrozenConv2d(w, 1, 0, device='cuda')
    y = torch.empty(B, cout, res, res, device='cuda')
    ours = t(lambda: fc.forward(x, out=y))
    wd = w.reshape(cout, cin).cuda()
    x3 = x.reshape(B, cin, res * res)
    y3 = torch.empty(B, cout, res * res, device='cuda')
    vend = t(lambda: torch.matmul(wd, x3, out=y3))
    fl = 2.0 * B * cin * cout * res * res
    print('%4d -> %4d @%3d^2  ours %.3f ms %5.1f TF   vendor bmm %.3f ms %5.1f TF' % (cin, cout, res, ours, fl / ours / 1e9, vend, fl / vend / 1e9))
