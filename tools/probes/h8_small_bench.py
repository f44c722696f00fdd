"""the 16-bit conv kernel on the small-map shapes of the c5 step (batch 8): 64- against 32-channel blocks (L2I_H8_SMALL_WM1 is read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
conv.PRECISION = 'f16'
b = 8
for cin, cout, k, s, pad, res in ((512, 512, 3, 1, 1, 4), (512, 512, 3, 1, 1, 8), (512, 512, 3, 1, 1, 16), (512, 512, 3, 1, 1, 32), (512, 512, 3, 1, 1, 64), (1024, 256, 1, 1, 0, 64), (256, 1024, 1, 1, 0, 64),
                                (2048, 512, 1, 1, 0, 32), (512, 2048, 1, 1, 0, 32), (512, 512, 3, 2, 1, 64), (512, 512, 3, 1, 1, 32)):
    hc = conv.H8Conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, s, pad, device='cuda')
    x = torch.randn(b, cin // 8, res, res, 8, device='cuda').to(torch.float16)
    oh, ow = hc.out_hw(res, res)
    y = torch.empty(b, cout // 8, oh, ow, 8, device='cuda', dtype=torch.float16)
    bias = torch.randn(cout, device='cuda')
    for _ in range(5):
        hc.forward(x, out=y, bias=bias, act=conv.ACT_RELU)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hc.forward(x, out=y, bias=bias, act=conv.ACT_RELU)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print('%4d->%-4d k%d s%d @%-3d %.1f us' % (cin, cout, k, s, res, 1e3 * float(np.median(ts))), flush=True)
