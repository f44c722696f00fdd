"""l2i_conv_img_h8 on the three image-side shapes of the c5 step (batch 8, 1024^2): ms and GB/s of the algorithmic bytes (image in, h8 out, sq_ref in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
conv.PRECISION = os.environ.get('PREC', 'f16')
b, res = 8, 1024
for name, cout, k, s, pad, with_sq in (('vgg conv1_1', 64, 3, 1, 1, False), ('vgg conv1_1 + sq', 64, 3, 1, 1, True), ('D from-RGB', 32, 1, 1, 0, False), ('resnet stem', 64, 7, 2, 3, False)):
    ic = conv.ImgConvH8(torch.randn(cout, 3, k, k) / (3 * k * k) ** 0.5, s, pad, device='cuda')
    x = torch.randn(b, 3, res, res, device='cuda')
    bias = torch.randn(cout, device='cuda')
    oh, ow = ic.out_hw(res, res)
    ref = torch.randn(b, cout // 8, oh, ow, 8, device='cuda').to(conv.h8_dtype()) if with_sq else None
    run = lambda: ic.forward(x, bias=bias, act=conv.ACT_RELU, sq=None if ref is None else (ref, torch.zeros(_lib.SQ_SLOTS, device='cuda'), [False]))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ms = float(np.median(ts))
    by = x.numel() * 4 + b * cout * oh * ow * 2 * (2 if with_sq else 1)
    print('%-18s %.4f ms  %.0f GB/s' % (name, ms, by / ms / 1e6), flush=True)
