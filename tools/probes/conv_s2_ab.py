"""python tools/probes/conv_s2_ab.py [batch]: the 3x3 stride-2 layer shapes of the 1024^2 step on the DMA-staged kernel of round 5 (l2i_conv_s2.hip, the
dispatch's choice) and on the generic register-staged kernel (tile_hint 2 = the (2, 2) tile the dispatch used to pick for them, and the generic
kernel's own best tile through L2I_CONV_S2_DMA=0 in a second process), interleaved: ms per launch (median of five timed groups), TFLOP/s.
Usage on the GPU box:  python tools/probes/conv_s2_ab.py 8; L2I_CONV_S2_DMA=0 python tools/probes/conv_s2_ab.py 8"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = torch.device('cuda')
SHAPES = [(32, 64, 1028, 0, False, 'D 1024'), (64, 128, 516, 0, False, 'D 512'), (128, 256, 260, 0, False, 'D 256'), (256, 512, 132, 0, False, 'D 128'), (512, 512, 68, 0, False, 'D 64'),
          (64, 32, 1028, 0, True, 'G up 1024 dgrad'), (128, 64, 516, 0, True, 'G up 512 dgrad'), (256, 128, 260, 0, True, 'G up 256 dgrad'), (512, 256, 132, 0, True, 'G up 128 dgrad'),
          (128, 128, 256, 1, False, 'R layer2'), (256, 256, 128, 1, False, 'R layer3'), (512, 512, 64, 1, False, 'R layer4')]
rs = np.random.RandomState(0)
dma_off = os.environ.get('L2I_CONV_S2_DMA') == '0'
print('batch', B, '(generic kernel only: L2I_CONV_S2_DMA=0)' if dma_off else '')
for cin, cout, res, pad, scale, tag in SHAPES:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 2, pad, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV)
    kw = dict(in_scale=torch.rand(B, cin, device=DEV) + 0.5) if scale else {}
    oh = (res + 2 * pad - 3) // 2 + 1
    y = torch.empty(B, cout, oh, oh, device=DEV)
    flop = 2.0 * B * cout * cin * 9 * oh * oh
    out = {}
    for hint in (0, 2):
        for _ in range(2):
            fc.forward(x, out=y, tile_hint=hint, **kw)
        torch.cuda.synchronize()
        n = max(3, int(8e-3 / (flop / 90e12)))
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fc.forward(x, out=y, tile_hint=hint, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n)
        out[hint] = sorted(ts)[2]
    print('%4d->%4d @%4d pad %d %-5s %-16s dispatch %.3f ms %6.1f TF | generic (2,2) tile %.3f ms %6.1f TF | x%.2f'
          % (cin, cout, res, pad, 'scale' if scale else '', tag, out[0], flop / out[0] / 1e9, out[2], flop / out[2] / 1e9, out[2] / out[0]), flush=True)
