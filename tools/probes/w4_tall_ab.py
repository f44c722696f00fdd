"""python tools/probes/w4_tall_ab.py [batch]: the position-split F(4x4,3x3) kernel on its 64 x 8-pixel tile / four waves ('all') against the 64 x 16-pixel tile /
eight waves ('tall'), interleaved in one process on the step's 3x3 stride-1 shapes: ms per launch (median of seven groups), plain / style-scaled / ReLU-on-load."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = torch.device('cuda')
SHAPES = [(64, 64, 1024, 'vgg conv1_2'), (128, 128, 512, 'vgg conv2_2'), (64, 128, 512, 'vgg conv2_1'), (32, 32, 1024, 'G 1024'), (64, 64, 512, 'G 512'),
          (128, 128, 256, 'G 256'), (256, 256, 128, 'G 128'), (512, 512, 64, 'G 64'), (64, 64, 256, 'R layer1'), (128, 128, 128, 'R layer2'), (256, 256, 64, 'R layer3')]
rs = np.random.RandomState(0)
for cin, cout, res, tag in SHAPES:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV).relu_()
    s = torch.rand(B, cin, device=DEV) + 0.5
    y = torch.empty(B, cout, res, res, device=DEV)
    flop = 2.0 * B * cout * cin * 9 * res * res
    out = {}
    ident = True
    for kw, nm in ((dict(), 'plain'), (dict(in_scale=s), 'scale'), (dict(in_mask=x, mask=(1.0, 0.0)), 'relu')):
        ys = {}
        for rep in range(2):
            for mode in ('all', 'tall'):
                conv.WINO4 = mode
                for _ in range(2):
                    fc.forward(x, out=y, **kw)
                torch.cuda.synchronize()
                ys[mode] = y.clone()
                n = max(3, int(6e-3 / (flop / 150e12)))
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        fc.forward(x, out=y, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / n)
                out[(mode, nm)] = min(out.get((mode, nm), 1e9), sorted(ts)[3])
        ident &= bool(torch.equal(ys['all'], ys['tall']))
    conv.WINO4 = 'all'
    print('%4d->%4d @%4d  %-12s 64x8 %.3f / %.3f / %.3f ms | 64x16 %.3f / %.3f / %.3f ms  x%.3f x%.3f x%.3f  identical %s' %
          (cin, cout, res, tag, out[('all', 'plain')], out[('all', 'scale')], out[('all', 'relu')], out[('tall', 'plain')], out[('tall', 'scale')], out[('tall', 'relu')],
           out[('all', 'plain')] / out[('tall', 'plain')], out[('all', 'scale')] / out[('tall', 'scale')], out[('all', 'relu')] / out[('tall', 'relu')], ident), flush=True)
