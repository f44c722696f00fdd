"""python tools/probes/wino4_bench.py [batch]: the 3x3 stride-1 layer shapes of the 1024^2 step on the F(2x2,3x3) kernel ('off'), the round-4 F(4x4,3x3)
kernel ('r4') and the round-5 position-split F(4x4,3x3) kernel ('all'), interleaved in one process: ms per launch (median of five timed groups),
TFLOP/s of the dense correlation, and the error against a float64 correlation (one sample)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from latent2im_amd import conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = torch.device('cuda')
SHAPES = [(64, 64, 1024, 'vgg conv1_2'), (128, 128, 512, 'vgg conv2_2'), (64, 128, 512, 'vgg conv2_1'), (32, 32, 1024, 'G 1024'), (64, 64, 512, 'G 512 / R layer1@256 x4'),
          (128, 128, 256, 'G 256'), (256, 256, 128, 'G 128'), (512, 512, 64, 'G 64'), (64, 64, 256, 'R layer1'), (128, 128, 128, 'R layer2'), (256, 256, 64, 'R layer3')]
rs = np.random.RandomState(0)
print('batch', B)
for cin, cout, res, tag in SHAPES:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV).relu_()
    s = torch.rand(B, cin, device=DEV) + 0.5
    y = torch.empty(B, cout, res, res, device=DEV)
    flop = 2.0 * B * cout * cin * 9 * res * res
    out = {}
    for mode in ('off', 'r4', 'all'):
        conv.WINO4 = mode
        for kw, nm in ((dict(), 'plain'), (dict(in_scale=s), 'scale'), (dict(in_mask=x, mask=(1.0, 0.0)), 'relu')):
            for _ in range(2):
                fc.forward(x, out=y, **kw)
            torch.cuda.synchronize()
            n = max(3, int(8e-3 / (flop / 150e12)))
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fc.forward(x, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / n)
            out[(mode, nm)] = sorted(ts)[2]
        # error on sample 0 against float64 (CPU)
        ref = F.conv2d(x[:1].double().cpu(), wt.double(), padding=1)
        fc.forward(x, out=y)
        out[(mode, 'err')] = float((y[:1].double().cpu() - ref).abs().max() / ref.abs().max())
    conv.WINO4 = 'all'
    o = lambda m, k: out[(m, k)]
    print('%4d->%4d @%4d  %-24s F2 %.3f / %.3f / %.3f ms | r4 %.3f / %.3f / %.3f | r5 %.3f / %.3f / %.3f ms (plain / scale / relu) %6.1f TF err %.1e | r5 vs r4 x%.2f x%.2f x%.2f'
          % (cin, cout, res, tag, o('off', 'plain'), o('off', 'scale'), o('off', 'relu'), o('r4', 'plain'), o('r4', 'scale'), o('r4', 'relu'),
             o('all', 'plain'), o('all', 'scale'), o('all', 'relu'), flop / o('all', 'plain') / 1e9, o('all', 'err'),
             o('r4', 'plain') / o('all', 'plain'), o('r4', 'scale') / o('all', 'scale'), o('r4', 'relu') / o('all', 'relu')), flush=True)
