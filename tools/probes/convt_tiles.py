"""Stride-2 transposed conv (fp32 kernel): auto tile against the forced narrow (tile_hint 1: 128 positions, 3 blocks per CU) and wide
(tile_hint 2: 256 positions, 2 per CU) tiles on the step's shapes — wave quantisation (blocks / resident blocks) decides more than the tile's
own efficiency on the mid-resolution layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
b = 8
for cin, cout, res, pad in [(512, 512, 4, 0), (512, 512, 8, 0), (512, 512, 16, 0), (512, 512, 32, 0), (512, 256, 64, 0), (256, 128, 128, 0), (128, 64, 256, 0), (64, 32, 512, 0),
                            (512, 512, 33, 0), (512, 256, 65, 0), (256, 128, 129, 0), (128, 64, 257, 0), (64, 32, 513, 0)]:
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(w, 2, pad, transposed=True, device='cuda')
    x = torch.randn(b, cin, res, res, device='cuda')
    oh, ow = fc.out_hw(res, res)
    y = torch.empty(b, cout, oh + 3, ow + 3, device='cuda') if res >= 32 and res % 2 == 0 else torch.empty(b, cout, oh, ow, device='cuda')
    kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5, out_scale=torch.rand(b, cout, device='cuda') + 0.5)
    t = {}
    for rd in range(5):
        for hint in (0, 1, 2):
            fc.forward(x, out=y, tile_hint=hint, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fc.forward(x, out=y, tile_hint=hint, **kw)
            e1.record(); torch.cuda.synchronize()
            if rd:
                t.setdefault(hint, []).append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * b * cout * cin * 9 * res * res
    print('%4d->%-4d @%-4d ' % (cin, cout, res) + '  '.join('hint %d: %.4f ms (%.0f TF)' % (h, np.median(v), fl / np.median(v) / 1e9) for h, v in t.items()), flush=True)
