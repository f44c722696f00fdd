"""python tools/probes/convt_ab.py [batch]: the stride-2 transposed 3x3 launches of the 1024^2 step (generator up layers: style scale in, demodulation out)
on the fused four-parity kernel: ms per launch (median of five timed groups), TFLOP/s.  Run twice on the GPU box — as is (round 5: input tile by LDS-DMA) and
with L2I_CONVT_DMA=0 (the register-staged path of rounds 1-4) — for the A/B of profiles/r05_convt_ab.txt."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = torch.device('cuda')
SHAPES = [(64, 32, 512, 'G up 1024'), (128, 64, 256, 'G up 512'), (256, 128, 128, 'G up 256'), (512, 256, 64, 'G up 128'), (512, 512, 32, 'G up 64'), (512, 512, 16, 'G up 32')]
rs = np.random.RandomState(0)
print('batch', B, '(register-staged input: L2I_CONVT_DMA=0)' if os.environ.get('L2I_CONVT_DMA') == '0' else '(input tile by LDS-DMA)')
for cin, cout, res, tag in SHAPES:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 2, 0, transposed=True, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV)
    s, d = torch.rand(B, cin, device=DEV) + 0.5, torch.rand(B, cout, device=DEV) + 0.5
    y = torch.empty(B, cout, 2 * res + 4, 2 * res + 4, device=DEV)          # the generator's padded (2H + 4)^2 map
    flop = 2.0 * B * cout * cin * 9 * res * res
    for _ in range(2):
        fc.forward(x, out=y, in_scale=s, out_scale=d)
    torch.cuda.synchronize()
    n = max(3, int(8e-3 / (flop / 80e12)))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fc.forward(x, out=y, in_scale=s, out_scale=d)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    t = sorted(ts)[2]
    print('%4d->%4d @%4d %-10s %.3f ms %6.1f TF' % (cin, cout, res, tag, t, flop / t / 1e9), flush=True)
