"""[r6] The position-split F(4x4,3x3) kernel on the step's shapes WITH the epilogues the step gives them (tools/probes/wino4_bench.py times bare launches):
  G  = style-scaled input, demodulation, noise, bias, leaky ReLU (generator.py StyledConv)      R  = bias + ReLU (ResNet-50 conv2 forward)
  Rg = out_mask (ResNet-50 conv2 input gradient)                                               V  = ReLU-on-load + bias (VGG conv1_2 / conv2_2)
usage: python tools/probes/w4_epi_ab.py [batch]; L2I_LIB=<another build> for the other arm of an A/B (run the arms alternately)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = 'cuda'
CASES = [(32, 32, 1024, 'G'), (64, 64, 512, 'G'), (128, 128, 256, 'G'), (256, 256, 128, 'G'), (512, 512, 64, 'G'),
         (64, 64, 256, 'R'), (64, 64, 256, 'Rg'), (128, 128, 128, 'R'), (128, 128, 128, 'Rg'), (256, 256, 64, 'R'),
         (64, 64, 1024, 'V'), (128, 128, 512, 'V'), (64, 64, 1024, 'Rg')]
rs = np.random.RandomState(0)
torch.manual_seed(0)
tot = 0.0
for cin, cout, res, kind in CASES:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV)
    y = torch.empty(B, cout, res, res, device=DEV)
    bias = torch.randn(cout, device=DEV)
    if kind == 'G':
        kw = dict(in_scale=torch.rand(B, cin, device=DEV) + 0.5, out_scale=torch.rand(B, cout, device=DEV) + 0.5, noise=torch.randn(B, 1, res, res, device=DEV), noise_w=0.05,
                  bias=bias, act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    elif kind == 'R':
        kw = dict(bias=bias, act=conv.ACT_RELU)
    elif kind == 'Rg':
        kw = dict(out_mask=torch.randn(B, cout, res, res, device=DEV))
    else:
        kw = dict(in_mask=x, mask=(1.0, 0.0), bias=bias)
    flop = 2.0 * B * cout * cin * 9 * res * res
    for _ in range(2):
        fc.forward(x, out=y, **kw)
    torch.cuda.synchronize()
    n = max(3, int(8e-3 / (flop / 150e12)))
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fc.forward(x, out=y, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    ms = sorted(ts)[3]
    tot += ms
    print('%4d->%4d @%4d %-2s  %.4f ms  %6.1f TFLOP/s  checksum %.6e' % (cin, cout, res, kind, ms, flop / ms / 1e9, float(y.double().sum())), flush=True)
    del x, y, kw, fc
    torch.cuda.empty_cache()
print('sum %.3f ms' % tot)
