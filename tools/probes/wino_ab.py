"""A/B timing of the Winograd kernel variants on the step's heavy 3x3 launches (plain, gradient-mask prologue, residual +
output-mask epilogue).  L2I_LIB_PATH selects the build."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv, _lib
if os.environ.get('L2I_LIB_PATH'):
    _lib.LIB_PATH = os.environ['L2I_LIB_PATH']
CASES = [(64, 64, 1024, 8), (128, 128, 512, 8), (64, 64, 512, 8), (256, 256, 128, 8), (512, 512, 64, 8), (32, 32, 1024, 8)]
out = []
for cin, cout, res, b in CASES:
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
    x = torch.randn(b, cin, res, res, device='cuda')
    y = torch.empty(b, cout, res, res, device='cuda')
    r = torch.randn(b, cout, res, res, device='cuda')
    bias = torch.randn(cout, device='cuda')
    for name, kw in (('plain', dict(bias=bias)), ('mask', dict(in_mask=x, mask=(1.0, 0.0), bias=bias)), ('res+omask', dict(residual=r, out_mask=r))):
        for _ in range(2):
            fc.forward(x, out=y, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fc.forward(x, out=y, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append('%dx%d@%d %s %.3fms %.0fTF' % (cin, cout, res, name, ms, 2 * b * cout * cin * 9 * res * res / ms / 1e9))
print(' | '.join(out))
