"""python tools/probes/one_wino4.py cin cout res batch [mode=all|off] [kind=plain|style|relu_in] [reps]: launches of one 3x3 layer on the F(4x4,3x3) kernel
(mode off: F(2x2,3x3)); prints ms per launch (for the ablation runs and the counter passes)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, res, b = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 512, 64, 8)
conv.WINO4 = sys.argv[5] if len(sys.argv) > 5 else 'all'
kind = sys.argv[6] if len(sys.argv) > 6 else 'plain'
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
y = torch.empty(b, cout, res, res, device='cuda')
kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5) if kind == 'style' else (dict(in_mask=x, mask=(1.0, 0.0)) if kind == 'relu_in' else {})
import gc; gc.disable()
for _ in range(3):
    fc.forward(x, out=y, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fc.forward(x, out=y, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print('%s %d->%d @%d b%d %s %s: %.4f ms  %.1f TFLOP/s algorithmic' % (os.path.basename(os.environ.get('L2I_LIB', 'libl2i_hip.so')), cin, cout, res, b, conv.WINO4, kind, ms,
                                                                 2.0 * b * cout * cin * 9 * res * res / ms / 1e9))
