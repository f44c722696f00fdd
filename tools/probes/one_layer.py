"""One conv layer launched a few times (for rocprofv3 counter passes): one_layer.py cin cout k stride pad res batch precision [mask]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, k, stride, pad, res, b = (int(v) for v in sys.argv[1:8])
conv.PRECISION = sys.argv[8] if len(sys.argv) > 8 else 'f32'
mask = len(sys.argv) > 9 and sys.argv[9] == 'mask'
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
fc = conv.FrozenConv2d(w, stride, pad, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
m = torch.randn(b, cin, res, res, device='cuda') if mask else None
oh, ow = fc.out_hw(res, res)
y = torch.empty(b, cout, oh, ow, device='cuda')
for _ in range(4):
    fc.forward(x, out=y, in_mask=m) if mask else fc.forward(x, out=y)
torch.cuda.synchronize()
