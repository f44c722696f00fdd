"""One Winograd launch shape for rocprofv3 --pmc passes: python one_wino_kind.py cin cout res kind (relu_in | style | plain | res)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, res = (int(v) for v in sys.argv[1:4])
kind = sys.argv[4]
b = 8
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
y = torch.empty(b, cout, res, res, device='cuda')
if kind == 'relu_in':
    kw = dict(in_mask=x, mask=(1.0, 0.0), bias=torch.randn(cout, device='cuda'))
elif kind == 'res':
    kw = dict(residual=torch.randn_like(y), out_mask=torch.randn_like(y), res_sub=torch.randn_like(y), res_coef=0.5)
elif kind == 'style':
    kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5, out_scale=torch.rand(b, cout, device='cuda') + 0.5, noise=torch.randn(b, 1, res, res, device='cuda'),
              noise_w=0.1, bias=torch.randn(cout, device='cuda'), act=conv.ACT_LRELU, gain=2 ** 0.5)
else:
    kw = dict(bias=torch.randn(cout, device='cuda'))
for _ in range(3):
    fc.forward(x, out=y, **kw)
torch.cuda.synchronize()
