"""One conv launch of a given family for rocprofv3 --pmc passes: python one_conv_kind.py s2|tr cin cout res kind."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
fam = sys.argv[1]
cin, cout, res = (int(v) for v in sys.argv[2:5])
kind = sys.argv[5]
b = 8
tr = fam == 'tr'
k, stride, pad = 3, 2, 0
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
fc = conv.FrozenConv2d(w, stride, pad, transposed=tr, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
oh, ow = fc.out_hw(res, res)
if tr:
    oh, ow = oh + 3, ow + 3
y = torch.empty(b, cout, oh, ow, device='cuda')
kw = {}
if kind == 'style':
    kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5, out_scale=torch.rand(b, cout, device='cuda') + 0.5)
elif kind == 'mask':
    kw = dict(in_mask=torch.randn_like(x), mask=(1.41, 0.28))
elif kind == 'plain' and not tr:
    kw = dict(bias=torch.randn(cout, device='cuda'), act=conv.ACT_LRELU, gain=2 ** 0.5)
for _ in range(3):
    fc.forward(x, out=y, **kw)
torch.cuda.synchronize()
