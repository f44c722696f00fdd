import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, res, b, pad = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (512, 256, 64, 8, 0)
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
fc = conv.FrozenConv2d(w, 2, pad, transposed=True, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
oh, ow = fc.out_hw(res, res)
y = torch.empty(b, cout, oh, ow, device='cuda')
for _ in range(4):
    fc.forward(x, out=y)
torch.cuda.synchronize()
