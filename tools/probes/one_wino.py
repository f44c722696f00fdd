import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, res, b = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 512, 64, 8)
if len(sys.argv) > 5:
    conv.USE_WINOGRAD = sys.argv[5] != "0"
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
y = torch.empty(b, cout, res, res, device='cuda')
for _ in range(4):
    fc.forward(x, out=y)
torch.cuda.synchronize()
