import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, res, b, hint = 256, 256, 128, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
x = torch.randn(b, cin, res, res, device='cuda')
y = torch.empty(b, cout, res, res, device='cuda')
for _ in range(6):
    fc.forward(x, out=y, tile_hint=hint)
torch.cuda.synchronize()
