"""One l2i_conv2d_h8 / l2i_conv_transpose2d_h8 launch shape for rocprofv3 --pmc passes: python one_h8.py cin cout k stride res [T] (batch 8, bias + ReLU epilogue)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
cin, cout, k, stride, res = (int(v) for v in sys.argv[1:6])
tr = len(sys.argv) > 6 and sys.argv[6] == 'T'
b = 8
pad = 0 if (k == 1 or stride == 2) else 1
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
hc = conv.H8Conv(w, stride, pad, transposed=tr, device='cuda')
x = torch.randn(b, cin // 8, res, res, 8, device='cuda').to(torch.bfloat16)
oh, ow = hc.out_hw(res, res)
y = torch.empty(b, cout // 8, oh, ow, 8, device='cuda', dtype=torch.bfloat16)
bias = torch.randn(cout, device='cuda')
for _ in range(3):
    hc.forward(x, out=y, bias=bias, act=conv.ACT_RELU)
torch.cuda.synchronize()
