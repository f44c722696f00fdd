"""Keeps one l2i_conv2d_h8 shape running for a few seconds (for clock / power sampling beside it): python h8_spin.py cin cout k stride res seconds"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv, _lib
if os.environ.get('L2I_ALT_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['L2I_ALT_LIB'])      # e.g. a timing-ablation build (tools/probes/h8_ablate.sh)
cin, cout, k, stride, res = (int(v) for v in sys.argv[1:6])
secs = float(sys.argv[6])
b = 8
pad = 0 if (k == 1 or stride == 2) else 1
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
hc = conv.H8Conv(w, stride, pad, device='cuda')
x = torch.randn(b, cin // 8, res, res, 8, device='cuda').to(torch.bfloat16)
oh, ow = hc.out_hw(res, res)
y = torch.empty(b, cout // 8, oh, ow, 8, device='cuda', dtype=torch.bfloat16)
bias = torch.randn(cout, device='cuda')
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        hc.forward(x, out=y, bias=bias, act=conv.ACT_RELU)
    torch.cuda.synchronize(); n += 50
print('launches %d  avg %.4f ms' % (n, (time.time() - t0) / n * 1e3))
