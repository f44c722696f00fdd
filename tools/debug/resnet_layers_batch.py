"""Debug aid: every conv shape of ResNet-50 on small maps at batch 8 / 16 against torch CPU (checker)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from latent2im_amd import conv
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
torch.set_num_threads(32)
shapes = []
hw = 8
cin = 64
for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
    shapes += [(cin, planes, 1, 1, 0, hw), (planes, planes, 3, stride, 1, hw), (planes, planes * 4, 1, 1, 0, hw // stride),
               (cin, planes * 4, 1, stride, 0, hw), (planes * 4, planes, 1, 1, 0, hw // stride), (planes, planes, 3, 1, 1, hw // stride)]
    cin = planes * 4
    hw //= stride
shapes.append((3, 64, 7, 2, 3, 32))
for B in (4, 8, 16):
    for (ci, co, k, s, pad, h) in shapes:
        rs = np.random.RandomState(ci + co + k + h)
        wt = T(rs.randn(co, ci, k, k) / np.sqrt(ci * k * k))
        x = T(rs.randn(B, ci, h, h)).requires_grad_(True)
        bias = T(rs.randn(co))
        ref = torch.relu(F.conv2d(x, wt, stride=s, padding=pad) + bias[None, :, None, None])
        fc = conv.FrozenConv2d(wt, s, pad, device='cuda')
        y = fc.forward(x.detach().cuda(), bias=bias.cuda(), act=conv.ACT_RELU)
        ef = float((y.cpu() - ref).abs().max() / ref.abs().max())
        gy = T(rs.randn(*ref.shape))
        pre = F.conv2d(x, wt, stride=s, padding=pad)
        gref, = torch.autograd.grad(pre, x, gy)
        gx = fc.dgrad(gy.cuda(), (h, h))
        eb = float((gx.cpu() - gref).abs().max() / gref.abs().max())
        flag = '  <<<<<' if max(ef, eb) > 1e-4 else ''
        print('B %2d  %4d->%4d k%d s%d %2dx%2d  fwd %.1e  dgrad %.1e%s' % (B, ci, co, k, s, h, h, ef, eb, flag), flush=True)
