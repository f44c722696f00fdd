"""debugging aid: where the 3x3 stride-2 DMA kernel differs from torch on one shape (error rows / columns / channels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from latent2im_amd import conv
cin, cout, h, w, pad, b = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (32, 32, 67, 67, 0, 1)
rs = np.random.RandomState(1)
wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
x = torch.tensor(rs.randn(b, cin, h, w), dtype=torch.float32)
fc = conv.FrozenConv2d(wt, 2, pad, device='cuda')
ref = F.conv2d(x.double(), wt.double(), stride=2, padding=pad)
for hint in (0, 2):
    y = fc.forward(x.cuda(), tile_hint=hint).double().cpu()
    e = (y - ref).abs()
    bad = (e > 1e-4 * ref.abs().max())
    print('hint', hint, 'max err', float(e.max()), 'bad', int(bad.sum()), 'of', bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print(' bad samples', sorted(set(idx[:, 0].tolist())), 'channels', sorted(set(idx[:, 1].tolist()))[:40])
        print(' bad rows', sorted(set(idx[:, 2].tolist())))
        print(' bad cols', sorted(set(idx[:, 3].tolist())))
