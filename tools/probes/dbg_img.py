"""debug aid: one l2i_conv_img_h8 launch against float64 torch (which outputs differ, and how)."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from latent2im_amd import conv

conv.PRECISION = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
cin, cout, k, stride, pad, h, w, b = [int(v) for v in (sys.argv[2:10] if len(sys.argv) > 9 else (3, 32, 1, 1, 0, 40, 64, 2))]
rs = np.random.RandomState(0)
wt = torch.from_numpy(rs.randn(cout, cin, k, k).astype(np.float32) / np.sqrt(cin * k * k))
x = torch.from_numpy(rs.randn(b, cin, h, w).astype(np.float32))
ic = conv.ImgConvH8(wt, stride, pad, device='cuda')
got = conv.from_h8(ic.forward(x.cuda()), cout).double().cpu()
dt = conv.h8_dtype()
ref = F.conv2d(x.to(dt).double(), wt.to(dt).double(), stride=stride, padding=pad)
bad = ~torch.isfinite(got) | ((got - ref).abs() > 2.0 ** -7 * ref.abs() + 1e-2)
print('planes', tuple(ic.planes.shape), 'bad', int(bad.sum()), 'of', bad.numel(), 'nan', int((~torch.isfinite(got)).sum()))
idx = bad.nonzero()
print('bad channels', sorted(set(idx[:, 1].tolist()))[:40])
print('bad rows', sorted(set(idx[:, 2].tolist()))[:40])
print('bad cols', sorted(set(idx[:, 3].tolist()))[:70])
for i in idx[:8].tolist():
    print(i, float(got[tuple(i)]), float(ref[tuple(i)]))
