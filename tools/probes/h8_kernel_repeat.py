"""Each 16-bit kernel repeated on identical inputs: distinct outputs per kernel.  Run two at once: a kernel whose result depends on what else
is on the GPU reads something it did not write (or races with itself)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
from latent2im_amd import kernels16 as K16
BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = 4
dev = 'cuda'
torch.manual_seed(0)
h = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:6]
def h8(c, hh, ww): return (torch.randn(b, c // 8, hh, ww, 8, device=dev) * 0.7).to(BF)
cases = {}
for cin, cout, k, s, pad, tr, res in ((512, 512, 3, 1, 1, False, 4), (512, 512, 3, 1, 1, False, 8), (512, 512, 3, 1, 1, False, 16), (256, 256, 3, 1, 1, False, 32), (128, 128, 3, 1, 1, False, 64),
                                      (512, 512, 3, 2, 0, True, 4), (512, 512, 3, 2, 0, True, 8), (512, 256, 3, 2, 0, True, 16), (256, 128, 3, 2, 0, True, 32),
                                      (64, 256, 1, 1, 0, False, 16), (512, 512, 3, 2, 0, False, 17)):
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    hc = conv.H8Conv(w, s, pad, transposed=tr, device=dev)
    x = h8(cin, res, res)
    sc = torch.rand(b, cout, device=dev) + 0.5
    bias = torch.randn(cout, device=dev)
    if k == 3 and not (s == 2 and not tr):
        planes = K16.modulate_planes(conv.pack_weight_h8_f32(w.transpose(0, 1).contiguous() if False else w).to(dev), torch.rand(b, cin, device=dev) + 0.5) if not tr else None
    else:
        planes = None
    def run(hc=hc, x=x, sc=sc, bias=bias, planes=planes, tr=tr):
        if planes is not None:
            return hc.forward(x, planes=planes, w_bstride=planes[0].numel() * 2, out_scale=sc, bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5)
        return hc.forward(x, out_scale=sc, bias=None if tr else bias)
    cases['conv %d->%d k%d s%d %s@%d%s' % (cin, cout, k, s, 'T' if tr else ' ', res, ' planes' if planes is not None else '')] = run
kk = torch.tensor([1., 3., 3., 1.]); k2 = (kk[:, None] * kk[None, :]); k2 = (k2 / k2.sum() * 4).to(dev); sep = K16.separable(k2)
for c, r in ((512, 8), (256, 32), (64, 128)):
    x = h8(c, 2 * r + 1, 2 * r + 1); bias = torch.randn(c, device=dev)
    cases['fir sep %dch @%d' % (c, 2 * r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5, sep=sep)
    cases['fir gen %dch @%d' % (c, 2 * r)] = lambda x=x, bias=bias: K16.upfirdn2d(x, k2, pad=(1, 1, 1, 1), bias=bias, act=conv.ACT_LRELU, gain=2 ** 0.5)
    y = h8(c, r, r); wm = torch.randn(b, 3, c, device=dev)
    cases['torgb %dch @%d' % (c, r)] = lambda y=y, wm=wm: K16.torgb_fwd(y, wm, torch.zeros(3, device=dev))
seen = {k: {} for k in cases}
for i in range(reps):
    for k, f in cases.items():
        v = h(f())
        seen[k][v] = seen[k].get(v, 0) + 1
for k in cases:
    print('%-40s %3d distinct in %d  %s' % (k, len(seen[k]), reps, sorted(seen[k].values(), reverse=True)[:4]))
