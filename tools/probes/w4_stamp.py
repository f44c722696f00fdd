"""[r6] Where a block of the position-split F(4x4,3x3) kernel spends its life: a build with -DL2I_W4_STAMP (tools/ab/libl2i_w4_stamp.so) writes the cycle counter of
wave 0 at six points of every block into the launch's (otherwise unused) `ws` buffer: entry | U(0), raw(0) landed | first transform done | K loop done |
halves exchanged | block done.  usage: L2I_LIB=tools/ab/libl2i_w4_stamp.so python tools/probes/w4_stamp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
B, DEV = 8, 'cuda'
lib = _lib.load()
orig = lib.l2i_conv2d_wino4_f32
STAMPS = {}
def wrapped(p, st):
    p.ws = STAMPS['ptr']
    return orig(p, st)
lib.l2i_conv2d_wino4_f32 = wrapped
rs = np.random.RandomState(0)
for cin, cout, res, kind in [(64, 64, 1024, 'plain'), (64, 64, 1024, 'G'), (32, 32, 1024, 'G'), (64, 64, 512, 'plain'), (128, 128, 256, 'plain'), (512, 512, 64, 'plain'), (64, 64, 256, 'plain')]:
    wt = torch.tensor(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9), dtype=torch.float32)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    x = torch.randn(B, cin, res, res, device=DEV)
    y = torch.empty(B, cout, res, res, device=DEV)
    kw = {}
    if kind == 'G':
        kw = dict(in_scale=torch.rand(B, cin, device=DEV) + 0.5, out_scale=torch.rand(B, cout, device=DEV) + 0.5, noise=torch.randn(B, 1, res, res, device=DEV), noise_w=0.05,
                  bias=torch.randn(cout, device=DEV), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    nblk = B * (res // 64) * (res // 8) * ((cout + 31) // 32)
    buf = torch.zeros((nblk + 8) * 8, dtype=torch.int64, device=DEV)
    STAMPS['ptr'] = buf.data_ptr()
    for _ in range(3):
        fc.forward(x, out=y, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fc.forward(x, out=y, **kw); e1.record(); torch.cuda.synchronize()
    t = buf.view(-1, 8)[:nblk, :6].cpu().numpy().astype(np.float64)
    d = np.diff(t, axis=1)
    life = t[:, 5] - t[:, 0]
    span = t[:, 5].max() - t[:, 0].min()
    names = ['entry->landed', 'first xf', 'K loop', 'drain+exchange', 'epilogue']
    print('%d->%d @%d %s: %d blocks, launch %.3f ms, span %.0f ticks; block life mean %.0f ticks' % (cin, cout, res, kind, nblk, e0.elapsed_time(e1), span, life.mean()))
    print('   ' + '  '.join('%s %.0f (%.0f%%)' % (n, d[:, i].mean(), 100 * d[:, i].mean() / life.mean()) for i, n in enumerate(names)))
    print('   ticks per ms: %.0f;  chunks %d -> K loop per chunk %.0f ticks' % (span / e0.elapsed_time(e1), cin // 4, d[:, 2].mean() / (cin // 4)))
