"""The fp32 Winograd kernel (its input transform uses v_pk_add_f32 with op_sel butterflies) repeated on stream B while conv_h8_kernel runs on
stream A: is it a victim of the packed-fp32 effect?  (No configuration of the training step runs the two side by side; this is a robustness check.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = 'cuda'
torch.manual_seed(0)
w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
hc = conv.H8Conv(w, 1, 1, device=dev)
xa = torch.randn(8, 16, 64, 64, 8, device=dev).to(torch.bfloat16); ya = torch.empty_like(xa); ba = torch.randn(128, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for cin, cout, res, b in ((64, 64, 128, 4), (128, 128, 64, 4), (256, 256, 32, 8)):
    wf = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(wf, 1, 1, device=dev)
    x = torch.randn(b, cin, res, res, device=dev); bias = torch.randn(cout, device=dev); sc = torch.rand(b, cin, device=dev) + 0.5
    for kind, kw in (('plain', dict(bias=bias)), ('style', dict(in_scale=sc, bias=bias))):
        for beside in (True, False):
            sums = torch.zeros(reps, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            for i in range(reps):
                if beside:
                    with torch.cuda.stream(sa):
                        for _ in range(4):
                            hc.forward(xa, out=ya, bias=ba, act=conv.ACT_RELU)
                with torch.cuda.stream(sb):
                    sums[i] = fc.forward(x, **kw).double().abs().sum()
            torch.cuda.synchronize()
            vals, counts = np.unique(sums.cpu().numpy(), return_counts=True)
            print('winograd %d->%d @%d %-5s %-15s %3d distinct checksums in %d' % (cin, cout, res, kind, 'beside conv_h8' if beside else 'alone', len(vals), reps), flush=True)
