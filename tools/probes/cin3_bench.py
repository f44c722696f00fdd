"""conv_cin3_kernel (VGG conv1_1 of the fp32 path: 3 -> 64, 3x3, 1024^2, batch 8) with and without the fused ContentLoss sum; optional library path (A/B)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
if len(sys.argv) > 1:
    lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    for name, (res, args) in _lib._SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    _lib._lib = lib
b, res = 8, 1024
fc = conv.FrozenConv2d(torch.randn(64, 3, 3, 3) / 27 ** 0.5, 1, 1, device='cuda')
x = torch.randn(b, 3, res, res, device='cuda')
bias = torch.randn(64, device='cuda')
y = torch.empty(b, 64, res, res, device='cuda')
ref = torch.randn(b, 64, res, res, device='cuda')
for with_sq in (False, True):
    run = lambda: fc.forward(x, out=y, bias=bias, sq=(ref, torch.zeros(_lib.SQ_SLOTS, device='cuda'), [False]) if with_sq else None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ms = float(np.median(ts))
    by = x.numel() * 4 + y.numel() * 4 * (2 if with_sq else 1)
    print('cin3 3->64 @1024 b8 %s %.4f ms  %.0f GB/s' % ('+sq' if with_sq else '   ', ms, by / ms / 1e6), flush=True)
