"""l2i_conv2d_h8 / l2i_conv_transpose2d_h8 on the step's conv shapes (batch 8): ms, TFLOP/s of the dense correlation, GB/s of the algorithmic bytes.
usage: python tools/probes/h8_bench.py [lib.so]   (another build of the library, e.g. a timing ablation from tools/probes/h8_ablate.sh); H8_ONLY=k3s1 limits the cases"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
if len(sys.argv) > 1:
    lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib._lib = lib
only = os.environ.get('H8_ONLY', '')
b = 8
CASES = [(64, 64, 3, 1, 1, False, 1024), (128, 128, 3, 1, 1, False, 512), (64, 128, 3, 1, 1, False, 512), (32, 32, 3, 1, 1, False, 1024), (256, 256, 3, 1, 1, False, 128),
         (512, 512, 3, 1, 1, False, 64), (512, 512, 3, 1, 1, False, 32), (256, 1024, 1, 1, 0, False, 64), (1024, 256, 1, 1, 0, False, 64), (64, 256, 1, 1, 0, False, 256),
         (512, 2048, 1, 1, 0, False, 32), (32, 64, 3, 2, 0, False, 1028), (256, 512, 3, 2, 0, False, 132), (128, 128, 3, 2, 1, False, 256), (256, 512, 1, 2, 0, False, 256),
         (64, 32, 3, 2, 0, True, 512), (512, 256, 3, 2, 0, True, 64), (128, 64, 3, 2, 0, True, 256)]
for cin, cout, k, s, pad, tr, res in CASES:
    if only and only != 'k%ds%d%s' % (k, s, 't' if tr else ''):
        continue
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    hc = conv.H8Conv(w, s, pad, transposed=tr, device='cuda')
    x = torch.randn(b, cin // 8, res, res, 8, device='cuda').to(torch.bfloat16)
    oh, ow = hc.out_hw(res, res)
    y = torch.empty(b, cout // 8, oh, ow, 8, device='cuda', dtype=torch.bfloat16)
    bias = torch.randn(cout, device='cuda')
    epi = os.environ.get('H8_EPI', 'b')                 # epilogue of the launch: b | br | o | rmo | ros (bias, residual, out_mask, res_mask, res_sub)
    kw = dict(bias=bias, act=conv.ACT_RELU)
    if epi != 'b':
        mk = lambda: torch.randn_like(y, dtype=torch.float32).to(torch.bfloat16)
        kw = {}
        if 'b' in epi: kw.update(bias=bias, act=conv.ACT_RELU)
        if 'r' in epi: kw.update(residual=mk())
        if 'o' in epi: kw.update(out_mask=mk(), mask=(1.0, 0.0))
        if 'm' in epi: kw.update(res_mask=mk())
        if 's' in epi: kw.update(res_sub=mk(), res_coef=0.5)
    for _ in range(3):
        hc.forward(x, out=y, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            hc.forward(x, out=y, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ms = float(np.median(ts))
    fl = 2.0 * b * cout * cin * k * k * (res * res if tr else oh * ow)
    by = 2.0 * (x.numel() + y.numel() * (1 + sum(c in epi for c in 'roms')))
    print('%4d->%-4d k%d s%d %s @%-4d  %.4f ms  %.0f TFLOP/s  %.0f GB/s' % (cin, cout, k, s, 'T' if tr else ' ', res, ms, fl / ms / 1e9, by / ms / 1e6), flush=True)
    del x, y, hc
    torch.cuda.empty_cache()
