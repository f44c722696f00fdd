"""[r6] l2i_conv2d_h8 on the MID-SIZE launches of the c5 step (ResNet-50 at 1024^2 input, batch 8: <= 2048 blocks, where the trace shows 35 - 80 us per
launch whatever the work) — per shape: us, TFLOP/s, GB/s of the algorithmic bytes, blocks.  Rotates over NBUF input / output buffers so that a launch
does not find its own previous output in L2.  usage: python tools/probes/h8_mid_bench.py [f16|bf16]; env L2I_H8_KS / L2I_H8_SPLITK ... select variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, _lib
if os.environ.get('H8_LIB'):                      # another build of the library (a timing ablation from tools/probes/h8_ablate.sh)
    import ctypes
    lib = ctypes.CDLL(os.path.abspath(os.environ['H8_LIB']))
    for name, (res_, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res_, args
    _lib._lib = lib
conv.PRECISION = sys.argv[1] if len(sys.argv) > 1 else 'f16'
DT = conv.h8_dtype()
b = 8
NBUF = 4
# cin, cout, k, stride, pad, res, epilogue
CASES = [(64, 64, 3, 1, 1, 256, 'b'), (64, 64, 3, 1, 1, 256, 'o'), (128, 128, 3, 1, 1, 128, 'b'), (128, 128, 3, 1, 1, 128, 'o'), (256, 256, 3, 1, 1, 64, 'b'), (256, 256, 3, 1, 1, 64, 'o'),
         (512, 512, 3, 1, 1, 32, 'b'), (512, 512, 3, 1, 1, 32, 'o'), (512, 512, 3, 1, 1, 16, 'b'), (512, 512, 3, 1, 1, 8, 'b'), (512, 512, 3, 1, 1, 4, 'b'),
         (256, 64, 1, 1, 0, 256, 'b'), (64, 256, 1, 1, 0, 256, 'br'), (256, 64, 1, 1, 0, 256, 'o'), (64, 256, 1, 1, 0, 256, 'rom'),
         (512, 128, 1, 1, 0, 128, 'b'), (128, 512, 1, 1, 0, 128, 'br'), (128, 512, 1, 1, 0, 128, 'rom'),
         (1024, 256, 1, 1, 0, 64, 'b'), (256, 1024, 1, 1, 0, 64, 'br'), (256, 1024, 1, 1, 0, 64, 'rom'), (1024, 256, 1, 1, 0, 64, 'o'),
         (2048, 512, 1, 1, 0, 32, 'b'), (512, 2048, 1, 1, 0, 32, 'br'), (512, 2048, 1, 1, 0, 32, 'rom'), (2048, 512, 1, 1, 0, 32, 'o')]
only = os.environ.get('H8_ONLY', '')
tot = 0.0
for cin, cout, k, s, pad, res, epi in CASES:
    if only and only != 'k%d' % k:
        continue
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    hc = conv.H8Conv(w, s, pad, device='cuda')
    xs = [torch.randn(b, cin // 8, res, res, 8, device='cuda').to(DT) for _ in range(NBUF)]
    oh, ow = hc.out_hw(res, res)
    ys = [torch.empty(b, cout // 8, oh, ow, 8, device='cuda', dtype=DT) for _ in range(NBUF)]
    bias = torch.randn(cout, device='cuda')
    mk = lambda: torch.randn(b, cout // 8, oh, ow, 8, device='cuda').to(DT)
    kw = {}
    if 'b' in epi: kw.update(bias=bias, act=conv.ACT_RELU)
    if 'r' in epi: kw.update(residual=mk())
    if 'o' in epi: kw.update(out_mask=mk(), mask=(1.0, 0.0))
    if 'm' in epi: kw.update(res_mask=kw['out_mask'])
    for i in range(NBUF):
        hc.forward(xs[i], out=ys[i], **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2 * NBUF):
            hc.forward(xs[i % NBUF], out=ys[i % NBUF], **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (2 * NBUF))
    ms = float(np.median(ts))
    fl = 2.0 * b * cout * cin * k * k * oh * ow
    by = 2.0 * (xs[0].numel() + ys[0].numel() * (1 + ('r' in epi) + ('o' in epi)))
    blocks = b * ((ow + 31) // 32) * ((oh + 7) // 8) * ((cout + 63) // 64)
    tot += ms
    print('%4d->%-4d k%d @%-4d %-3s %6d blocks  %7.1f us  %6.0f TFLOP/s  %5.0f GB/s' % (cin, cout, k, res, epi, blocks, ms * 1e3, fl / ms / 1e9, by / ms / 1e6), flush=True)
    del xs, ys, hc
    torch.cuda.empty_cache()
print('sum %.1f us' % (tot * 1e3))
