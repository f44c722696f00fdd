"""python tools/probes/torgb_bench.py: ToRGB forward (fp32 NCHW and bf16 h8) on the step's top-resolution shapes, batch 8: ms and TB/s of the bytes it must move."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import kernels as K, kernels16 as K16, conv
B = 8
for c, r in ((32, 1024), (64, 512), (128, 256), (256, 128), (512, 64)):
    x = torch.randn(B, c, r, r, device='cuda')
    wm = torch.randn(B, 3, c, device='cuda')
    bias = torch.zeros(3, device='cuda')
    xh = conv.to_h8(x)
    for name, f, nbytes in (('f32', lambda: K.torgb_fwd(x, wm, bias), x.numel() * 4 + B * 3 * r * r * 4), ('h8 ', lambda: K16.torgb_fwd(xh, wm, bias), x.numel() * 2 + B * 3 * r * r * 4)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('torgb %s %4d ch @%4d: %.4f ms  %.2f TB/s' % (name, c, r, ms, nbytes / ms / 1e9))
