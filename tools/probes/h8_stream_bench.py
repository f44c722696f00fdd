"""The streaming kernels of the 16-bit path (l2i_stream_h8.hip) on the c5 step's large shapes (batch 8): ms and TB/s of the algorithmic bytes.
usage: python tools/probes/h8_stream_bench.py [libA.so libB.so]   (two libraries: interleaved A/B with result comparison)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from latent2im_amd import _lib, conv
from latent2im_amd import kernels16 as K16

BF = torch.bfloat16
b = 8


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return lib


def h8(c, h, w):
    return (torch.randn(b, c // 8, h, w, 8, device='cuda') * 0.7).to(BF)


def cases():
    k = torch.tensor([1., 3., 3., 1.])
    k2 = (k[:, None] * k[None, :])
    k2 = (k2 / k2.sum() * 4).cuda()
    sep = K16.separable(k2)
    for c, r in ((32, 1024), (64, 512), (128, 256), (256, 128)):
        x = h8(c, r + 4, r + 4)
        nz = torch.randn(b, 1, r, r, device='cuda')
        bias = torch.randn(c, device='cuda')
        yield ('fir up-layer blur+epi %3dch @%d' % (c, r), lambda x=x, nz=nz, bias=bias: K16.upfirdn2d(x, k2, pad=(1, -2, 1, -2), noise=nz, noise_w=0.1, bias=bias, act=conv.ACT_LRELU,
                                                                                                     gain=2 ** 0.5, sep=sep), 2 * b * c * ((r + 4) ** 2 + r * r) + 4 * b * r * r)
        g = h8(c, r, r)
        yield ('fir blur bwd plain    %3dch @%d' % (c, r), lambda g=g: K16.upfirdn2d(g, k2, pad=(2, 5, 2, 5), sep=sep), 2 * b * c * (r * r + (r + 4) ** 2))
    for c, r in ((32, 1024), (64, 512), (128, 256)):
        x = h8(c, r + 1, r + 1)
        yield ('fir D blur (2,2)      %3dch @%d' % (c, r), lambda x=x: K16.upfirdn2d(x[:, :, :r, :r].contiguous(), k2 / 4, pad=(2, 2, 2, 2), sep=K16.separable(k2 / 4)), 2 * b * c * (r * r + (r + 1) ** 2))
    for c, r in ((32, 1024), (64, 512), (128, 256), (512, 64)):
        y, gin = h8(c, r, r), h8(c, r, r)
        grgb = torch.randn(b, 3, r, r, device='cuda')
        wm = torch.randn(b, 3, c, device='cuda')
        sc = torch.rand(b, c, device='cuda')
        bias = torch.randn(c, device='cuda')
        nz = torch.randn(b, 1, r, r, device='cuda')

        def act(y=y, gin=gin, grgb=grgb, wm=wm, sc=sc, bias=bias, nz=nz, c=c):
            red, red_rgb = torch.zeros(b, c, device='cuda'), torch.zeros(b, c, 3, device='cuda')
            return K16.sg2_act_bwd(y, gin, sc, grgb, wm, bias, nz, 0.1, 0.2, 2 ** 0.5, red, red_rgb), red, red_rgb
        yield ('sg2_act_bwd full      %3dch @%d' % (c, r), act, 2 * b * c * r * r * 3 + 4 * b * r * r * 4)

        def act2(y=y, gin=gin, sc=sc, bias=bias, nz=nz, c=c):
            red = torch.zeros(b, c, device='cuda')
            return K16.sg2_act_bwd(y, gin, sc, None, None, bias, nz, 0.1, 0.2, 2 ** 0.5, red), red
        yield ('sg2_act_bwd no-rgb    %3dch @%d' % (c, r), act2, 2 * b * c * r * r * 3 + 4 * b * r * r)
        yield ('dot_reduce a*b        %3dch @%d' % (c, r), lambda y=y, gin=gin: K16.dot_reduce(y, gin), 2 * b * c * r * r * 2)
        yield ('torgb_fwd             %3dch @%d' % (c, r), lambda y=y, wm=wm: K16.torgb_fwd(y, wm, torch.zeros(3, device='cuda')), 2 * b * c * r * r + 4 * b * 3 * r * r)


def flat(o):
    if torch.is_tensor(o):
        return [o]
    out = []
    for t in o:
        out += flat(t)
    return out


def main():
    libs = {'A': None}
    if len(sys.argv) >= 3:
        libs = {'A': load(sys.argv[1]), 'B': load(sys.argv[2])}
        print('A = %s\nB = %s' % (sys.argv[1], sys.argv[2]))
    _lib.load()
    for name, fn, nbytes in cases():
        times = {k: [] for k in libs}
        outs = {}
        for rd in range(6):
            order = list(libs) if rd % 2 == 0 else list(libs)[::-1]
            for key in order:
                if libs[key] is not None:
                    _lib._lib = libs[key]
                o = fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rd > 0:
                    times[key].append(e0.elapsed_time(e1) / 3)
                else:
                    outs[key] = [t.float().clone() for t in flat(o)]
        msg = '%-36s' % name
        for key in libs:
            ms = float(np.median(times[key]))
            msg += '  %s %.4f ms %5.2f TB/s' % (key, ms, nbytes / ms / 1e9)
        if len(libs) == 2:
            d = max(float((x - y).abs().max() / (x.abs().max() + 1e-30)) for x, y in zip(outs['A'], outs['B']))
            msg += '  B/A %.3f  max|A-B|/max %.1e' % (float(np.median(times['B'])) / float(np.median(times['A'])), d)
        print(msg, flush=True)
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
