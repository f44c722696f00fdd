"""Interleaved A/B timing of two builds of libl2i_hip.so on the step's heavy conv launches, in ONE process (guide rule: perf deltas come
from within-probe interleaved rounds).  usage: python tools/probes/conv_ab.py <libA.so> <libB.so> [family ...]

Cases are the launches of the c3 step (profiles/r03_c3_launches.json shape tuples): Winograd 3x3 (plain / relu_in / style scale + noise
epilogue / gradient mask / residual epilogue), stride-2 3x3, stride-2 transposed 3x3, strided 1x1."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from latent2im_amd import _lib, conv


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return lib


# (family, cin, cout, k, stride, pad, transposed, res, batch, kind)
CASES = [
    ('wino', 64, 64, 3, 1, 1, False, 1024, 8, 'relu_in'), ('wino', 128, 128, 3, 1, 1, False, 512, 8, 'relu_in'), ('wino', 64, 128, 3, 1, 1, False, 512, 8, 'relu_in'),
    ('wino', 64, 64, 3, 1, 1, False, 1024, 8, 'res'), ('wino', 128, 128, 3, 1, 1, False, 512, 8, 'res'),
    ('wino', 32, 32, 3, 1, 1, False, 1024, 8, 'style'), ('wino', 64, 64, 3, 1, 1, False, 512, 8, 'style'), ('wino', 128, 128, 3, 1, 1, False, 256, 8, 'style'),
    ('wino', 256, 256, 3, 1, 1, False, 128, 8, 'style'), ('wino', 512, 512, 3, 1, 1, False, 64, 8, 'style'), ('wino', 512, 512, 3, 1, 1, False, 32, 8, 'plain'),
    ('wino', 256, 256, 3, 1, 1, False, 64, 8, 'plain'), ('wino', 64, 64, 3, 1, 1, False, 256, 8, 'plain'), ('wino', 256, 256, 3, 1, 1, False, 64, 8, 'mask'),
    ('wino', 64, 64, 3, 1, 1, False, 512, 8, 'mask'),
    ('s2', 32, 64, 3, 2, 0, False, 1028, 8, 'plain'), ('s2', 64, 128, 3, 2, 0, False, 516, 8, 'plain'), ('s2', 128, 256, 3, 2, 0, False, 260, 8, 'plain'),
    ('s2', 256, 512, 3, 2, 0, False, 132, 8, 'plain'), ('s2', 512, 512, 3, 2, 1, False, 64, 8, 'plain'), ('s2', 128, 128, 3, 2, 1, False, 256, 8, 'plain'),
    ('s2', 256, 512, 1, 2, 0, False, 256, 8, 'plain'), ('s2', 1024, 2048, 1, 2, 0, False, 64, 8, 'plain'),
    ('tr', 64, 32, 3, 2, 0, True, 512, 8, 'style'), ('tr', 128, 64, 3, 2, 0, True, 256, 8, 'style'), ('tr', 256, 128, 3, 2, 0, True, 128, 8, 'style'),
    ('tr', 512, 256, 3, 2, 0, True, 64, 8, 'style'), ('tr', 512, 512, 3, 2, 0, True, 32, 8, 'style'), ('tr', 64, 32, 3, 2, 0, True, 512, 8, 'mask'),
    ('tr', 512, 256, 3, 2, 0, True, 64, 8, 'mask'),
]


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    fams = set(sys.argv[3:])
    libs = {'A': load(pa), 'B': load(pb)}
    rounds = 7
    print('A = %s\nB = %s' % (pa, pb))
    for fam, cin, cout, k, stride, pad, tr, res, b, kind in CASES:
        if fams and fam not in fams:
            continue
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        fc = conv.FrozenConv2d(w, stride, pad, transposed=tr, device='cuda')
        x = torch.randn(b, cin, res, res, device='cuda')
        oh, ow = fc.out_hw(res, res)
        if tr:                                      # the generator's up layers: output padded to whole 16-byte rows
            oh, ow = oh + 3, ow + 3
        y = torch.empty(b, cout, oh, ow, device='cuda')
        kw = {}
        if kind == 'relu_in':
            kw = dict(in_mask=x, mask=(1.0, 0.0), bias=torch.randn(cout, device='cuda'))
        elif kind == 'res':
            r = torch.randn_like(y)
            kw = dict(residual=r, out_mask=r, res_sub=torch.randn_like(y), res_coef=0.5)
        elif kind == 'style':
            kw = dict(in_scale=torch.rand(b, cin, device='cuda') + 0.5, out_scale=torch.rand(b, cout, device='cuda') + 0.5)
            if not tr:
                kw.update(noise=torch.randn(b, 1, oh, ow, device='cuda'), noise_w=0.1, bias=torch.randn(cout, device='cuda'), act=conv.ACT_LRELU, gain=2 ** 0.5)
        elif kind == 'mask':
            kw = dict(in_mask=torch.randn_like(x), mask=(1.0, 0.0) if fam == 'wino' else (1.41, 0.28))
        elif kind == 'plain':
            kw = dict(bias=torch.randn(cout, device='cuda')) if not tr else {}
        times = {'A': [], 'B': []}
        outs = {}
        for rd in range(rounds + 1):
            for name in ('A', 'B') if rd % 2 == 0 else ('B', 'A'):
                _lib._lib = libs[name]
                fc.forward(x, out=y, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fc.forward(x, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                if rd > 0:
                    times[name].append(e0.elapsed_time(e1) / 3)
                else:
                    outs[name] = y.clone()
        diff = float((outs['A'] - outs['B']).abs().max() / outs['A'].abs().max())
        ma, mb = float(np.median(times['A'])), float(np.median(times['B']))
        fl = 2.0 * b * cout * cin * k * k * (res * res if tr else (oh * ow))
        print('%-5s %4d->%-4d k%d s%d @%-4d %-8s A %.4f ms (%.0f TF)  B %.4f ms (%.0f TF)  B/A %.3f  min %.4f/%.4f  max|A-B|/max %.1e'
              % (fam, cin, cout, k, stride, res, kind, ma, fl / ma / 1e9, mb, fl / mb / 1e9, mb / ma, min(times['A']), min(times['B']), diff), flush=True)
        del x, y, fc, kw
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
