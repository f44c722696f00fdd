"""Micro-benchmark of l2i_conv2d_f32 on the layer shapes of the 1024^2 walk-training step (GPU box only).
Prints TFLOP/s (2*MAC of the dense contraction) per shape and tile configuration."""
import sys
import os
import json
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from latent2im_amd import conv, _lib

if os.environ.get('L2I_LIB_PATH'):          # A/B runs of two builds in one session
    _lib.LIB_PATH = os.environ['L2I_LIB_PATH']

SHAPES = [  # name, cin, cout, k, stride, pad, transposed, res, batch
    ('g64_512', 512, 512, 3, 1, 1, False, 64, 8),
    ('g128_256', 256, 256, 3, 1, 1, False, 128, 8),
    ('g256_128', 128, 128, 3, 1, 1, False, 256, 8),
    ('g512_64', 64, 64, 3, 1, 1, False, 512, 8),
    ('g1024_32', 32, 32, 3, 1, 1, False, 1024, 8),
    ('g_up512', 128, 64, 3, 2, 0, True, 256, 8),
    ('g16_512', 512, 512, 3, 1, 1, False, 16, 8),
    ('g4_512', 512, 512, 3, 1, 1, False, 4, 8),
    ('r_1x1_256', 256, 64, 1, 1, 0, False, 256, 8),
    ('r_1x1_2048', 1024, 2048, 1, 1, 0, False, 32, 8),
    ('r_stem', 3, 64, 7, 2, 3, False, 1024, 8),
    ('v_64', 64, 64, 3, 1, 1, False, 1024, 4),
    ('g_up128', 512, 256, 3, 2, 0, True, 64, 8),
    ('g_up128_p1', 512, 256, 3, 2, 1, True, 64, 8),
    ('g_up64', 512, 512, 3, 2, 0, True, 32, 8),
    ('g_up64_p1', 512, 512, 3, 2, 1, True, 32, 8),
    ('g_up512_p1', 128, 64, 3, 2, 1, True, 256, 8),
    ('g_up1024', 64, 32, 3, 2, 0, True, 512, 8),
    ('g_up256', 256, 128, 3, 2, 0, True, 128, 8),
    ('r_256_1024', 256, 1024, 1, 1, 0, False, 64, 8),
    ('r_1024_256', 1024, 256, 1, 1, 0, False, 64, 8),
    ('r_2048_512', 2048, 512, 1, 1, 0, False, 32, 8),
    ('r_512_128', 512, 128, 1, 1, 0, False, 128, 8),
    ('r_64_256', 64, 256, 1, 1, 0, False, 256, 8),
    ('r_3x3s2_128', 128, 128, 3, 2, 1, False, 256, 8),
    ('r_3x3s2_256', 256, 256, 3, 2, 1, False, 128, 8),
    ('d_3x3s2_64', 64, 128, 3, 2, 0, False, 513, 8),
    ('d_3x3s2_256', 256, 512, 3, 2, 0, False, 129, 8),
    ('r_1x1s2_256', 256, 512, 1, 2, 0, False, 256, 8),
    ('v_128', 128, 128, 3, 1, 1, False, 512, 8),
    ('r_3x3_64', 64, 64, 3, 1, 1, False, 256, 8),
    ('r_3x3_128', 128, 128, 3, 1, 1, False, 128, 8),
    ('r_3x3_512', 512, 512, 3, 1, 1, False, 32, 8),
]


def main():
    hints = [int(h) for h in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 3]
    if len(sys.argv) > 2:
        conv.PRECISION = sys.argv[2]
    if len(sys.argv) > 3:
        conv.USE_WINOGRAD = sys.argv[3] != '0'
    rows = []
    only = os.environ.get('L2I_BENCH_ONLY', '')
    for name, cin, cout, k, stride, pad, tr, res, b in SHAPES:
        if only and only not in name:
            continue
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        fc = conv.FrozenConv2d(w, stride, pad, transposed=tr, device='cuda')
        x = torch.randn(b, cin, res, res, device='cuda')
        oh, ow = fc.out_hw(res, res)
        y = torch.empty(b, cout, oh, ow, device='cuda')
        macs = b * cout * cin * k * k * (res * res if tr else oh * ow)
        for hint in hints:
            if hint and [4, 2, 1, 2, 1, 1, 4, 2][hint - 1] * 32 > (cout + 31) // 32 * 32:
                continue
            try:
                for _ in range(2):
                    fc.forward(x, out=y, tile_hint=hint)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 5
                e0.record()
                for _ in range(n):
                    fc.forward(x, out=y, tile_hint=hint)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                tf = 2 * macs / ms / 1e9
                rows.append(dict(shape=name, hint=hint, ms=round(ms, 4), tflops=round(tf, 2)))
                print('%-12s hint=%d  %8.3f ms  %7.2f TFLOP/s' % (name, hint, ms, tf), flush=True)
            except Exception as e:
                print(name, hint, 'ERR', e, flush=True)
    print(json.dumps(rows))


if __name__ == '__main__':
    main()
