"""Replays the conv launches of one config-3 training step shape by shape (GPU box only) and prints the time of each beside the time
recorded in tools/c3_launch_shapes.json (bench.py --dump_launches of the build that file was taken from).

    python tools/bench_layers.py [substring of the family name] [min ms_per_step]
    L2I_LIB_PATH=/path/to/other/libl2i_hip.so  python tools/bench_layers.py gemm1x1      # A/B of two builds

Shape tuples are the ones conv.run_launch / run_fused_transposed put into conv.PROFILE:
    (B, cin, cout, kh, kw, stride, H, W, OH, OW, step, in_mask, in_scale, epilogue operands + act)
    epilogue letters: d out_scale, n noise, b bias, r residual, m res_mask, o out_mask, a accumulate, s res_sub
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from latent2im_amd import _lib, conv

if os.environ.get('L2I_LIB_PATH'):
    _lib.LIB_PATH = os.environ['L2I_LIB_PATH']
HERE = os.path.dirname(os.path.abspath(__file__))


def build(shape, family):
    B, cin, cout, kh, kw, stride, H, W, OH, OW, step, mask, scale = shape[:13]
    dev = 'cuda'
    x = torch.randn(B, cin, H, W, device=dev)
    kw_ = {}
    if mask:
        kw_['in_mask'] = torch.randn(B, cin, H, W, device=dev)
    if scale:
        kw_['in_scale'] = torch.rand(B, cin, device=dev) + 0.5
    w = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    if family.startswith('transposed'):
        F = conv.FusedTransposed(w, 0 if (OH - 2 * H) in (1, 4) else 1).to(dev)
        y = torch.empty(B, cout, OH, OW, device=dev)
        return lambda: conv.run_fused_transposed(F, x, y, **kw_), 2.0 * B * cout * cin * kh * kw * H * W
    if step != 1:
        return None, 0
    pad = max(0, ((OH - 1) * stride + kh - H + 1) // 2)
    L = conv.Launch(w, stride, pad, pad, device=dev)
    y = torch.zeros(B, cout, OH, OW, device=dev)
    epi = shape[13]
    act = int(epi[-1])
    ops = epi[:-1]
    full = lambda: torch.randn(B, cout, OH, OW, device=dev)
    if 'd' in ops:
        kw_['out_scale'] = torch.rand(B, cout, device=dev) + 0.5
    if 'n' in ops:
        kw_['noise'], kw_['noise_w'] = torch.randn(B, 1, OH, OW, device=dev), 0.05
    if 'b' in ops:
        kw_['bias'] = torch.randn(cout, device=dev)
    if 'r' in ops:
        kw_['residual'] = full()
    if 'm' in ops:
        kw_['res_mask'] = full()
    if 'o' in ops:
        kw_['out_mask'] = full()
    if 's' in ops:
        kw_['res_sub'], kw_['res_coef'] = full(), 0.5
    if 'a' in ops:
        kw_['accumulate'] = True
    return lambda: conv.run_launch(L, x, y, act=act, **kw_), 2.0 * B * cout * cin * kh * kw * OH * OW


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    rows = json.load(open(os.path.join(HERE, 'c3_launch_shapes.json')))
    tot_old = tot_new = 0.0
    out = []
    for r in rows:
        if only and only not in r['family']:
            continue
        if r['n'] * r['ms'] < min_ms:
            continue
        fn, flops = build(r['shape'], r['family'])
        if fn is None:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = max(3, min(50, int(20.0 / max(r['ms'], 0.02))))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot_old += r['n'] * r['ms']
        tot_new += r['n'] * ms
        out.append(dict(family=r['family'], shape=r['shape'], n=r['n'], ms=round(ms, 4), recorded_ms=r['ms']))
        print('%-18s %-70s n=%4.1f  %7.3f ms (recorded %7.3f)  %6.1f TFLOP/s' % (r['family'], str(r['shape']), r['n'], ms, r['ms'], flops / ms / 1e9), flush=True)
        del fn
        torch.cuda.empty_cache()
    print('per step: %.2f ms now, %.2f ms recorded' % (tot_new, tot_old))
    if os.environ.get('L2I_BENCH_OUT'):
        json.dump(out, open(os.environ['L2I_BENCH_OUT'], 'w'))


if __name__ == '__main__':
    main()
