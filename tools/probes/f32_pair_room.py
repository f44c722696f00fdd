"""[r6] How much room would a fused fp32 pair (expand + identity + ReLU -> next reduce + ReLU) have?  Times the two fp32 launches of ResNet-50's trunk shapes at
batch 8 (rotating buffers) beside their floors: fp32 MFMA at 157.3 TFLOP/s, HBM at 5 TB/s for the two launches and for a fused one."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from latent2im_amd import conv  # noqa: E402

DEV, B, NSET, INNER, REPS = 'cuda', 8, 3, 6, 5


def timeit(fn):
    for i in range(2):
        fn(i % NSET)
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(INNER):
            fn(i % NSET)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / INNER * 1e3)
    return float(np.median(ts))


rs = np.random.RandomState(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
for name, c1, c2, c3, hw in [('layer1 256^2', 64, 256, 64, 256), ('layer2 128^2', 128, 512, 128, 128), ('layer3 64^2', 256, 1024, 256, 64), ('layer4 32^2', 512, 2048, 512, 32)]:
    A = conv.FrozenConv2d(T(rs.randn(c2, c1, 1, 1) / np.sqrt(c1)), 1, 0, device=DEV)
    Bc = conv.FrozenConv2d(T(rs.randn(c3, c2, 1, 1) / np.sqrt(c2)), 1, 0, device=DEV)
    mk = lambda c: [torch.randn(B, c, hw, hw, device=DEV) for _ in range(NSET)]
    xs, rss, mids, outs = mk(c1), mk(c2), mk(c2), mk(c3)
    ba, bb = torch.randn(c2, device=DEV), torch.randn(c3, device=DEV)
    t1 = timeit(lambda i: A.forward(xs[i], out=mids[i], bias=ba, residual=rss[i], act=conv.ACT_RELU))
    t2 = timeit(lambda i: Bc.forward(mids[i], out=outs[i], bias=bb, act=conv.ACT_RELU))
    tp = float('nan')
    if conv.pair_f32_shapes_ok(c1, c2, c3, hw * hw):
        def pair(i):
            d = []
            A.forward(xs[i], out=mids[i], bias=ba, residual=rss[i], act=conv.ACT_RELU, _defer=d)
            Bc.forward(mids[i], out=outs[i], bias=bb, act=conv.ACT_RELU, _defer=d)
            conv.launch_pair_f32(d)
        tp = timeit(pair)
    npx = B * hw * hw
    flop = 2.0 * npx * (c1 * c2 + c2 * c3)
    by2 = 4.0 * npx * ((c1 + 2 * c2) + (c2 + c3))
    by1 = 4.0 * npx * (c1 + 2 * c2 + c3)
    print('%-14s expand %6.1f us  reduce %6.1f us  sum %6.1f  PAIR %6.1f | MFMA floor %6.1f  HBM two %6.1f  HBM fused %6.1f' %
          (name, t1, t2, t1 + t2, tp, flop / 157.3e6, by2 / 5e6, by1 / 5e6), flush=True)
