"""[r6] l2i_conv1x1_pair_h8 against the two l2i_conv2d_h8 launches it replaces, on ResNet-50's trunk shapes at batch 8 (1024^2 regressor input):
forward form (conv3 + identity + ReLU -> next block's conv1 + ReLU, sign planes written) and backward form (sign-plane masks).  Median of `reps` timed groups of
`inner` launches over rotating buffer sets (so that no operand stays in L2 / MALL between launches).   usage: python tools/probes/pair_bench.py [f16|bf16]"""
import sys
import numpy as np
import torch

sys.path.insert(0, '.')
from latent2im_amd import conv  # noqa: E402

DEV = 'cuda'
conv.PRECISION = sys.argv[1] if len(sys.argv) > 1 else 'f16'
SHAPES = [('layer1 256^2', 64, 256, 64, 256), ('layer1->2 256^2', 64, 256, 128, 256), ('layer2 128^2', 128, 512, 128, 128), ('layer2->3 128^2', 128, 512, 256, 128),
          ('layer3 64^2', 256, 1024, 256, 64)]
B, NSET, INNER, REPS = 8, 4, 8, 7
import os
if os.environ.get('PAIR_BENCH_ONLY'):
    SHAPES = [q for q in SHAPES if q[0].startswith(os.environ['PAIR_BENCH_ONLY'])]


def timeit(fn):
    for i in range(3):
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


def main():
    rs = np.random.RandomState(0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    print('# %s elements, batch %d; us per launch (pair) / per two launches' % (conv.PRECISION, B))
    for name, c1, c2, c3, hw in SHAPES:
        A = conv.H8Conv(T(rs.randn(c2, c1, 1, 1) / np.sqrt(c1)), 1, 0, device=DEV)
        Bc = conv.H8Conv(T(rs.randn(c3, c2, 1, 1) / np.sqrt(c2)), 1, 0, device=DEV)
        dt = conv.h8_dtype()
        mk = lambda c: [torch.randn(B, c // 8, hw, hw, 8, device=DEV).to(dt) for _ in range(NSET)]
        xs, rss, mids, outs = mk(c1), mk(c2), mk(c2), mk(c3)
        pl = lambda c: [torch.randint(0, 256, (B, c // 8, hw, hw), device=DEV, dtype=torch.uint8) for _ in range(NSET)]
        pm, po = pl(c2), pl(c3)
        ba, bb = torch.randn(c2, device=DEV), torch.randn(c3, device=DEV)
        unit = B * c2 * hw * hw * 2 / 1e6            # MB of one wide map

        def two_f(i):
            A.forward(xs[i], out=mids[i], bias=ba, residual=rss[i], act=conv.ACT_RELU, mask_out=pm[i])
            Bc.forward(mids[i], out=outs[i], bias=bb, act=conv.ACT_RELU, mask_out=po[i])

        def pair_f(i, variant=0):
            d = []
            A.forward(xs[i], out=mids[i], bias=ba, residual=rss[i], act=conv.ACT_RELU, mask_out=pm[i], _defer=d)
            Bc.forward(mids[i], out=outs[i], bias=bb, act=conv.ACT_RELU, mask_out=po[i], _defer=d)
            conv.launch_pair_h8(d, variant=variant)

        def two_b(i):
            A.forward(xs[i], out=mids[i], residual=rss[i], out_mask=pm[i], res_mask=pm[i], mask_bits=True)
            Bc.forward(mids[i], out=outs[i], out_mask=po[i], mask_bits=True)

        def pair_b(i, variant=0):
            d = []
            A.forward(xs[i], out=mids[i], residual=rss[i], out_mask=pm[i], res_mask=pm[i], mask_bits=True, _defer=d)
            Bc.forward(mids[i], out=outs[i], out_mask=po[i], mask_bits=True, _defer=d)
            conv.launch_pair_h8(d, variant=variant)

        if os.environ.get('PAIR_BENCH_CHAIN3') and c3 <= 128 and c1 <= 128:
            H3 = conv.H8Conv(T(rs.randn(c1, c1, 3, 3) / np.sqrt(9 * c1)), 1, 1, device=DEV, cin_pad=16)
            x3, y3s, p3 = mk(c1), mk(c1), pl(c1)
            b3 = torch.randn(c1, device=DEV)

            def three_f(i):
                H3.forward(x3[i], out=xs[i], bias=b3, act=conv.ACT_RELU, mask_out=p3[i])
                two_f(i)

            def chain_f(i, variant=0):
                d = []
                H3.forward(x3[i], out=xs[i], bias=b3, act=conv.ACT_RELU, mask_out=p3[i], _defer=d)
                A.forward(xs[i], out=mids[i], bias=ba, residual=rss[i], act=conv.ACT_RELU, mask_out=pm[i], _defer=d)
                Bc.forward(mids[i], out=outs[i], bias=bb, act=conv.ACT_RELU, mask_out=po[i], _defer=d)
                d[0][0].y = None
                conv.launch_pair_h8(d, variant=variant)

            def three_b(i):
                H3.forward(x3[i], out=xs[i], out_mask=p3[i], mask_bits=True)
                two_b(i)

            def chain_b(i, variant=0):
                d = []
                H3.forward(x3[i], out=xs[i], out_mask=p3[i], mask_bits=True, _defer=d)
                A.forward(xs[i], out=mids[i], residual=rss[i], out_mask=pm[i], res_mask=pm[i], mask_bits=True, _defer=d)
                Bc.forward(mids[i], out=outs[i], out_mask=po[i], mask_bits=True, _defer=d)
                d[0][0].y = None
                conv.launch_pair_h8(d, variant=variant)

            row = '%-18s chain3 | fwd three launches %6.1f  3x3 + pair %6.1f' % (name, timeit(three_f), timeit(lambda i: (H3.forward(x3[i], out=xs[i], bias=b3, act=conv.ACT_RELU, mask_out=p3[i]), pair_f(i, 0 if c1 == 64 else 1))))
            for v in (0, 1):
                row += '  chain3[v%d] %6.1f' % (v, timeit(lambda i: chain_f(i, v)))
            row += ' | bwd three %6.1f' % timeit(three_b)
            for v in (0, 1):
                row += '  chain3[v%d] %6.1f' % (v, timeit(lambda i: chain_b(i, v)))
            print(row, flush=True)
            continue
        t2f, t2b = timeit(two_f), timeit(two_b)
        row = '%-18s wide map %6.1f MB | fwd two %6.1f' % (name, unit, t2f)
        for v in (0, 1):
            row += '  pair[v%d] %6.1f' % (v, timeit(lambda i: pair_f(i, v)))
        row += ' | bwd two %6.1f' % t2b
        for v in (0, 1):
            row += '  pair[v%d] %6.1f' % (v, timeit(lambda i: pair_b(i, v)))
        alg = (c1 + 2 * c2 + c3) / c2 * unit
        row += ' | pair bytes %.0f MB = %.1f us at 5 TB/s' % (alg, alg / 5.0)
        print(row, flush=True)


if __name__ == '__main__':
    main()
