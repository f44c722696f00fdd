"""Whole walk-training step (1024^2, batch 8, full loss) as ONE chain against TWO strided half-batch chains on two streams (each with its
own three loss-branch streams): does the HBM-bound work of one chain overlap the matrix-bound work of the other?  (GPU box; timing and a
consistency check of the two walk gradients.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import capture, constants, selfcheck, synth

B = 8
constants.SYNTH_NOISE_STRENGTH = 0.0
g = selfcheck.build_graph(1024, ['Smiling'], B, lr=1e-4)
zs = torch.Tensor(synth.z_sample(B, seed=0)).cuda()
alpha = torch.full((B, 1), 0.3, device='cuda')
chains = [torch.cuda.Stream(), torch.cuda.Stream()]
sides = [(torch.cuda.Stream(), torch.cuda.Stream()), (torch.cuda.Stream(), torch.cuda.Stream())]


def one():
    capture.forward_backward(g, zs, alpha)
    return g.walk.w.grad.clone()


def two():
    cur = torch.cuda.current_stream()
    g.optimizers.zero_grad()
    feeds = []
    for c in range(2):
        chains[c].wait_stream(cur)
        with torch.cuda.stream(chains[c]):
            g._streams = sides[c]
            feed, _ = capture.forward(g, zs[c::2].contiguous(), alpha[c::2].contiguous())
            feeds.append(feed)
    for c in range(2):
        with torch.cuda.stream(chains[c]):
            g._streams = sides[c]
            loss = g.get_w_loss(feeds[c]) * 0.5
            loss.backward()
    for c in range(2):
        cur.wait_stream(chains[c])
    return g.walk.w.grad.clone()


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g1 = one()
g2 = two()
torch.cuda.synchronize()
print('gradient: max |two - one| / max |one| = %.2e' % float((g2 - g1).abs().max() / g1.abs().max()))
print('one chain  B=8        %.2f ms' % timeit(one))
print('two chains 2 x B=4    %.2f ms' % timeit(two))
