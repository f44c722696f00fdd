"""Does splitting a batch-8 generator pass into two batch-4 chains on two streams overlap the HBM-bound elementwise kernels of one chain
with the matrix-bound convs of the other?  (GPU box only; timing probe, nothing is checked.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import synth
from latent2im_amd.generator import Generator

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = Generator(synth.generator_state(res, seed=100, noise_strength=0.05), res, device='cuda')
B = 8
lat = torch.randn(B, G.n_latent, 512, device='cuda') * 0.5


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


def fwd_one():
    with torch.no_grad():
        G.synthesis(lat)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            G.synthesis(lat[0::2].contiguous())
        with torch.cuda.stream(s2):
            G.synthesis(lat[1::2].contiguous())
    cur.wait_stream(s1); cur.wait_stream(s2)


def fwdbwd_one():
    l = lat.clone().requires_grad_(True)
    img = G.synthesis(l)
    img.backward(torch.ones_like(img))


def fwdbwd_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for s, sl in ((s1, slice(0, None, 2)), (s2, slice(1, None, 2))):
        with torch.cuda.stream(s):
            l = lat[sl].contiguous().requires_grad_(True)
            img = G.synthesis(l)
            img.backward(torch.ones_like(img))
    cur.wait_stream(s1); cur.wait_stream(s2)


def fwd_half():
    with torch.no_grad():
        G.synthesis(lat[0::2].contiguous())


print('forward  B=8 one stream      %.2f ms' % timeit(fwd_one))
print('forward  B=4 alone           %.2f ms' % timeit(fwd_half))
print('forward  2 x B=4 two streams %.2f ms' % timeit(fwd_two))
print('fwd+bwd  B=8 one stream      %.2f ms' % timeit(fwdbwd_one))
print('fwd+bwd  2 x B=4 two streams %.2f ms' % timeit(fwdbwd_two))
