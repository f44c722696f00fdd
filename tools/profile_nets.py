"""Time each frozen network of the step in isolation at the bench shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent2im_amd import selfcheck, synth

res, B = 1024, 8
np.random.seed(0)
g = selfcheck.build_graph(res, ['Smiling'], B)
G, D, R, V = g.module.netG, g.module.netD, g.regressor, g.vgg19


def timeit(name, fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('%-28s %8.2f ms' % (name, min(ts)), flush=True)
    return min(ts)


z = torch.from_numpy(synth.z_sample(B, 0)).float().cuda()
w = G.style(z)
lat = torch.stack([w] * G.n_latent, 1).contiguous()
img = G.synthesis(lat).detach()
img2 = (img + 0.01 * torch.randn_like(img)).detach()
tot = 0
tot += timeit('G forward (x0 pass)', lambda: G.synthesis(lat))
def g_fb():
    l = lat.clone().requires_grad_(True)
    G.synthesis(l).backward(torch.ones_like(img))
tot += timeit('G forward+backward (x1)', g_fb)
tot += timeit('R forward (x0)', lambda: R(img))
def r_fb():
    x = img.clone().requires_grad_(True); R(x).sum().backward()
tot += timeit('R forward+backward (x1)', r_fb)
def v_fb():
    x = img2.clone().requires_grad_(True); V.content_losses(img, x).sum().backward()
tot += timeit('V taps(x0)+loss fwd+bwd(x1)', v_fb)
def d_fb():
    x = img.clone().requires_grad_(True); D(x).sum().backward()
tot += timeit('D forward+backward (x1)', d_fb)
print('sum %.1f ms' % tot)
