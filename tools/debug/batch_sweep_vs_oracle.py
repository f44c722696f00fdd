"""Debug aid: every frozen network at small resolutions over a sweep of batch sizes against the CPU oracle (checker), to localise
batch-dependent kernel paths (sample packing, split-K, tile selection)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from latent2im_amd import synth
from latent2im_amd.generator import Generator
from latent2im_amd.regressor import ResNet50
from latent2im_amd.perceptual import VGG19Prefix
from latent2im_amd.discriminator import Discriminator
from oracle import sg2, nets as onets, step as ostep
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.set_num_threads(32)
rel = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max() / b.double().abs().max())
for size in (32, 64, 128):
    stG, stR, stV, stD = synth.generator_state(size, seed=100), synth.resnet50_state(seed=300), synth.vgg19_prefix_state(seed=400), synth.discriminator_state(size, seed=200)
    G, R, V, D = Generator(stG, size, device='cuda'), ResNet50(stR, device='cuda'), VGG19Prefix(stV, device='cuda'), Discriminator(stD, size, device='cuda')
    PG, PR, PV, PD = (ostep.to_torch(s) for s in (stG, stR, stV, stD))
    for B in (1, 2, 4, 8, 16):
        z = T(synth.z_sample(B, seed=3)).float()
        w = G.style(z.cuda())
        lat = torch.stack([w] * G.n_latent, 1).contiguous()
        img = G.synthesis(lat)
        wo = sg2.style_mlp(PG, z)
        img_o = sg2.generator_synthesis(PG, torch.stack([wo] * G.n_latent, 1), None)
        e_img = rel(img, img_o)
        e_r = rel(R(img_o.cuda()), onets.resnet50_forward(PR, img_o))
        e_d = rel(D(img_o.cuda()), sg2.discriminator_forward(PD, img_o))
        other = torch.roll(img_o, 3, 3)
        _, lo = ostep.content_loss(PV, other, img_o)
        e_v = rel(V.content_losses(other.cuda(), img_o.cuda()), torch.stack(lo))
        print('size %4d B %2d  G %.2e  R %.2e  D %.2e  V %.2e' % (size, B, e_img, e_r, e_d, e_v), flush=True)
