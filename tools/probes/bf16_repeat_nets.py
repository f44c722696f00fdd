"""Repeats the forward of each 16-bit network on identical inputs in one process and reports the number of distinct outputs per network.
usage: python tools/probes/bf16_repeat_nets.py [size] [batch] [repeats]   (run two at once to perturb the timing)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, nets16, synth
conv.PRECISION = 'bf16'
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = 'cuda'
rs = np.random.RandomState(1)
G = nets16.Generator(synth.generator_state(size, seed=100, noise_strength=0.05), size, device=dev)
D = nets16.Discriminator(synth.discriminator_state(size, seed=200), size, device=dev)
R = nets16.ResNet50(synth.resnet50_state(seed=300), device=dev)
V = nets16.VGG19Prefix(synth.vgg19_prefix_state(seed=400), device=dev)
lat = torch.from_numpy(rs.randn(batch, G.n_latent, 512)).float().to(dev)
noise = [torch.from_numpy(n).float().to(dev) for n in synth.noise_maps(size, batch)]
img = torch.from_numpy(rs.randn(batch, 3, size, size)).float().to(dev)
h = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
runs = {'G': lambda: h(G.synthesis(lat, noise)), 'D': lambda: h(D(img)), 'R': lambda: h(R(img)),
        'V': lambda: h(torch.cat([t.float().reshape(-1) for t in V.taps(img)[:5] if torch.is_tensor(t)]))}
seen = {k: {} for k in runs}
with torch.no_grad():
    for i in range(reps):
        for k, f in runs.items():
            v = f()
            torch.cuda.synchronize()
            seen[k][v] = seen[k].get(v, 0) + 1
for k in runs:
    print(k, len(seen[k]), 'distinct outputs in', reps, 'repeats', sorted(seen[k].values(), reverse=True)[:6])
