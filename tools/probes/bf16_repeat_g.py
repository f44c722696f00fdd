"""Two generator passes back to back (no host sync between them), repeated on identical inputs: distinct outputs per pass.  Run two at once."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, nets16, synth
conv.PRECISION = 'bf16'
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sync = len(sys.argv) > 4 and sys.argv[4] == 'sync'
dev = 'cuda'
rs = np.random.RandomState(1)
G = nets16.Generator(synth.generator_state(size, seed=100), size, device=dev)
l1 = torch.from_numpy(rs.randn(batch, G.n_latent, 512)).float().to(dev)
l2 = (l1 + 0.02 * torch.from_numpy(rs.randn(batch, G.n_latent, 512)).float().to(dev)).requires_grad_(True)
h = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:6]
seen = [{}, {}, {}]
gy = torch.from_numpy(rs.randn(batch, 3, size, size)).float().to(dev)
for i in range(reps):
    with torch.no_grad():
        x0 = G.synthesis(l1)
    if sync:
        torch.cuda.synchronize()
    x1 = G.synthesis(l2)
    l2.grad = None
    x1.backward(gy)
    torch.cuda.synchronize()
    for k, v in enumerate((h(x0), h(x1), h(l2.grad))):
        seen[k][v] = seen[k].get(v, 0) + 1
for k, name in enumerate(('x0', 'x1', 'dlatent')):
    print(name, len(seen[k]), 'distinct in', reps, sorted(seen[k].values(), reverse=True)[:5])
