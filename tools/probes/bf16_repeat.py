"""Repeats ONE training step (no optimizer update) on identical inputs in one process and reports how many distinct walk gradients came out.
usage: python tools/probes/bf16_repeat.py [precision] [size] [batch] [repeats] [no_gan]   (run two at once to perturb the timing)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from latent2im_amd import conv, selfcheck, synth
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
no_gan = len(sys.argv) > 5 and sys.argv[5] == '1'
conv.PRECISION = prec
from latent2im_amd import constants
if os.environ.get('L2I_SERIAL'):
    constants.CONCURRENT_LOSS_BRANCHES = False
g = selfcheck.build_graph(size, ['Smiling', 'Young'], batch, lr=1e-3)
zs = synth.z_sample(batch, seed=3)
alpha = np.ones((batch, 2)) * np.asarray([0.3, 0.7])
seen = {}
first = None
for i in range(reps):
    r = selfcheck.run_step(g, zs, alpha, no_gan_loss=no_gan, optimize=False)
    torch.cuda.synchronize()
    gr = r['grad'].detach().float().cpu().numpy()
    terms = [float(r['loss'])] + [float(r['terms'][k]) for k in ('reg', 'cont')] + ([] if no_gan else [float(r['terms']['gan'])])
    h = hashlib.md5(gr.tobytes()).hexdigest()[:8]
    hh = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:6]
    parts = 'x0 %s x1 %s a0 %s' % (hh(r['x0']), hh(r['x1']), hh(r['a0']))
    if first is None:
        first = gr
    cos = float((gr * first).sum() / (np.linalg.norm(gr) * np.linalg.norm(first)))
    seen.setdefault(h, []).append(i)
    if len(seen[h]) == 1:
        print('new gradient at repeat %d: %s cos vs first %.6f losses %s | %s' % (i, h, cos, ' '.join('%.8f' % t for t in terms), parts), flush=True)
print('%s %d^2 batch %d: %d repeats, %d distinct gradients: %s' % (prec, size, batch, reps, len(seen), {k: len(v) for k, v in seen.items()}))
