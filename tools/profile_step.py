"""Per-shape breakdown of the conv launches of one training step (GPU box): time, TFLOP/s, share."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from latent2im_amd import constants, conv, selfcheck, synth

constants.CONCURRENT_LOSS_BRANCHES = False          # one stream: per-launch durations do not overlap

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
np.random.seed(1234)
g = selfcheck.build_graph(res, ['Smiling'], B, lr=1e-4)
zs = synth.z_sample(B * 3, seed=0)
alpha = np.ones((B, 1)) * 0.3
selfcheck.run_step(g, zs[:B], alpha)
torch.cuda.synchronize()
conv.PROFILE = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
selfcheck.run_step(g, zs[B:2 * B], alpha)
e1.record()
torch.cuda.synchronize()
prof, conv.PROFILE = conv.PROFILE, None
agg = defaultdict(lambda: [0, 0.0, 0.0])
for a, b, fl, d, _name in prof:
    k = d
    agg[k][0] += 1
    agg[k][1] += a.elapsed_time(b)
    agg[k][2] += fl
tot = sum(v[1] for v in agg.values())
print('step %.1f ms, conv %.1f ms in %d launches, %.1f TFLOP/s' % (e0.elapsed_time(e1), tot, len(prof), sum(v[2] for v in agg.values()) / tot / 1e9))
print('%-74s %4s %9s %8s %6s' % ('(B,cin,cout,kh,kw,stride,H,W,OH,OW,step,mask,scale)', 'n', 'ms', 'TF/s', 'share'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("L2I_PROFILE_ROWS", "60"))]:
    print('%-74s %4d %9.3f %8.1f %5.1f%%' % (str(k), v[0], v[1], v[2] / v[1] / 1e9, 100 * v[1] / tot))


def cat(d):
    B, cin, cout, kh, kw, s, H, W, OH, OW, step, mask, scale = d[:13]
    if cout <= 3:
        return 'cout<=3'
    if step == 2:
        return 'phase(step2)'
    if kh == 1 and kw == 1:
        return '1x1 s%d' % s
    if cin <= 4:
        return 'cin<=3'
    if H <= 32:
        return 'kxk small map'
    if s == 2:
        return 'kxk s2'
    return '3x3 big'


cats = defaultdict(lambda: [0, 0.0, 0.0])
for k, v in agg.items():
    c = cat(k)
    cats[c][0] += v[0]
    cats[c][1] += v[1]
    cats[c][2] += v[2]
print('--- by category ---')
for c, v in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print('%-16s n=%4d ms=%8.2f  TF/s=%6.1f  share=%5.1f%%' % (c, v[0], v[1], v[2] / v[1] / 1e9, 100 * v[1] / tot))
