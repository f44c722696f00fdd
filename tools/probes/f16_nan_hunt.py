"""debugging aid: where the fp16 path's training step first goes non-finite (bench conditions: noise weights 0.05, batch 8, several Adam steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent2im_amd import conv, constants, selfcheck, synth, nets16
conv.PRECISION = sys.argv[1] if len(sys.argv) > 1 else 'f16'
constants.SYNTH_NOISE_STRENGTH = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
size, batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024, int(sys.argv[4]) if len(sys.argv) > 4 else 8
attrs = ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk']
np.random.seed(1234)
g = selfcheck.build_graph(size, attrs, batch, lr=1e-4, transform='scene')
print('scales', g.loss_scaler.log2 if g.loss_scaler is not None else None)
zs = synth.z_sample(batch * 6, seed=0)
for i in range(6):
    alpha = np.ones((batch, 5)) * np.random.uniform(-1, 1, 5)
    nets16.PROBE = [] if i in (0, 5) else None
    r = selfcheck.run_step(g, zs[i * batch:(i + 1) * batch], alpha, clamp=True)
    torch.cuda.synchronize()
    t = {k: float(v) for k, v in r['terms'].items()}
    print('step', i, 'loss', float(r['loss']), t, 'x1 finite', bool(torch.isfinite(r['x1']).all()), 'x1 max', float(r['x1'].abs().max()), 'grad finite', bool(torch.isfinite(r['grad']).all()),
          'grad max', float(r['grad'].abs().max()), 'walk finite', bool(torch.isfinite(g.walk.w).all()))
    if nets16.PROBE:
        bad = [(tag, mx) for tag, shape, mx, med, zf in nets16.PROBE if not np.isfinite(mx) or mx > 3e4]
        print('   maps with max > 3e4 or non-finite:', bad[:12])
        fw = [(tag, round(mx, 1)) for tag, shape, mx, med, zf in nets16.PROBE if '.fwd.' in tag]
        print('   forward maxima:', fw[:40])
    nets16.PROBE = None

# the same workload replayed from the hipGraph (bench.py --config c5)
from latent2im_amd import capture
print('--- captured step')
np.random.seed(1234)
g2 = selfcheck.build_graph(size, attrs, batch, lr=1e-4, transform='scene')
cap = capture.CapturedStep(g2, batch, 5, clamp=True)
for i in range(4):
    alpha = np.ones((batch, 5)) * np.random.uniform(-1, 1, 5)
    r = cap(zs[i * batch:(i + 1) * batch], alpha)
    torch.cuda.synchronize()
    print('replay', i, 'loss', float(r['loss']), {k: float(v) for k, v in r['terms'].items()}, 'x0 finite', bool(torch.isfinite(r['x0']).all()), 'x1 finite', bool(torch.isfinite(r['x1']).all()),
          'a0', [round(float(v), 4) for v in r['a0'][0]], 'eps finite', bool(torch.isfinite(r['eps']).all()), 'grad finite', bool(torch.isfinite(r['grad']).all()), 'walk finite', bool(torch.isfinite(g2.walk.w).all()))
