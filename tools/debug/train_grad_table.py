"""Debug aid: per-parameter gradient error of one regressor training step, HIP vs the float64 oracle, beside the float32 oracle's own error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from latent2im_amd import synth, conv, regressor_train as RT
from oracle import nets as onets, step as ostep
torch.set_num_threads(32)
if len(sys.argv) > 1 and sys.argv[1] == 'nowino':
    conv.USE_WINOGRAD = False
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rs = np.random.RandomState(5)
st = synth.resnet50_state(seed=300)
data = T(rs.randn(8, 3, 128, 128).astype(np.float32) * 0.5)
label = T(rs.rand(8, 40).astype(np.float32))
model = RT.TrainableResNet50(st, device='cuda')
P64 = {k: v.clone() for k, v in ostep.to_torch(st, torch.float64).items()}
_, g64 = onets.resnet50_train_step(P64, data.double(), label.double())
P = {k: v.clone() for k, v in ostep.to_torch(st).items()}
_, g1 = onets.resnet50_train_step(P, data, label)
preds = model(data.cuda())
loss, g = RT.mse_loss_and_grad(preds, label.cuda())
grads = model.backward(g)
rows = []
for k, want in g64.items():
    sc = float(want.abs().max()) + 1e-300
    rows.append((float((grads[k].cpu().double() - want).abs().max()) / sc, float((g1[k].double() - want).abs().max()) / sc, k))
for e, o, k in rows:
    print('%-36s hip %.2e  oracle32 %.2e  %s' % (k, e, o, '<<<' if e > max(2 * o, 5e-3) else ''))
print('median hip %.2e  median oracle32 %.2e' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
