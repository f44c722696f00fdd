"""Debug aid: the LAST bottleneck block of the trainable ResNet-50: forward intermediates and parameter gradients, HIP vs torch float64 autograd on
the same inputs (the block's input `cur` and the incoming gradient `g` taken from the HIP run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from latent2im_amd import synth, regressor_train as RT
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rs = np.random.RandomState(5)
st = synth.resnet50_state(seed=300)
data = T(rs.randn(8, 3, 128, 128).astype(np.float32) * 0.5).cuda()
label = T(rs.rand(8, 40).astype(np.float32)).cuda()
model = RT.TrainableResNet50(st, device='cuda')
preds = model(data)
sv = model._saved
cur, y1, s1, y2, s2, s3, sd, out = sv['blocks'][-1]
loss, gp = RT.mse_loss_and_grad(preds, label)
b, c, h, w = sv['last']
g = (torch.mm(gp, model.fc_w) * (1.0 / (h * w))).reshape(b, c, 1, 1).expand(b, c, h, w).contiguous()
blk = model.blocks[-1]
# torch float64 reference of the block on the CPU
D = lambda t: t.detach().double().cpu()
ws = {k: D(blk[k].weight).requires_grad_(True) for k in ('c1', 'c2', 'c3')}
bs = {k: (D(blk[k].weight).requires_grad_(True), D(blk[k].bias).requires_grad_(True)) for k in ('b1', 'b2', 'b3')}
bn = lambda t, k: F.batch_norm(t, None, None, bs[k][0], bs[k][1], training=True, eps=1e-5)
x = D(cur)
z1 = F.conv2d(x, ws['c1']); r1 = F.relu(bn(z1, 'b1'))
z2 = F.conv2d(r1, ws['c2'], padding=1); r2 = F.relu(bn(z2, 'b2'))
z3 = F.conv2d(r2, ws['c3']); o = F.relu(bn(z3, 'b3') + x)
rel = lambda a, bb: float((D(a) - bb.detach()).abs().max() / bb.detach().abs().max())
print('fwd  y1 %.2e  y2 %.2e  out %.2e   (z: s1 %.2e s2 %.2e s3 %.2e)' % (rel(y1, r1), rel(y2, r2), rel(out, o), rel(s1[0], z1), rel(s2[0], z2), rel(s3[0], z3)))
params = [ws['c1'], bs['b1'][0], bs['b1'][1], ws['c2'], bs['b2'][0], bs['b2'][1], ws['c3'], bs['b3'][0], bs['b3'][1]]
gr = torch.autograd.grad(o, params, D(g))
names = ['c1.w', 'b1.w', 'b1.b', 'c2.w', 'b2.w', 'b2.b', 'c3.w', 'b3.w', 'b3.b']
# HIP backward of the block, step by step
g_z3, dw3, db3, gm = blk['b3'].backward(g, s3, out_mask=out, want_masked=True)
wg3 = blk['c3'].wgrad(y2, g_z3)
g_y2 = blk['c3'].dgrad(g_z3, (y2.shape[2], y2.shape[3]))
g_z2, dw2, db2, _ = blk['b2'].backward(g_y2, s2, out_mask=y2)
wg2 = blk['c2'].wgrad(y1, g_z2)
g_y1 = blk['c2'].dgrad(g_z2, (y1.shape[2], y1.shape[3]))
g_z1, dw1, db1, _ = blk['b1'].backward(g_y1, s1, out_mask=y1)
wg1 = blk['c1'].wgrad(cur, g_z1)
got = [wg1, dw1, db1, wg2, dw2, db2, wg3, dw3, db3]
for n, a, r in zip(names, got, gr):
    print('%-6s %.2e' % (n, rel(a, r)))
mask_ref = (o > 0)
print('mask flips in out: %d of %d' % (int((D(out) > 0).ne(mask_ref).sum()), mask_ref.numel()))
print('g_z3 vs autograd dz3:', rel(g_z3, torch.autograd.grad(o, z3, D(g))[0]))
