"""Debug aid: block outputs of the training-mode ResNet-50 forward, HIP vs the float64 oracle, beside the float32 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from latent2im_amd import synth, conv, regressor_train as RT
from oracle import nets as onets, step as ostep
if len(sys.argv) > 1 and sys.argv[1] == 'nowino':
    conv.USE_WINOGRAD = False
torch.set_num_threads(32)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rs = np.random.RandomState(5)
st = synth.resnet50_state(seed=300)
data = T(rs.randn(8, 3, 128, 128).astype(np.float32) * 0.5)
model = RT.TrainableResNet50(st, device='cuda')
model(data.cuda())
hip = [b[7].cpu().double() for b in model._saved['blocks']]


def blocks(P, x):
    outs = []
    bn = lambda p, t: F.batch_norm(t, None, None, P[p + '.weight'], P[p + '.bias'], training=True, eps=1e-5)
    x = F.max_pool2d(F.relu(bn('bn1', F.conv2d(x, P['conv1.weight'], stride=2, padding=3))), 3, 2, 1)
    for li, (planes, nb, stride) in enumerate(onets.RESNET50_LAYERS):
        for b in range(nb):
            p = 'layer%d.%d' % (li + 1, b)
            s = stride if b == 0 else 1
            idt = x
            o = F.relu(bn(p + '.bn1', F.conv2d(x, P[p + '.conv1.weight'])))
            o = F.relu(bn(p + '.bn2', F.conv2d(o, P[p + '.conv2.weight'], stride=s, padding=1)))
            o = bn(p + '.bn3', F.conv2d(o, P[p + '.conv3.weight']))
            if b == 0:
                idt = bn(p + '.downsample.1', F.conv2d(x, P[p + '.downsample.0.weight'], stride=s))
            x = F.relu(o + idt)
            outs.append(x)
    return outs


o64 = blocks(ostep.to_torch(st, torch.float64), data.double())
o32 = blocks(ostep.to_torch(st), data)
for i, (h, a, b) in enumerate(zip(hip, o64, o32)):
    m = float(a.abs().max())
    print('block %2d  hip %.2e  oracle32 %.2e   flips hip %d oracle32 %d of %d' % (i, float((h - a).abs().max()) / m, float((b.double() - a).abs().max()) / m,
          int((h > 0).ne(a > 0).sum()), int((b > 0).ne(a > 0).sum()), a.numel()))
