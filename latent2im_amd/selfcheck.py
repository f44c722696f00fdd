"""smoke(): one tiny walk-training step (64^2 generator, batch 4, full loss) on cuda:0, checked against the CPU oracle.
Called by __graft_entry__.smoke(); the oracle import lives HERE only because smoke() is one of the three places allowed
to use the checker (it is never on the product path)."""
import types

import numpy as np
import torch


def build_graph(resolution, attr_names, batch_size, lr=1e-3, walk_seed=7, device=None, transform='face'):
    """A faceGraph (CelebA attribute table) or SceneGraph (transient-scene table) on synthetic weights without touching the CLI
    (used by smoke(), bench.py and the tests)."""
    from . import constants, graph, synth
    constants.resolution = resolution
    constants.BATCH_SIZE = batch_size
    constants.ALLOW_SYNTHETIC_WEIGHTS = True              # explicit: this helper exists to build the seeded random-init networks
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'dataset', 'attributes_celeba.txt' if transform == 'face' else 'attributes_scene.txt')) as f:
        names = [l.strip() for l in f if l.strip()]
    table = {n: i for i, n in enumerate(names)}
    state = np.random.get_state()
    cls = {'face': graph.faceGraph, 'scene': graph.SceneGraph}[transform]      # (find_model_using_name prints; bench.py owns stdout)
    g = cls(lr=lr, walk_type='linear', loss='l2', trainEmbed=False, attrList=list(attr_names), attrTable=table,
            layers=None, stylegan_opts=types.SimpleNamespace(latent='w'))
    np.random.set_state(state)
    if hasattr(g.walk, 'w'):                               # the linear walk; the MLP walks keep torch's own (seeded by the caller) init
        with torch.no_grad():
            g.walk.w.copy_(torch.from_numpy(synth.walk_init(len(attr_names), g.module.netG.n_latent, seed=walk_seed)))
    return g


def run_step(g, zs, alpha, no_content_loss=False, no_gan_loss=False, clamp=False, layers=None, optimize=True):
    """train.py:56-110 with an explicit alpha instead of the global-RNG draw.  Returns dict of device tensors."""
    from . import capture
    dev = g.device
    z = torch.Tensor(zs).to(dev)
    ag = torch.tensor(alpha).float().to(dev)
    if optimize:
        feed, r = capture.forward(g, z, ag, clamp=clamp, layers=layers, content=not no_content_loss)
        loss = g.optimizeParametersAll(feed, False, False, no_content_loss=no_content_loss, no_gan_loss=no_gan_loss)
        r.update(loss=loss.detach(), terms=g.last_terms)
    else:
        r = capture.forward_backward(g, z, ag, no_content_loss=no_content_loss, no_gan_loss=no_gan_loss, clamp=clamp, layers=layers)
    r['grad'] = g.walk.w.grad.detach().clone() if hasattr(g.walk, 'w') else None
    return r


def smoke():
    from . import synth
    from oracle import step as ostep                     # checker only
    assert torch.cuda.is_available(), 'smoke() needs the MI355X'
    import os
    import oracle
    torch.set_num_threads(min(32, oracle.host_cpus()))           # the CPU checker: as many threads as the host's quota really gives (oracle.host_cpus)
    torch.cuda.set_device(0)
    g = build_graph(64, ['Smiling'], 4)
    zs = synth.z_sample(4, seed=0)
    alpha = np.ones((4, 1)) * 0.19
    r = run_step(g, zs, alpha)
    torch.cuda.synchronize()
    dt = torch.float64
    nets = dict(G=ostep.to_torch(synth.generator_state(64, seed=100), dt), D=ostep.to_torch(synth.discriminator_state(64, seed=200), dt),
                R=ostep.to_torch(synth.resnet50_state(seed=300), dt), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400), dt))
    o = ostep.train_step(nets, torch.from_numpy(synth.walk_init(1, 10, seed=7)).to(dt), torch.from_numpy(zs), torch.from_numpy(alpha), [31])
    img_err = float((r['x1'].detach().double().cpu() - o['x1']).abs().max())
    loss_err = abs(float(r['loss']) - float(o['loss']))
    gerr = float((r['grad'].double().cpu() - o['grad']).abs().max() / o['grad'].abs().max())
    print('smoke: loss %.6f (oracle %.6f)  max|dimg| %.2e  walk-grad rel-to-max err %.2e' % (float(r['loss']), float(o['loss']), img_err, gerr))
    assert img_err < 1e-3 * float(o['x1'].abs().max()) + 1e-4, img_err
    assert loss_err < 1e-3 * abs(float(o['loss'])) + 1e-4, loss_err
    assert gerr < 1e-2, gerr
