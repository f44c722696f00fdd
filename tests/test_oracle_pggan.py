"""BASELINE config 1 (PGGAN 256^2 random-init generator, 1 attribute, batch 4, linear walk in z, CPU): the oracle's PGGAN
restatement against the fixture produced by the reference's own model_256.Generator and PGGAN TransformGraph methods."""
import json
import os
import zlib

import numpy as np
import torch

from latent2im_amd import synth
from oracle import nets, pggan
from oracle import step as ostep

HERE = os.path.dirname(os.path.abspath(__file__))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _state():
    layout = json.load(open(os.path.join(HERE, 'golden', 'pggan_layout.json')))
    st = {}
    for k, shape in layout:
        rs = np.random.RandomState(zlib.crc32(('PG.' + k).encode()) & 0x7FFFFFFF)
        st[k] = T((rs.randn(*shape) * (0.05 if len(shape) > 1 else 0.1)).astype(np.float32))
    return st


def test_pggan_generator_and_graph_plumbing(golden):
    g = golden('pggan')
    P = _state()
    z = T(synth.z_sample(4, seed=0)).float()
    img = pggan.generator_forward(P, z[:2, :511])
    assert list(img.shape) == list(g['shape']) == [2, 3, 256, 256]
    np.testing.assert_allclose(img.numpy().sum(3), g['img_rowsum'], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(img.numpy()[:, :, 100:116, 100:116], g['img_crop'], rtol=1e-3, atol=1e-4)
    # graph plumbing: half-resolution logits, clamp-variant alphas, z-walk, broadcasting BCE, `or` weighting
    PR = ostep.to_torch(synth.resnet50_state(seed=300))
    PV = ostep.to_torch(synth.vgg19_prefix_state(seed=400))
    walk = T(g['walk_w0']).requires_grad_(True)
    x0 = pggan.get_logits(P, z)
    assert list(x0.shape) == list(g['x0_shape']) == [4, 3, 128, 128]
    np.testing.assert_allclose(x0.numpy().sum(3), g['x0_rowsum'], rtol=1e-3, atol=2e-3)
    a0 = nets.resnet50_forward(PR, x0)[:, [31]]
    np.testing.assert_allclose(a0.numpy(), g['a0'], rtol=1e-3, atol=1e-4)
    target, eps = ostep.get_alphas_clamp(a0, torch.full((4, 1), 0.3))
    np.testing.assert_allclose(eps.numpy(), g['eps'], rtol=1e-3, atol=1e-4)
    z1 = pggan.walk_linear_z_free(z, eps, walk)
    np.testing.assert_allclose(z1.detach().numpy(), g['z1'], rtol=1e-4, atol=1e-5)
    x1 = pggan.get_logits(P, z1)
    reg = pggan.reg_loss_quirk(nets.resnet50_forward(PR, x1)[:, [31]], target)
    cont, terms = ostep.content_loss(PV, x0, x1)
    loss = pggan.total_loss(reg, cont, None, no_content_loss=False, no_gan_loss=True)
    np.testing.assert_allclose(reg.detach().numpy(), g['reg'], rtol=1e-4)
    np.testing.assert_allclose(torch.stack(terms).detach().numpy(), g['cont'], rtol=2e-3, atol=1e-9)
    np.testing.assert_allclose(loss.detach().numpy(), g['loss'], rtol=1e-4)
    grad, = torch.autograd.grad(loss, walk)
    assert np.abs(grad.numpy() - g['grad']).max() <= 1e-2 * np.abs(g['grad']).max()
