"""The oracle (oracle/) against the fixtures captured from the reference's own Python (tests/golden/make_golden.py).
CPU only.  This is what pins the oracle (prompt item 3 / SURVEY 8c)."""
import json
import os

import numpy as np
import pytest
import torch

from latent2im_amd import specs, synth
from oracle import sg2, step as ostep

HERE = os.path.dirname(os.path.abspath(__file__))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_state_dict_layout_matches_reference():
    lay = json.load(open(os.path.join(HERE, 'golden', 'state_dict_layout.json')))
    for size in (32, 64, 256, 1024):
        assert [[k, list(v)] for k, v in specs.generator_spec(size).items()] == lay['G%d' % size]
        assert [[k, list(v)] for k, v in specs.discriminator_spec(size).items()] == lay['D%d' % size]


def test_fused_bias_act_table(golden):
    g = golden('fused_bias_act')
    x, b, ref = T(g['x']), T(g['b']), T(g['ref'])
    e = torch.zeros(0)
    for act in (1, 3):
        for grad in (0, 1, 2):
            y = sg2.fused_bias_act(x, b if grad == 0 else e, ref if grad == 1 else e, act, grad, 0.2, 2 ** 0.5)
            close(y, g['y_%d%d' % (act, grad)], 1e-6, 1e-7)
    close(sg2.fused_leaky_relu(T(g['x2']), T(g['b2'])), g['y2_30'], 1e-6, 1e-7)
    gi, gb = sg2.fused_leaky_relu_backward(x, ref)
    close(gi, g['y_31'], 1e-6, 1e-7)
    close(gb, g['y_31'].sum((0, 2, 3)), 1e-5, 1e-6)


def test_upfirdn2d_cases(golden):
    g = golden('upfirdn2d')
    for i, (n, c, h, w, up, down, p0, p1, gain) in enumerate(g['cases']):
        up, down, p0, p1 = int(up), int(down), int(p0), int(p1)
        rs = np.random.RandomState(20 + i)
        x = rs.randn(int(n), int(c), int(h), int(w)).astype(np.float32)
        k = T(synth.fir_kernel(gain=gain))
        y = sg2.upfirdn2d(T(x), k, up, down, (p0, p1))
        assert y.shape[2] == sg2.upfirdn2d_out_size(int(h), up, down, p0, p1, 4)
        close(y, g['y_%d' % i], 1e-5, 1e-6)
        if h == w:
            gx = sg2.upfirdn2d_backward(T(g['gy_%d' % i]), k, up, down, (p0, p1), (int(h), int(w)))
            close(gx, g['gx_%d' % i], 1e-5, 1e-6)


@pytest.mark.parametrize('name,demod,up', [('same', True, False), ('up', True, True), ('rgb', False, False),
                                           ('same512', True, False), ('up512', True, True)])
def test_modulated_conv(golden, name, demod, up):
    g = golden('modconv')
    P = {k[len(name) + 3:]: T(g[k]) for k in g.files if k.startswith(name + '.P.')}
    P = {'m.' + k: v for k, v in P.items()}
    x, w = T(g[name + '.x']).requires_grad_(True), T(g[name + '.w']).requires_grad_(True)
    y = sg2.modulated_conv2d(P, 'm', x, w, demodulate=demod, upsample=up)
    close(y, g[name + '.y'], 2e-4, 2e-5)
    gx, gw = torch.autograd.grad(y, [x, w], T(g[name + '.gy']))
    close(gx, g[name + '.gx'], 2e-4, 2e-5)
    close(gw, g[name + '.gw'], 2e-4, 2e-4)


def test_styled_conv_and_to_rgb(golden):
    g = golden('modconv')
    P = {k[len('styled.P.'):]: T(g[k]) for k in g.files if k.startswith('styled.P.')}
    P = {'s.' + k: v for k, v in P.items()}
    y = sg2.styled_conv(P, 's', T(g['styled.x']), T(g['styled.w']), T(g['styled.noise']), upsample=True)
    close(y, g['styled.y'], 2e-4, 2e-5)
    P = {'t.' + k[len('torgb.P.'):]: T(g[k]) for k in g.files if k.startswith('torgb.P.')}
    x, w, sk = (T(g['torgb.' + n]).requires_grad_(True) for n in ('x', 'w', 'skip'))
    y = sg2.to_rgb(P, 't', x, w, sk)
    close(y, g['torgb.y'], 2e-4, 2e-5)
    gx, gw, gs = torch.autograd.grad(y, [x, w, sk], T(g['torgb.gy']))
    close(gx, g['torgb.gx'], 2e-4, 2e-5)
    close(gw, g['torgb.gw'], 2e-4, 2e-4)
    close(gs, g['torgb.gskip'], 2e-4, 2e-5)


@pytest.mark.parametrize('size,batch', [(32, 4), (64, 4), (256, 2)])
def test_generator_and_discriminator(golden, size, batch):
    g = golden('generator')
    PG = ostep.to_torch(synth.generator_state(size, seed=100, noise_strength=0.5))
    z = T(synth.z_sample(batch, seed=0)).float()
    w = sg2.style_mlp(PG, z)
    close(w, g['w_%d' % size], 1e-4, 1e-5)
    n_latent = 2 * int(np.log2(size)) - 2
    lat = torch.stack([w * (1.0 + 0.05 * i) for i in range(n_latent)], 1)
    noise = [T(n) for n in synth.noise_maps(size, batch)]
    img = sg2.generator_synthesis(PG, lat, noise)
    if size <= 64:
        close(img, g['img_%d' % size], 1e-3, 1e-4)
        img0 = sg2.generator_synthesis(PG, torch.stack([w] * n_latent, 1), None)
        close(img0, g['img0_%d' % size], 1e-3, 1e-4)
    else:
        a = img.numpy()
        close(a[:, :, 96:128, 112:144], g['img_256_crop'], 1e-3, 1e-4)
        idx = g['img_256_probe_idx']
        close(a[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]], g['img_256_probe'], 1e-3, 1e-4)
        close(a.sum(3), g['img_256_rowsum'], 1e-3, 2e-3)
    PD = ostep.to_torch(synth.discriminator_state(size, seed=200))
    imgs = img if batch >= 4 else torch.cat([img, img.flip(3)])
    close(sg2.discriminator_forward(PD, imgs), g['d_%d' % size], 1e-3, 1e-4)


def _nets64(dt=torch.float32):
    return dict(G=ostep.to_torch(synth.generator_state(64, seed=100), dt), D=ostep.to_torch(synth.discriminator_state(64, seed=200), dt),
                R=ostep.to_torch(synth.resnet50_state(seed=300), dt), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400), dt))


def grad_close(a, b, frac):
    """Walk gradients are compared relative to the largest entry: the network is piecewise linear (ReLU /
    leaky-ReLU / max-pool masks), so float32 rounding flips masks and the REFERENCE's own float32 gradient sits
    3.1e-3 * max|g| away from its float64 evaluation (fixture single64 vs single.0)."""
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - b).max() / np.abs(b).max()
    assert err <= frac, err


def test_training_step_float64_pins_semantics(golden):
    """Oracle and reference, both evaluated in float64, agree to rounding: identical algorithm."""
    g = golden('step')
    nets = _nets64(torch.float64)
    zs = synth.z_sample(12, seed=0)
    r = ostep.train_step(nets, T(synth.walk_init(1, 10, seed=7)).double(), T(zs[0:4]), T(g['single.alphas'][0]), [31])
    close(r['loss'], g['single64.loss'], 1e-12, 0)
    close(r['alpha_org'], g['single64.a0'], 1e-11, 0)
    close(torch.stack(r['cont_terms']), g['single64.cont'], 1e-10, 0)
    grad_close(r['grad'], g['single64.grad'], 1e-10)
    r = ostep.train_step(nets, T(synth.walk_init(5, 10, seed=7)).double(), T(zs[0:4]), T(g['multi.delta']),
                         [31, 39, 20, 15, 5], clamp_variant=True)
    close(r['loss'], g['multi64.loss'], 1e-12, 0)
    close(r['eps'], g['multi64.eps'], 1e-10, 1e-14)
    grad_close(r['grad'], g['multi64.grad'], 1e-10)
    # how far the reference's own float32 run is from this value (documented in DESIGN.md)
    ref32 = np.abs(g['single.0.grad'] - g['single64.grad']).max() / np.abs(g['single64.grad']).max()
    assert 1e-4 < ref32 < 1e-2


def test_training_steps_single_attr(golden):
    """Three consecutive reference optimizeParametersAll steps (train.py flow): loss terms and walk gradient from
    the reference's own walk trajectory; the Adam restatement replayed on the reference's gradients."""
    g = golden('step')
    nets = _nets64()
    walk = T(synth.walk_init(1, 10, seed=7))
    opt = ostep.Adam(walk.clone(), lr=1e-3)
    zs = synth.z_sample(12, seed=0)
    assert str(g['single.loss_dtype']) == 'torch.float64'
    for i in range(3):
        r = ostep.train_step(nets, walk, T(zs[4 * i:4 * i + 4]).float(), T(g['single.alphas'][i]).float(), [31])
        assert r['loss'].dtype == torch.float64
        close(r['alpha_org'], g['single.%d.a0' % i], 1e-3, 1e-4)
        close(r['reg'], g['single.%d.reg' % i], 1e-4, 1e-6)
        close(torch.stack(r['cont_terms']), g['single.%d.cont' % i], 1e-3, 1e-6)
        close(r['gan'], g['single.%d.gan' % i], 1e-3, 1e-5)
        close(r['loss'], g['single.%d.loss' % i], 1e-4, 1e-6)
        grad_close(r['grad'], g['single.%d.grad' % i], 1e-2)
        close(r['x1'].sum(3), g['single.%d.x1_rowsum' % i], 1e-3, 2e-3)
        opt.step(T(g['single.%d.grad' % i]))                    # Adam on the reference's gradient sequence
        close(opt.p, g['single.%d.walk' % i], 1e-5, 1e-7)
        walk = T(g['single.%d.walk' % i])


def test_training_step_regonly_layers_and_multi_attr(golden):
    g = golden('step')
    nets = _nets64()
    zs = synth.z_sample(12, seed=0)
    r = ostep.train_step(nets, T(synth.walk_init(1, 10, seed=7)), T(zs[0:4]).float(), T(g['single.alphas'][0]).float(),
                         [31], layers=[0, 1, 2, 3, 4, 5], no_content_loss=True, no_gan_loss=True)
    close(r['loss'], g['regonly.loss'], 1e-4, 1e-6)
    grad_close(r['grad'], g['regonly.grad'], 1e-2)
    assert np.all(r['grad'].numpy()[:, 6:] == 0)
    idx = [int(i) for i in g['multi.attr_idx']]
    assert idx == [31, 39, 20, 15, 5]
    r = ostep.train_step(nets, T(synth.walk_init(5, 10, seed=7)), T(zs[0:4]).float(), T(g['multi.delta']).float(),
                         idx, clamp_variant=True)
    close(r['alpha_org'], g['multi.a0'], 1e-3, 1e-4)
    close(r['eps'], g['multi.eps'], 1e-3, 1e-4)
    close(r['loss'], g['multi.loss'], 1e-4, 1e-6)
    grad_close(r['grad'], g['multi.grad'], 1e-2)


def test_adam_restatement_matches_torch_optim():
    rs = np.random.RandomState(0)
    p0 = T(rs.randn(3, 4, 5).astype(np.float32))
    p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([p], lr=1e-2, betas=(0.5, 0.99))
    mine = ostep.Adam(p0.clone(), lr=1e-2)
    for i in range(5):
        gr = T(rs.randn(3, 4, 5).astype(np.float32))
        p.grad = gr.clone()
        ref.step()
        mine.step(gr)
        close(mine.p, p.detach().numpy(), 1e-5, 1e-6)


@pytest.mark.parametrize('batch,attrs,clamp,flags', [(8, 1, False, {}), (4, 2, True, {}), (8, 1, False, dict(no_content_loss=True, no_gan_loss=True))])
def test_bounded_memory_step_is_the_same_function(batch, attrs, clamp, flags):
    """oracle.step.train_step_bounded (one minibatch-stddev subgroup and one loss branch at a time: what the 1024^2 batch-8 GPU
    parity test can afford) against the plain train_step on a batch that has two subgroups."""
    size = 32
    nets = dict(G=ostep.to_torch(synth.generator_state(size, seed=100)), D=ostep.to_torch(synth.discriminator_state(size, seed=200)),
                R=ostep.to_torch(synth.resnet50_state(seed=300)), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400)))
    assert ostep.stddev_subgroups(8) == [[0, 2, 4, 6], [1, 3, 5, 7]] and ostep.stddev_subgroups(2) == [[0, 1]]
    assert ostep.stddev_subgroups(16)[1] == [1, 5, 9, 13]
    walk = T(synth.walk_init(attrs, 8, seed=7))
    zs = T(synth.z_sample(batch, seed=5)).float()
    alpha = torch.full((batch, attrs), 0.3) * torch.arange(1, attrs + 1)
    idx = [31, 39][:attrs]
    a = ostep.train_step(nets, walk, zs, alpha, idx, clamp_variant=clamp, **flags)
    b = ostep.train_step_bounded(nets, walk, zs, alpha, idx, clamp_variant=clamp, **flags)
    for k in ('x0', 'x1', 'alpha_org', 'eps', 'target', 'reg', 'cont', 'gan', 'loss', 'grad'):
        assert a[k].dtype == b[k].dtype, k
        np.testing.assert_allclose(b[k].numpy(), a[k].numpy(), rtol=1e-5, atol=1e-6 * float(a[k].abs().max()), err_msg=k)


def test_storage_rounding_oracle_of_the_16bit_resnet():
    """oracle/nets16.py (the float64 ResNet-50 with the 16-bit path's storage rounding restated): close to the exact network in value
    (bf16 feature maps: a few 1e-4 of the output), different in gradient (masks flip: the price of the format), and chaotic at the 1e-7
    level — the floor tests/test_h8_gpu.py holds the kernels to must be a real, non-zero number."""
    import numpy as np
    import torch
    from latent2im_amd import synth
    from oracle import nets, nets16, step as ostep
    P = ostep.to_torch(synth.resnet50_state(seed=300), torch.float64)
    rs = np.random.RandomState(3)
    x = torch.tensor(rs.randn(1, 3, 64, 64) * 0.5)
    gy = torch.tensor(rs.randn(1, 40))

    def grad(f):
        xg = x.clone().requires_grad_(True)
        y = f(xg)
        g, = torch.autograd.grad(y, xg, gy)
        return y.detach(), g
    y0, g0 = grad(lambda t: nets.resnet50_forward(P, t))
    y1, g1 = grad(lambda t: nets16.resnet50_forward_bf16(P, t))
    torch.manual_seed(0)
    y2, g2 = grad(lambda t: nets16.resnet50_forward_bf16(P, t, perturb=1e-7))
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    assert float((y1 - y0).abs().max() / y0.abs().max()) < 5e-3
    assert 0.9 < cos(g1, g0) < 0.9995                       # the format moves the gradient ...
    assert 0.97 < cos(g2, g1) < 0.99999                     # ... and so does an fp32-sized perturbation before the rounding (the chaos floor)
    assert torch.equal(nets16.q(torch.tensor([1.0, 1.00390625, 3.0e38], dtype=torch.float64)),
                       torch.tensor([1.0, 1.0, 3.0e38], dtype=torch.float64).float().to(torch.bfloat16).double())
