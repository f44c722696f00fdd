"""BASELINE config 1 on the product side (latent2im_amd/pggan.py on the HIP kernels, through the C ABI): the PGGAN-256 generator and the
PGGAN graph's z-walk training step against the fixture produced by the reference's own model_256.Generator / PGGAN TransformGraph methods
(tests/golden/pggan.npz, make_golden.py::gen_pggan) and against the CPU oracle (oracle/pggan.py)."""
import json
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from latent2im_amd import constants, synth
from latent2im_amd import kernels as K
from latent2im_amd import pggan as pg
from latent2im_amd.perceptual import VGG19Prefix
from latent2im_amd.regressor import ResNet50
from oracle import pggan as opg

pytestmark = pytest.mark.gpu
DEV = 'cuda'
HERE = os.path.dirname(os.path.abspath(__file__))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def fixture_state():
    """The weights the fixture was generated with (make_golden.py::pggan_state): RandomState(crc32('PG.' + name)) per tensor."""
    st = {}
    for k, shape in json.load(open(os.path.join(HERE, 'golden', 'pggan_layout.json'))):
        rs = np.random.RandomState(zlib.crc32(('PG.' + k).encode()) & 0x7FFFFFFF)
        st[k] = (rs.randn(*shape) * (0.05 if len(shape) > 1 else 0.1)).astype(np.float32)
    return st


@pytest.mark.parametrize('shape,slope', [((3, 511), 1.0), ((2, 512, 4, 4), 0.2), ((2, 128, 64, 64), 0.2), ((2, 37, 6, 6), 0.2), ((1, 8, 3, 5), 0.2)])
def test_pixelnorm_act_forward_backward(shape, slope):
    """l2i_pixelnorm_act_f32 / _bwd_f32 against model_256.PixelNorm + LeakyReLU evaluated by torch in float64."""
    torch.manual_seed(0)
    x = torch.randn(*shape)
    gy = torch.randn(*shape)
    xd = x.double().requires_grad_(True)
    n = xd / torch.sqrt(torch.mean(xd ** 2, dim=1, keepdim=True) + 1e-8)
    y = F.leaky_relu(n, slope) if slope != 1.0 else n
    y.backward(gy.double())
    got = K.pixelnorm_act(x.to(DEV), slope=slope)
    np.testing.assert_allclose(got.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-6)
    gx = K.pixelnorm_act_bwd(gy.to(DEV), x.to(DEV), slope=slope)
    np.testing.assert_allclose(gx.cpu().numpy(), xd.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('shape', [(2, 3, 4, 4), (1, 5, 6, 10), (2, 16, 128, 128)])
def test_upsample_nearest_and_pool_are_exact_and_adjoint(shape):
    torch.manual_seed(1)
    x = torch.randn(*shape)
    up = K.upsample2x_nearest(x.to(DEV))
    assert torch.equal(up.cpu(), F.interpolate(x, scale_factor=2, mode='nearest'))
    big = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3])
    half = K.pool2x2(big.to(DEV), 0.25).cpu()
    np.testing.assert_allclose(half.numpy(), F.interpolate(big, size=(shape[2], shape[3]), mode='bilinear', align_corners=False).numpy(), rtol=1e-6, atol=1e-6)
    assert torch.equal(K.pool2x2(up, 0.25).cpu(), x)                       # four equal values: 0.25 * ((v + v) + (v + v)) == v
    # <up(x), big> == <x, pool_sum(big)>
    lhs = float((up.cpu().double() * big.double()).sum())
    rhs = float((x.double() * K.pool2x2(big.to(DEV), 1.0).cpu().double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))
    with pytest.raises(RuntimeError):
        K.upsample2x_nearest(torch.randn(1, 1, 3, 3, device=DEV))         # odd width: refused, not silently wrong


def test_pggan_generator_matches_reference_fixture_and_oracle(golden):
    g = golden('pggan')
    st = fixture_state()
    G = pg.Generator(st, device=DEV)
    z = T(synth.z_sample(4, seed=0)).float()
    img = G(z[:2, :511].to(DEV), step=6, alpha=0)
    assert list(img.shape) == list(g['shape']) == [2, 3, 256, 256]
    np.testing.assert_allclose(img.cpu().numpy().sum(3), g['img_rowsum'], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(img.cpu().numpy()[:, :, 100:116, 100:116], g['img_crop'], rtol=1e-3, atol=1e-4)
    # other (step, alpha) settings of model_256.Generator.forward against the oracle: a blended last block, a plain to_rgb, the 4x4 stage
    P = {k: T(v) for k, v in st.items()}
    for step, alpha in ((3, 0.0), (2, 0.4), (4, -1), (0, 0.0), (1, 1.0)):
        want = opg.generator_forward(P, z[:2, :511], step=step, alpha=alpha)
        got = G(z[:2, :511].to(DEV), step=step, alpha=alpha)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-3, atol=1e-4, err_msg='step %d alpha %g' % (step, alpha))
    with pytest.raises(RuntimeError):
        G(z[:2].to(DEV))                                                   # 512 columns: the in-repo generator takes 511 (+ label)


def test_pggan_generator_latent_gradient_vs_float64_oracle():
    """d(sum(img * probe)) / dz through the hand-scheduled backward (PixelNorm backward, nearest-upsample adjoint, dgrad convs) against
    autograd on the float64 oracle, for the graph's setting and for a blended last block."""
    st = fixture_state()
    G = pg.Generator(st, device=DEV)
    P = {k: T(v).double() for k, v in st.items()}
    z = T(synth.z_sample(2, seed=3)).float()[:, :511]
    for step, alpha in ((3, 0.0), (2, 0.4), (3, -1)):
        zd = z.double().requires_grad_(True)
        want_img = opg.generator_forward(P, zd, step=step, alpha=alpha)
        probe = torch.randn(want_img.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
        (want_img * probe).sum().backward()
        zg = z.to(DEV).requires_grad_(True)
        img = G(zg, step=step, alpha=alpha)
        (img * probe.float().to(DEV)).sum().backward()
        err = float((zg.grad.cpu().double() - zd.grad).abs().max() / zd.grad.abs().max())
        assert err < 5e-3, (step, alpha, err)                              # leaky-ReLU masks flip under float32 rounding (test_networks_gpu.grad_ok)


def _graph(g, state=None):
    constants.BATCH_SIZE = 4
    nets = (pg.Generator(state if state is not None else fixture_state(), device=DEV), ResNet50(synth.resnet50_state(seed=300), device=DEV),
            VGG19Prefix(synth.vgg19_prefix_state(seed=400), device=DEV), {'G': 'fixture'})
    state = np.random.get_state()
    graph = pg.faceGraph(lr=1e-3, walk_type='linear', loss='l2', trainEmbed=False, attrList=['Smiling'], attrTable={'Smiling': 31}, layers=None,
                         pgan_opts=None, nets=nets)
    np.random.set_state(state)
    with torch.no_grad():
        graph.walk.w.copy_(T(g['walk_w0']))
    return graph


def test_pggan_graph_step_matches_reference_fixture(golden):
    """The whole config-1 step through the product graph: half-resolution logits, regressor column, clamp-pair alphas, z walk, the
    broadcasting BCE in float64, content loss, `or` weight rule, walk gradient, Adam step."""
    g = golden('pggan')
    graph = _graph(g)
    z = T(synth.z_sample(4, seed=0)).float().to(DEV)
    x0 = graph.get_logits({'z': z})
    assert list(x0.shape) == list(g['x0_shape']) == [4, 3, 128, 128]
    np.testing.assert_allclose(x0.cpu().numpy().sum(3), g['x0_rowsum'], rtol=1e-3, atol=2e-3)
    a0 = graph.get_reg_preds(x0)
    np.testing.assert_allclose(a0.cpu().numpy(), g['a0'], rtol=1e-3, atol=1e-4)
    target, eps = graph.get_alphas(a0, torch.full((4, 1), 0.3, device=DEV))
    np.testing.assert_allclose(target.cpu().numpy(), g['target'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(eps.cpu().numpy(), g['eps'], rtol=1e-3, atol=1e-4)
    z1 = graph.get_z_new_tensor(z, eps)
    np.testing.assert_allclose(z1.detach().cpu().numpy(), g['z1'], rtol=1e-4, atol=1e-5)
    x1 = graph.get_logits({'z': z1})
    np.testing.assert_allclose(x1.detach().cpu().numpy().sum(3), g['x1_rowsum'], rtol=1e-3, atol=2e-3)
    feed = {'z': z1, 'org': x0, 'logit': x1, 'alpha': target}
    w_before = graph.walk.w.detach().clone()
    loss = graph.optimizeParametersAll(feed, False, False, no_content_loss=False, no_gan_loss=True)
    assert loss.dtype == torch.float64
    np.testing.assert_allclose(float(graph.last_terms['reg']), g['reg'], rtol=1e-4)
    np.testing.assert_allclose(float(loss.detach()), g['loss'], rtol=1e-4)
    # z1 - z ~ 1e-3: the four content terms are ~1e-12, i.e. float32 rounding of two nearly equal images (tests/test_oracle_pggan.py: atol 1e-9)
    np.testing.assert_allclose(float(graph.last_terms['cont']), g['cont'].mean(), rtol=2e-3, atol=1e-9)
    grad = graph.walk.w.grad.cpu().numpy()
    # the fixture's walk is N(0, 0.02): x1 - x0 is at float32 rounding level, so the content term's share of this gradient is rounding noise
    # of whoever evaluates it (max|g| = 7e-10).  Direction and scale are pinned here; test_pggan_graph_gradient_strong_walk_vs_float64_oracle
    # holds the gradient tightly where it carries signal.
    gm = np.abs(g['grad']).max()
    assert np.abs(grad - g['grad']).max() <= 0.1 * gm and np.median(np.abs(grad - g['grad'])) <= 1e-2 * gm
    step = (graph.walk.w.detach() - w_before).abs().max()
    assert 0 < float(step) < 1.5e-3                                        # Adam moved the walk (|g| ~ 1e-10 is below Adam's eps: the step is lr * g / eps)
    with pytest.raises(TypeError):
        graph.optimizeParametersAll(feed, False, False, no_gan_loss=False)  # the GAN term cannot be evaluated on the in-repo discriminator
    # the same sequence as one call (train_multi_attr.py order), second step: the loss stays finite and float64
    l2, *_ = pg.walk_training_step(graph, synth.z_sample(4, seed=1), np.full((4, 1), -0.2), no_content_loss=True)
    assert l2.dtype == torch.float64 and np.isfinite(float(l2.detach()))
    # vis path: apply_alpha returns (edited image, alpha_org)
    out, aorg = graph.apply_alpha({'z': synth.z_sample(4, seed=0)}, np.full((4, 1), 0.8))
    assert list(out.shape) == [4, 3, 128, 128] and list(aorg.shape) == [4, 1]
    np.testing.assert_allclose(aorg.cpu().numpy(), g['a0'], rtol=1e-3, atol=1e-4)


def test_pggan_graph_gradient_strong_walk_vs_float64_oracle(golden):
    """Same step where every loss term carries signal — a generator initialised like the reference's constructor (N(0,1) weight_orig,
    synth.pggan_generator_state: its image depends on z, the fixture's barely does) and a walk 50x larger (z1 - z = O(0.3 z)) — against
    the float64 oracle (oracle/pggan.py + oracle/nets.py + oracle/step.py): images, loss terms, walk gradient."""
    from oracle import nets as onets
    from oracle import step as ostep
    g = golden('pggan')
    st = synth.pggan_generator_state(seed=11)
    graph = _graph(g, st)
    w0 = T(g['walk_w0']) * 50.0
    with torch.no_grad():
        graph.walk.w.copy_(w0)
    z = T(synth.z_sample(4, seed=0)).float()
    zd = z.to(DEV)
    x0 = graph.get_logits({'z': zd})
    target, eps = graph.get_alphas(graph.get_reg_preds(x0), torch.full((4, 1), 0.3, device=DEV))
    x1 = graph.get_logits({'z': graph.get_z_new_tensor(zd, eps)})
    loss = graph.get_w_loss({'org': x0, 'logit': x1, 'alpha': target}, no_content_loss=False, no_gan_loss=True)
    loss.backward()
    dt = torch.float64
    P = {k: T(v).to(dt) for k, v in st.items()}
    PR, PV = ostep.to_torch(synth.resnet50_state(seed=300), dt), ostep.to_torch(synth.vgg19_prefix_state(seed=400), dt)
    walk = w0.to(dt).requires_grad_(True)
    ox0 = opg.get_logits(P, z.to(dt))
    otarget, oeps = ostep.get_alphas_clamp(onets.resnet50_forward(PR, ox0)[:, [31]], torch.full((4, 1), 0.3, dtype=dt))
    ox1 = opg.get_logits(P, opg.walk_linear_z_free(z.to(dt), oeps, walk))
    oreg = opg.reg_loss_quirk(onets.resnet50_forward(PR, ox1)[:, [31]], otarget)
    ocont, _ = ostep.content_loss(PV, ox0, ox1)
    oloss = opg.total_loss(oreg, ocont, None, no_content_loss=False, no_gan_loss=True)
    ograd, = torch.autograd.grad(oloss, walk)
    assert float((ox1 - ox0).detach().abs().max()) > 0.05 * float(ox0.abs().max())           # the edit is visible: this test has signal
    np.testing.assert_allclose(x1.detach().cpu().numpy(), ox1.detach().numpy(), rtol=1e-3, atol=1e-4 * float(ox1.detach().abs().max()))
    np.testing.assert_allclose(float(graph.last_terms['reg'].detach()), float(oreg), rtol=1e-4)
    np.testing.assert_allclose(float(graph.last_terms['cont'].detach()), float(ocont), rtol=1e-3)
    np.testing.assert_allclose(float(loss.detach()), float(oloss), rtol=1e-4)
    err = float((graph.walk.w.grad.cpu().double() - ograd).abs().max() / ograd.abs().max())
    assert err < 5e-3, err


def test_pggan_graph_is_found_by_the_plugin_lookup_and_checkpoints_round_trip(golden, tmp_path):
    from graphs import find_model_using_name
    cls = find_model_using_name('pggan', 'face')
    assert cls is pg.faceGraph and find_model_using_name('pggan', 'scene') is pg.SceneGraph
    graph = _graph(golden('pggan'))
    graph.save_multi_models(str(tmp_path / 'model_w_0'), str(tmp_path / 'gan'))
    w = graph.walk.w.detach().clone()
    with torch.no_grad():
        graph.walk.w.zero_()
    graph.load_multi_models(str(tmp_path / 'model_w_0_walk_module.ckpt'), None)
    assert type(graph.walk).__module__ == 'graphs.pggan.transform_base' and torch.equal(graph.walk.w.detach(), w)
    # vis path (vis_w.py flow): alphas from vis_image_batch, one PNG strip per sample
    zs = synth.z_sample(2, seed=4)
    a2g, a2t = graph.vis_image_batch({'z': zs}, None, 0, num_panels=3)
    paths = graph.vis_multi_image_batch_alphas({'z': zs}, str(tmp_path / 'vis'), a2g, a2t, 0)
    from PIL import Image
    assert len(paths) == 2 and Image.open(paths[0]).size == (3 * 128, 128)
