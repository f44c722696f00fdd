"""The segment tables of l2i_segmented_matvec_f32 (latent2im_amd.generator._ModPlan: every modulation / demodulation / ToRGB weight of a
generator pass in two launches, every d s and the whole latent gradient in two more) against the layer-by-layer formulas they replace
(reference networks.py:148-156, 231-239, 346-351), on the CPU: the kernel is replaced by a numpy execution of the SAME tables
(tests/emu.py:emulate_segmented_matvec), so offsets, pitches, pre / epilogue codes and the (segment, row block) grid are what is tested."""
import numpy as np
import pytest
import torch

from latent2im_amd import generator, kernels, synth
from tests import emu


@pytest.mark.parametrize('size,batch', [(32, 3), (128, 2)])
def test_modplan_tables_match_layerwise_formulas(monkeypatch, size, batch):
    G = generator.Generator(synth.generator_state(size, seed=100), size, device='cpu')
    plan = G.modplan
    monkeypatch.setattr(kernels, 'segmented_matvec', emu.emulate_segmented_matvec)
    rs = np.random.RandomState(size)
    B = batch
    lat = torch.from_numpy(rs.randn(B, G.n_latent, 512)).float()
    s_all, d_all, w_all = plan.forward(lat)
    conv_idx, rgb_idx = G.latent_index()
    for li, L in enumerate(G.layers):
        s = torch.addmm(L.mod.b, lat[:, conv_idx[li]], L.mod.A.t())
        demod = torch.rsqrt(torch.mm(s * s, L.T.t()) + 1e-8)
        np.testing.assert_allclose(plan.s(s_all, B, li).numpy(), s.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(plan.demod(d_all, B, li).numpy(), demod.numpy(), rtol=1e-5, atol=1e-6)
    for j, R in enumerate(G.rgbs):
        srgb = torch.addmm(R.mod.b, lat[:, rgb_idx[j]], R.mod.A.t())
        wmod = R.W.unsqueeze(0) * srgb.unsqueeze(1)
        np.testing.assert_allclose(plan.wmod(w_all, B, j).numpy(), wmod.numpy(), rtol=1e-5, atol=1e-6)
    # backward: random per-layer reductions -> the latent gradient, against the per-layer accumulation it replaces
    red_dz, q_all, red_rgb = plan.reductions(B, 'cpu')
    red_dz.copy_(torch.from_numpy(rs.randn(red_dz.numel())).float())
    q_all.copy_(torch.from_numpy(rs.randn(q_all.numel())).float())
    red_rgb.copy_(torch.from_numpy(rs.randn(red_rgb.numel())).float())
    g = plan.backward(B, s_all, d_all, red_dz, q_all, red_rgb)
    want = torch.zeros(B, G.n_latent, 512)
    for li, L in enumerate(G.layers):
        s, demod = plan.s(s_all, B, li), plan.demod(d_all, B, li)
        d_demod = plan.demod(red_dz, B, li) / demod
        d_s = plan.s(q_all, B, li) - s * torch.mm(d_demod * demod * demod * demod, L.T)
        want[:, conv_idx[li]] += torch.mm(d_s, L.mod.A)
    for j, R in enumerate(G.rgbs):
        d_srgb = (plan.red_rgb(red_rgb, B, j) * R.W.t().unsqueeze(0)).sum(2)
        want[:, rgb_idx[j]] += torch.mm(d_srgb, R.mod.A)
    scale = float(want.abs().max())
    assert float((g - want).abs().max()) < 1e-5 * scale
