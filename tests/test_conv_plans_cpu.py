"""Host-side launch plans of latent2im_amd.conv (packing, stride-2 phases, input-gradient plans) checked on CPU through
the semantic emulator of the kernel (tests/emu.py) against torch's own conv / conv_transpose / autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from latent2im_amd import conv
from tests import emu


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    monkeypatch.setattr(conv, 'run_launch', emu.emulate_launch)
    monkeypatch.setattr(conv, 'run_fused_transposed', emu.emulate_fused_transposed)


CASES = [  # (cin, cout, k, stride, pad, transposed, h, w)
    (5, 7, 3, 1, 1, False, 9, 9), (5, 7, 1, 1, 0, False, 6, 8), (4, 6, 3, 2, 1, False, 12, 12), (4, 6, 3, 2, 1, False, 11, 13),
    (4, 6, 1, 2, 0, False, 8, 8), (3, 8, 7, 2, 3, False, 20, 20), (4, 6, 3, 2, 0, False, 9, 9), (4, 6, 1, 2, 0, False, 7, 7),
    (6, 4, 3, 2, 0, True, 5, 5), (6, 4, 3, 2, 0, True, 4, 6), (33, 40, 3, 1, 1, False, 5, 5),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,tr,h,w', CASES)
def test_forward_and_dgrad_plans(cin, cout, k, stride, pad, tr, h, w):
    rs = np.random.RandomState(cin * 100 + cout + k)
    wt = torch.from_numpy(rs.randn(cout, cin, k, k)).float()
    x = torch.from_numpy(rs.randn(2, cin, h, w)).float().requires_grad_(True)
    fc = conv.FrozenConv2d(wt, stride, pad, transposed=tr, device='cpu')
    if tr:
        ref = F.conv_transpose2d(x, wt.transpose(0, 1), stride=2, padding=pad)
    else:
        ref = F.conv2d(x, wt, stride=stride, padding=pad)
    y = fc.forward(x.detach())
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    gy = torch.from_numpy(rs.randn(*ref.shape)).float()
    gref, = torch.autograd.grad(ref, x, gy)
    gx = fc.dgrad(gy, (h, w))
    np.testing.assert_allclose(gx.numpy(), gref.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('cin,cout,k,pad,tr,h', [(6, 4, 3, 0, True, 5), (6, 6, 3, 1, False, 12), (6, 6, 3, 1, False, 11), (3, 8, 7, 3, False, 20),
                                                 (5, 6, 3, 0, False, 9), (8, 5, 7, 3, False, 14), (8, 4, 3, 0, True, 5), (6, 8, 3, 1, False, 12)])
def test_fused_and_phase_transposed_agree(monkeypatch, fused, cin, cout, k, pad, tr, h):
    monkeypatch.setattr(conv, 'USE_FUSED_TRANSPOSED', fused)
    rs = np.random.RandomState(k + h)
    wt = torch.from_numpy(rs.randn(cout, cin, k, k)).float()
    x = torch.from_numpy(rs.randn(2, cin, h, h)).float().requires_grad_(True)
    fc = conv.FrozenConv2d(wt, 2, pad, transposed=tr, device='cpu')
    ref = F.conv_transpose2d(x, wt.transpose(0, 1), stride=2, padding=pad) if tr else F.conv2d(x, wt, stride=2, padding=pad)
    gy = torch.from_numpy(rs.randn(*ref.shape)).float()
    gref, = torch.autograd.grad(ref, x, gy)
    np.testing.assert_allclose(fc.forward(x.detach()).numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(fc.dgrad(gy, (h, h)).numpy(), gref.numpy(), rtol=1e-4, atol=1e-4)
    # the fused kernel walks its input channels four at a time (K = 3) and is not used for <= 4 output channels
    assert (fc.fwd_fused is not None) == (tr and cin % 4 == 0)
    assert (fc.bwd_fused is not None) == (not tr and cin > 4 and cout % 4 == 0)


def test_pack_weight_layout():
    w = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).reshape(2, 3, 2, 2)
    p = conv.pack_weight(w)
    assert p.shape == (3, 4, 32)
    assert p[1, 2, 1] == w[1, 1, 1, 0] and p[:, :, 2:].abs().sum() == 0


def test_epilogue_and_prologue_semantics():
    rs = np.random.RandomState(3)
    wt = torch.from_numpy(rs.randn(6, 4, 3, 3)).float()
    x = torch.from_numpy(rs.randn(2, 4, 8, 8)).float()
    s = torch.from_numpy(rs.rand(2, 4)).float() + 0.5
    d = torch.from_numpy(rs.rand(2, 6)).float() + 0.5
    nz = torch.from_numpy(rs.randn(2, 1, 8, 8)).float()
    b = torch.from_numpy(rs.randn(6)).float()
    fc = conv.FrozenConv2d(wt, 1, 1, device='cpu')
    y = fc.forward(x, in_scale=s, out_scale=d, noise=nz, noise_w=0.3, bias=b, act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    ref = F.conv2d(x * s[:, :, None, None], wt, padding=1) * d[:, :, None, None] + 0.3 * nz + b[None, :, None, None]
    ref = F.leaky_relu(ref, 0.2) * 2 ** 0.5
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('cin,cout,h,w,pad', [(8, 5, 6, 8, 1), (16, 40, 7, 9, 1), (8, 32, 8, 8, 0)])
def test_winograd_pack_layout_and_transform(cin, cout, h, w, pad):
    """pack_weight_wino + the kernel's transforms (tests/emu.py:wino_conv) reproduce the direct correlation exactly in
    float64 — pins the [Cin][4][CoutP][4] layout and the G / B / A matrices the HIP kernel hard-codes."""
    rs = np.random.RandomState(cin + cout)
    wt = torch.tensor(rs.randn(cout, cin, 3, 3))
    x = torch.tensor(rs.randn(2, cin, h, w))
    U = conv.pack_weight_wino(wt)
    assert U.shape == (cin, 4, (cout + 31) // 32 * 32, 4) and U.dtype == torch.float32
    U64 = torch.einsum('ik,ockl,jl->ocij', torch.tensor(conv._WINO_G), wt, torch.tensor(conv._WINO_G)).permute(1, 2, 0, 3)
    assert torch.allclose(U[:, :, :cout, :].double(), U64, rtol=1e-6, atol=1e-7)
    got = emu.wino_conv(x, U64.contiguous(), cout, pad)
    want = F.conv2d(x, wt, padding=pad)
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)
    # the flipped / transposed weights of the input-gradient plan go through the same pack
    fc = conv.FrozenConv2d(wt.float(), 1, 1, device='cpu')
    gy = torch.tensor(rs.randn(2, cout, h, w))
    xr = x.clone().requires_grad_(True)
    gref, = torch.autograd.grad(F.conv2d(xr, wt, padding=1), xr, gy)
    Ub = fc.bwd[0].wino_pack()
    if Ub is not None:                                   # needs Cin(of the gradient conv) = cout % 8 == 0
        assert torch.allclose(emu.wino_conv(gy, Ub.double(), cin, 1), gref, rtol=1e-5, atol=1e-5)


def test_winograd_f4x4_weight_pack_is_the_kernels_lds_image():
    """conv.pack_weight_wino4: U = G g G^T per (cout, cin) in the order csrc/l2i_wino4.hip copies it into LDS ([Cin/4][CoutP/16] images of
    [6 i][4 cin][16 cout][4 j] ++ [6][4][16][2 j]).  Unpacked with that layout and pushed through Y = A^T [U . (B^T d B)] A in float64 it must
    reproduce the direct 3x3 correlation; padded output channels are zero."""
    import torch.nn.functional as F
    from latent2im_amd import conv
    rs = np.random.RandomState(0)
    cout, cin = 24, 8
    w = torch.tensor(rs.randn(cout, cin, 3, 3))
    pk = conv.pack_weight_wino4(w)
    assert pk.dtype == torch.float32 and tuple(pk.shape) == (cin // 4, 2, 2304)
    img = pk.double().numpy()
    U = np.zeros((32, cin, 6, 6))
    for c4 in range(cin // 4):
        for mb in range(2):
            a = img[c4, mb, :6 * 4 * 16 * 4].reshape(6, 4, 16, 4)
            b = img[c4, mb, 6 * 4 * 16 * 4:].reshape(6, 4, 16, 2)
            U[mb * 16:(mb + 1) * 16, c4 * 4:(c4 + 1) * 4, :, :4] = a.transpose(2, 1, 0, 3)
            U[mb * 16:(mb + 1) * 16, c4 * 4:(c4 + 1) * 4, :, 4:] = b.transpose(2, 1, 0, 3)
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    x = rs.randn(cin, 6, 6)
    V = np.einsum('ik,ckl,jl->cij', BT, x, BT)
    Y = np.einsum('ik,okl,jl->oij', AT, np.einsum('ocij,cij->oij', U, V), AT)
    ref = F.conv2d(torch.tensor(x)[None], w)[0].numpy()
    assert np.abs(Y[:cout] - ref).max() < 1e-4 * np.abs(ref).max()          # (the pack is stored in float32)
    assert np.abs(Y[cout:]).max() == 0.0


def test_kernel_source_hash_follows_the_sources_not_the_binary():
    """bench.py keys roofline.traffic to _lib.source_hash(): stable across calls, sensitive to the kernel sources and the header."""
    import os
    from latent2im_amd import _lib
    h = _lib.source_hash()
    assert h == _lib.source_hash() and len(h) == 16
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'l2i.h')
    src = open(hdr).read()
    assert '#define L2I_ABI_VERSION %d' % _lib.ABI_VERSION in src


def test_fused_trunk_shape_predicates():
    """[r6] Which of ResNet-50's chained convs the library fuses (conv.pair_h8_shapes_ok / chain3_h8_shapes_ok / pair_f32_shapes_ok mirror the refusals of
    l2i_conv1x1_pair_h8 / l2i_conv_chain3_h8 / l2i_conv1x1_pair_f32): the trunk shapes of a 1024^2 and of a 256^2 regressor input, and what must stay separate."""
    from latent2im_amd import conv
    # (first conv's input channels, wide channels, second conv's output channels, map side)
    assert conv.pair_h8_shapes_ok(64, 256, 64, 256 * 256) and conv.pair_h8_shapes_ok(64, 256, 128, 256 * 256) and conv.pair_h8_shapes_ok(128, 512, 128, 128 * 128)
    assert conv.pair_h8_shapes_ok(128, 512, 256, 128 * 128) and conv.pair_h8_shapes_ok(256, 1024, 256, 64 * 64)
    assert not conv.pair_h8_shapes_ok(512, 2048, 512, 32 * 32) and not conv.pair_h8_shapes_ok(256, 1024, 512, 64 * 64)      # layer4: too few pixels for the tiling
    assert not conv.pair_h8_shapes_ok(64, 256, 64, 9 * 9)                                                              # H * W must be a multiple of 128
    assert conv.chain3_h8_shapes_ok(64, 256, 64, 256, 256) and conv.chain3_h8_shapes_ok(128, 512, 128, 32, 32) and conv.chain3_h8_shapes_ok(64, 256, 128, 64, 64)
    assert not conv.chain3_h8_shapes_ok(256, 1024, 256, 64, 64) and not conv.chain3_h8_shapes_ok(64, 256, 64, 16, 16)      # the 3x3 head: C <= 128, W % 32 == 0
    assert conv.pair_f32_shapes_ok(64, 256, 64, 256 * 256) and conv.pair_f32_shapes_ok(128, 512, 128, 128 * 128) and not conv.pair_f32_shapes_ok(256, 1024, 256, 64 * 64)
    assert not conv.pair_f32_shapes_ok(64, 256, 64, 8 * 16)
