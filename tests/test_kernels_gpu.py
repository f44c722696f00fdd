"""GPU parity of every C-ABI kernel (include/l2i.h) against the CPU oracle / torch CPU ops on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from latent2im_amd import _lib, conv, kernels, synth
from oracle import sg2

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def close(a, b, rtol=1e-4, atol=1e-4):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


CONV_CASES = [  # cin, cout, k, stride, pad, transposed, h, w, batch
    (8, 32, 3, 1, 1, False, 16, 16, 2), (6, 40, 3, 1, 1, False, 37, 45, 1), (64, 64, 3, 1, 1, False, 64, 64, 1),
    (32, 128, 1, 1, 0, False, 32, 32, 2), (3, 64, 7, 2, 3, False, 64, 64, 1), (16, 24, 3, 2, 1, False, 33, 31, 2),
    (16, 24, 1, 2, 0, False, 16, 16, 1), (12, 20, 3, 2, 0, True, 8, 8, 2), (32, 32, 3, 2, 0, True, 33, 33, 1),
    (512, 512, 3, 1, 1, False, 4, 4, 2), (130, 70, 3, 1, 1, False, 8, 8, 1), (5, 3, 3, 1, 1, False, 40, 40, 1),
    (64, 3, 7, 2, 3, False, 16, 16, 1),
    # <= 3 output channels on wide maps: the register-streaming 3x3 kernel (forward of a 64 -> 3 layer; input-gradient of a 3 -> 40 layer, ragged bands / strips)
    (64, 3, 3, 1, 1, False, 40, 256, 2), (3, 40, 3, 1, 1, False, 21, 200, 1), (16, 2, 3, 1, 1, False, 9, 516, 1),
    # tiny maps (ResNet-50 tail at small inputs): several samples per tile, 1-pixel rows
    (256, 512, 1, 2, 0, False, 2, 2, 4), (2048, 512, 1, 1, 0, False, 1, 1, 4), (512, 2048, 1, 1, 0, False, 1, 1, 3), (64, 64, 3, 1, 1, False, 1, 1, 5),
    (128, 128, 3, 2, 1, False, 3, 3, 2),
    # tiny maps at batch >= 8: a tile packs 8..128 samples, the per-(sample, channel) scale table is longer than the block (round-2 bug:
    # ResNet-50's tail was wrong for batch >= 8 on <= 4x4 maps)
    (128, 512, 1, 1, 0, False, 4, 4, 8), (512, 128, 1, 1, 0, False, 4, 4, 16), (1024, 256, 1, 1, 0, False, 2, 2, 16),
    (2048, 512, 1, 1, 0, False, 1, 1, 16), (256, 512, 1, 2, 0, False, 8, 8, 8), (512, 512, 3, 1, 1, False, 2, 2, 16),
]


@pytest.fixture(params=['off', 'all'])
def wino4(request):
    """Both Winograd kernels: F(2x2,3x3) everywhere ('off'), F(4x4,3x3) on the unmasked launches of maps >= 64 wide ('all')."""
    prev, conv.WINO4 = conv.WINO4, request.param
    yield request.param
    conv.WINO4 = prev


@pytest.fixture
def wino4_off():
    """The generic-kernel tests keep their fp32-exact bounds: 3x3 stride-1 shapes wide enough for F(4x4,3x3) run on the F(2x2) kernel here
    (test_winograd_3x3_conv covers F(4x4) with its own bound)."""
    prev, conv.WINO4 = conv.WINO4, 'off'
    yield
    conv.WINO4 = prev


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('hint', [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_conv_forward_dgrad_vs_torch(case, hint, wino4_off):
    cin, cout, k, stride, pad, tr, h, w, b = case
    if hint and [4, 2, 1, 2, 1, 1, 4, 2][hint - 1] * 32 > (cout + 31) // 32 * 32 + 96:
        pytest.skip('tile much larger than the layer')
    rs = np.random.RandomState(cin + 7 * cout + k + hint)
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    x = T(rs.randn(b, cin, h, w)).requires_grad_(True)
    ref = F.conv_transpose2d(x, wt.transpose(0, 1), stride=2, padding=pad) if tr else F.conv2d(x, wt, stride=stride, padding=pad)
    fc = conv.FrozenConv2d(wt, stride, pad, transposed=tr, device=DEV)
    try:
        y = fc.forward(x.detach().to(DEV), tile_hint=hint)
    except Exception as e:
        if hint and 'cannot be staged' in str(e):
            pytest.skip('forced tile does not fit this problem (auto selection never picks it)')
        raise
    torch.cuda.synchronize()
    close(y, ref, 1e-4, 2e-5)
    gy = T(rs.randn(*ref.shape))
    gref, = torch.autograd.grad(ref, x, gy)
    try:
        gx = fc.dgrad(gy.to(DEV), (h, w), tile_hint=hint)
    except Exception as e:
        if hint and 'cannot be staged' in str(e):
            pytest.skip('forced tile does not fit this problem (auto selection never picks it)')
        raise
    close(gx, gref, 1e-4, 2e-5)


@pytest.mark.parametrize('cin,cout,pad,h,w,b', [(16, 24, 1, 64, 64, 2), (32, 64, 0, 132, 132, 1), (8, 40, 0, 69, 133, 2), (64, 96, 1, 72, 200, 1),
                                               (256, 128, 1, 16, 64, 2), (32, 64, 0, 67, 1028, 1), (12, 70, 0, 19, 65, 3)])
def test_conv3x3_stride2_dma_kernel(cin, cout, pad, h, w, b):
    """[r5] conv3x3s2_dma_kernel (l2i_conv_s2.hip: 3x3 stride-2 layers with both operands staged by LDS-DMA, odd plane pitch, three-stage ring) against
    a float64 correlation with every fusion it serves on the path — style scale in / demodulation out / noise / bias / leaky ReLU (the gradient of a
    generator up layer and the discriminator's strided convs), residual + masks + accumulate — on pad 0 and pad 1, odd input pitches (pad 0 only),
    widths that leave partial tiles and windows that run past the right image edge, channel counts that leave half-filled channel blocks, long K
    loops; and against the generic kernel on the same launch (forced through tile_hint), which must agree to fp32 rounding."""
    rs = np.random.RandomState(cin + cout + h + w)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
    oh, ow = (h + 2 * pad - 3) // 2 + 1, (w + 2 * pad - 3) // 2 + 1
    assert ow >= 32
    bias, res, rmk, omk, prev = T(rs.randn(cout)), *(T(rs.randn(b, cout, oh, ow)) for _ in range(4))
    nz = T(rs.randn(b, 1, oh, ow))
    g = lambda t: t.to(DEV)
    D = lambda t: t.double()
    fc = conv.FrozenConv2d(wt, 2, pad, device=DEV)
    c64 = lambda xx: F.conv2d(xx, D(wt), stride=2, padding=pad)
    ref1 = F.leaky_relu(c64(D(x) * D(s)[:, :, None, None]) * D(d)[:, :, None, None] + D(nz) * 0.3 + D(bias)[None, :, None, None], 0.2) * 2 ** 0.5
    ref2 = torch.relu(c64(D(x)) + D(bias)[None, :, None, None] + torch.where(rmk > 0, D(res), torch.zeros_like(D(res))))
    ref3 = torch.where(omk > 0, c64(D(x)), torch.zeros_like(D(res))) * 0.5 + D(prev)
    for hint in (0, 2):                                   # 0: the dispatch (DMA kernel); 2: the generic kernel's (2, 2) tile
        launched = conv.PROFILE = []
        try:
            y1 = fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5, tile_hint=hint)
            y2 = fc.forward(g(x), bias=g(bias), residual=g(res), res_mask=g(rmk), act=conv.ACT_RELU, tile_hint=hint)
            y3 = g(prev).clone()
            fc.forward(g(x), out=y3, out_mask=g(omk), out_gain=0.5, accumulate=True, tile_hint=hint)
        finally:
            conv.PROFILE = None
        assert all(q[5] == 'implicit_gemm_f32' for q in launched)
        for got, want in ((y1, ref1), (y2, ref2), (y3, ref3)):
            err = float((got.double().cpu() - want).abs().max() / want.abs().max())
            assert err < 5e-6, (hint, err)


@pytest.mark.parametrize('c,h,w,pad', [(3, 40, 260, (1, -2, 1, -2)), (2, 70, 196, (2, 1, 2, 1)), (5, 36, 68, (1, -2, 1, -2)), (2, 19, 33, (1, 1, 1, 1))])
def test_upfirdn2d_masked_epilogue(c, h, w, pad):
    """[r5] l2i_upfirdn2d_masked_f32: the reference FIR with the gradient of a (leaky) ReLU fused behind it (the discriminator's backward), on the
    register-streaming kernel (>= 192 columns), the staged kernel and the generic one; against the unmasked entry point times the mask, bit for bit,
    and against the reference's upfirdn2d_native arithmetic through it."""
    rs = np.random.RandomState(c + h + w)
    x = T(rs.randn(2, c, h, w)).to(DEV)
    k1 = np.array([1., 3., 3., 1.])
    k = T(np.outer(k1, k1) / 64.0).to(DEV)
    add = None
    y0 = kernels.upfirdn2d(x, k, pad=pad)
    m = T(rs.randn(*y0.shape)).to(DEV)
    ym = kernels.upfirdn2d(x, k, pad=pad, mask=m, mask_vals=(2 ** 0.5, 0.2 * 2 ** 0.5))
    want = y0 * torch.where(m > 0, torch.tensor(2 ** 0.5, device=DEV), torch.tensor(0.2 * 2 ** 0.5, device=DEV))
    assert torch.equal(ym, want)
    add = T(rs.randn(*y0.shape)).to(DEV)
    ya = kernels.upfirdn2d(x, k, pad=pad, addend=add, mask=m, mask_vals=(1.0, 0.0))
    assert torch.equal(ya, kernels.upfirdn2d(x, k, pad=pad, addend=add) * (m > 0))


def test_conv_prologue_epilogue_fusions():
    rs = np.random.RandomState(5)
    wt = T(rs.randn(48, 40, 3, 3) / 19.0)
    x, s, d = T(rs.randn(2, 40, 20, 20)), T(rs.rand(2, 40) + 0.5), T(rs.rand(2, 48) + 0.5)
    nz, bias = T(rs.randn(2, 1, 20, 20)), T(rs.randn(48))
    msk, res, rmask = T(rs.randn(2, 40, 20, 20)), T(rs.randn(2, 48, 20, 20)), T(rs.randn(2, 48, 20, 20))
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    g = lambda t: t.to(DEV)
    y = fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    ref = F.leaky_relu(F.conv2d(x * s[:, :, None, None], wt, padding=1) * d[:, :, None, None] + 0.3 * nz + bias[None, :, None, None], 0.2) * 2 ** 0.5
    close(y, ref, 1e-4, 2e-5)
    y = fc.forward(g(x), in_mask=g(msk), mask=(2 ** 0.5, 0.2 * 2 ** 0.5), bias=g(bias), residual=g(res), res_mask=g(rmask), act=conv.ACT_RELU, out_gain=0.5)
    xm = x * torch.where(msk > 0, torch.tensor(2 ** 0.5), torch.tensor(0.2 * 2 ** 0.5))
    ref = torch.relu(F.conv2d(xm, wt, padding=1) + bias[None, :, None, None] + res * (rmask > 0)) * 0.5
    close(y, ref, 1e-4, 2e-5)
    y0 = g(res).clone()
    fc.forward(g(x), out=y0, accumulate=True)
    close(y0, res + F.conv2d(x, wt, padding=1), 1e-4, 2e-5)
    # the streaming <= 3-channel 3x3 kernel with a gradient mask, an output gain and accumulation
    wt3 = T(rs.randn(3, 24, 3, 3) / 15.0)
    x3, m3, y3 = T(rs.randn(2, 24, 18, 260)), T(rs.randn(2, 24, 18, 260)), T(rs.randn(2, 3, 18, 260))
    fc3 = conv.FrozenConv2d(wt3, 1, 1, device=DEV)
    out = g(y3).clone()
    fc3.forward(g(x3), out=out, in_mask=g(m3), mask=(2 ** 0.5, 0.2 * 2 ** 0.5), out_gain=0.5, accumulate=True)
    xm3 = x3 * torch.where(m3 > 0, torch.tensor(2 ** 0.5), torch.tensor(0.2 * 2 ** 0.5))
    close(out, y3 + 0.5 * F.conv2d(xm3, wt3, padding=1), 1e-4, 2e-5)


@pytest.mark.parametrize('cin,cout,k,pad,tr,h,w,b', [(24, 40, 3, 0, True, 17, 17, 2), (40, 24, 3, 1, False, 34, 30, 2), (3, 64, 7, 3, False, 64, 64, 1),
                                                     (64, 3, 7, 3, False, 32, 32, 2), (512, 512, 3, 0, True, 4, 4, 3), (16, 32, 3, 0, False, 9, 9, 2),
                                                     (512, 96, 3, 0, True, 8, 8, 8), (256, 64, 3, 0, True, 16, 16, 4), (64, 256, 3, 0, False, 17, 17, 8),   # these three: split-K
                                                     (20, 40, 3, 0, True, 33, 31, 2), (6, 40, 3, 0, True, 12, 12, 2), (136, 32, 3, 1, False, 40, 40, 1),    # Cin % 8 != 0; Cin % 4 != 0 (not fused)
                                                     (3, 64, 7, 3, False, 128, 96, 2), (3, 44, 7, 3, False, 72, 40, 1), (2, 16, 7, 3, False, 70, 66, 1),
                                                     (64, 48, 3, 0, True, 40, 48, 2), (128, 64, 3, 0, True, 64, 64, 1), (32, 32, 3, 0, True, 35, 70, 3), (256, 96, 3, 0, True, 32, 32, 2)])   # [r5] maps >= 32 positions wide, unmasked: the DMA-input variant; before them: stem gradient onto <= 3 channels: the small-output kernel (several tiles; a partial channel chunk; odd gradient width: per-parity launches)
def test_fused_transposed_conv_matches_per_parity_launches(cin, cout, k, pad, tr, h, w, b):
    """One-launch stride-2 transposed conv (l2i_conv_transpose2d_f32) == the four per-parity l2i_conv2d_f32 launches == torch."""
    rs = np.random.RandomState(cin + cout + h)
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    fc = conv.FrozenConv2d(wt, 2, pad, transposed=tr, device=DEV)
    g = lambda t: t.to(DEV)
    if tr:      # forward of an up layer: style scale in, demod scale out
        x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
        ref = F.conv_transpose2d(x * s[:, :, None, None], wt.transpose(0, 1), stride=2, padding=pad) * d[:, :, None, None]
        y1 = fc.forward(g(x), in_scale=g(s), out_scale=g(d))
        conv.USE_FUSED_TRANSPOSED = False
        try:
            y0 = fc.forward(g(x), in_scale=g(s), out_scale=g(d))
        finally:
            conv.USE_FUSED_TRANSPOSED = True
    else:       # input-gradient of a stride-2 conv with a leaky-ReLU mask on the incoming gradient
        oh, ow = fc.out_hw(h, w)
        gy, y = T(rs.randn(b, cout, oh, ow)), T(rs.randn(b, cout, oh, ow))
        xr = T(rs.randn(b, cin, h, w)).requires_grad_(True)
        gm = gy * torch.where(y > 0, torch.tensor(1.0), torch.tensor(0.2))
        ref, = torch.autograd.grad(F.conv2d(xr, wt, stride=2, padding=pad), xr, gm)
        y1 = fc.dgrad(g(gy), (h, w), in_mask=g(y), mask=(1.0, 0.2))
        conv.USE_FUSED_TRANSPOSED = False
        try:
            y0 = fc.dgrad(g(gy), (h, w), in_mask=g(y), mask=(1.0, 0.2))
        finally:
            conv.USE_FUSED_TRANSPOSED = True
    close(y1, ref, 1e-4, 2e-5)
    close(y0, ref, 1e-4, 2e-5)
    if k == 3 and cin % 4 == 0 and (tr or cout % 4 == 0):          # both position-tile sizes of the fused kernel
        for hint in (1, 2):
            yh = fc.forward(g(x), in_scale=g(s), out_scale=g(d), tile_hint=hint) if tr else fc.dgrad(g(gy), (h, w), in_mask=g(y), mask=(1.0, 0.2), tile_hint=hint)
            close(yh, ref, 1e-4, 2e-5)


BF16X3_CASES = [  # cin, cout, k, stride, pad, h, w, batch
    (64, 64, 3, 1, 1, 64, 64, 2), (32, 32, 3, 1, 1, 40, 96, 1), (16, 40, 3, 1, 1, 36, 36, 1), (512, 512, 3, 1, 1, 32, 32, 1), (48, 100, 3, 1, 1, 33, 64, 2),
    (128, 96, 1, 1, 0, 32, 64, 2), (512, 256, 1, 1, 0, 64, 64, 1), (256, 512, 1, 1, 0, 37, 40, 2), (1024, 96, 1, 1, 0, 32, 32, 1),
    (64, 64, 3, 2, 1, 64, 64, 2), (32, 48, 3, 2, 1, 66, 128, 1), (128, 128, 3, 2, 0, 65, 68, 1), (32, 64, 3, 2, 0, 129, 132, 2),
    (64, 128, 1, 2, 0, 64, 64, 2), (256, 512, 1, 2, 0, 66, 72, 1),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,h,w,b', BF16X3_CASES)
def test_bf16x3_split_precision_conv(cin, cout, k, stride, pad, h, w, b):
    """The bf16 MFMA path with the 3-term operand split (csrc/l2i_conv16.hip: pipelined 1x1 / 3x3, stride 1 / 2): fp32-class accuracy
    (products exact to ~2^-17) against a float64 reference, same fused prologue / epilogue semantics as the fp32 kernels, partial tiles,
    channel counts that are not multiples of the block, and the input-gradient of the stride-1 layers (same kernel, flipped pack)."""
    rs = np.random.RandomState(cin + cout + h + k + stride)
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
    bias, msk = T(rs.randn(cout)), T(rs.randn(b, cin, h, w))
    g = lambda t: t.to(DEV)
    fc = conv.FrozenConv2d(wt, stride, pad, device=DEV)
    oh, ow = fc.out_hw(h, w)
    nz, res, rmask = T(rs.randn(b, 1, oh, ow)), T(rs.randn(b, cout, oh, ow)), T(rs.randn(b, cout, oh, ow))
    xd = x.double()
    cv = lambda t: F.conv2d(t, wt.double(), stride=stride, padding=pad)
    ref = F.leaky_relu(cv(xd * s.double()[:, :, None, None]) * d.double()[:, :, None, None] + 0.3 * nz.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    xm = xd * torch.where(msk > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64))
    ref2 = torch.relu(cv(xm) + bias.double()[None, :, None, None] + res.double() * (rmask > 0)) * 0.5
    launched = []
    conv.PRECISION, conv.PROFILE = 'bf16x3', launched
    try:
        y = fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
        y2 = fc.forward(g(x), in_mask=g(msk), mask=(1.0, 0.2), bias=g(bias), residual=g(res), res_mask=g(rmask), act=conv.ACT_RELU, out_gain=0.5)
        y0 = g(res).clone()
        fc.forward(g(x), out=y0, accumulate=True)
        if stride == 1:
            gy = T(rs.randn(b, cout, oh, ow))
            xr = xd.clone().requires_grad_(True)
            gref, = torch.autograd.grad(cv(xr), xr, gy.double())
            gx = fc.dgrad(g(gy), (h, w))
    finally:
        conv.PRECISION, conv.PROFILE = 'f32', None
    torch.cuda.synchronize()
    # the split kernel ran, not a fallback (the input-gradient has Cout input channels: eligible when those are whole 16 / 32-channel groups)
    # (HBM-bound 1x1 stride-1 layers — Cin*Cout < 2^16 or a gradient mask — deliberately stay on the DMA-fed fp32 GEMM: conv._bf16x3_eligible)
    small_1x1 = k == 1 and stride == 1 and cin * cout < 65536
    want = ['l2i_conv2d_f32' if small_1x1 else 'l2i_conv2d_bf16x3_f32', 'l2i_conv2d_f32' if (k == 1 and stride == 1) else 'l2i_conv2d_bf16x3_f32',
            'l2i_conv2d_f32' if small_1x1 else 'l2i_conv2d_bf16x3_f32']
    assert [q[4] for q in launched[:3]] == want, [q[4] for q in launched]
    if stride == 1 and cout % (32 if k == 1 else 16) == 0 and not small_1x1:
        assert launched[3][4] == 'l2i_conv2d_bf16x3_f32'
    checks = [(y, ref), (y2, ref2), (y0, res.double() + cv(xd))] + ([(gx, gref)] if stride == 1 else [])
    for got, want in checks:
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        assert err < 2e-5, err
    # ContentLoss value fused into the shared vectorised epilogue (sq_ref): only where the output rows are whole 16-byte vectors
    if k == 3 and stride == 1:
        sq_acc, fused = torch.zeros(_lib.SQ_SLOTS, device=DEV), [False]
        conv.PRECISION = 'bf16x3'
        try:
            y4 = fc.forward(g(x), in_mask=g(msk), mask=(1.0, 0.0), bias=g(bias), sq=(g(res), sq_acc, fused))
        finally:
            conv.PRECISION = 'f32'
        assert fused[0] == (ow % 4 == 0)
        if fused[0]:
            want_sq = float(((y4.double().cpu() - res.double()) ** 2).sum())
            assert abs(float(sq_acc.double().sum()) - want_sq) <= 1e-5 * want_sq


@pytest.mark.parametrize('cin,cout,h,w,b', [(64, 64, 64, 64, 2), (32, 32, 40, 96, 1), (128, 96, 32, 64, 2), (16, 40, 36, 36, 1),
                                            (512, 512, 32, 32, 1), (8, 64, 64, 128, 3), (64, 128, 128, 32, 1), (40, 24, 50, 132, 2)])
def test_winograd_3x3_conv(cin, cout, h, w, b, wino4):
    """F(2x2,3x3) / F(4x4,3x3) fp32 kernels vs a float64 reference: every prologue / epilogue fusion, partial tiles, channel counts that
    do not fill a block, forward and input-gradient; the direct kernel on the same problem for scale.  Bound: 5e-6 of max|y| for F(2x2)
    (measured 3e-7 .. 1e-6), 3e-5 for F(4x4) (its transforms carry constants up to 8: measured 1e-6 .. 1e-5)."""
    if wino4 == 'all' and (w < 32 or cin % 4):
        pytest.skip('F(4x4,3x3) takes maps >= 32 wide with Cin % 4 == 0: this shape runs the F(2x2) kernel in both settings')
    bound = 3e-5 if wino4 == 'all' else 5e-6
    rs = np.random.RandomState(cin + cout + h)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
    bias, msk = T(rs.randn(cout)), T(rs.randn(b, cin, h, w))
    res, rmk, omk = (T(rs.randn(b, cout, h, w)) for _ in range(3))
    nz = T(rs.randn(b, 1, h, w))
    y_prev = T(rs.randn(b, cout, h, w))
    g = lambda t: t.to(DEV)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    assert fc.fwd[0].wino_pack() is not None
    if cin % 4 == 0:
        assert fc.fwd[0].wino4_pack() is not None
    D = lambda t: t.double()
    c64 = lambda xx: F.conv2d(xx, D(wt), padding=1)
    ref1 = F.leaky_relu(c64(D(x) * D(s)[:, :, None, None]) * D(d)[:, :, None, None] + D(nz) * 0.3 + D(bias)[None, :, None, None], 0.2) * 2 ** 0.5
    xm = D(x) * torch.where(msk > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64))
    ref2 = torch.relu(c64(xm) + D(bias)[None, :, None, None] + torch.where(rmk > 0, D(res), torch.zeros_like(D(res))))
    ref3 = torch.where(omk > 0, c64(D(x)), torch.zeros_like(D(res))) * 0.5 + D(y_prev)
    gy = T(rs.randn(b, cout, h, w))
    xr = D(x).clone().requires_grad_(True)
    gref, = torch.autograd.grad(c64(xr), xr, D(gy))
    y1 = fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    y2 = fc.forward(g(x), in_mask=g(msk), mask=(1.0, 0.2), bias=g(bias), residual=g(res), res_mask=g(rmk), act=conv.ACT_RELU)
    y3 = g(y_prev).clone()
    fc.forward(g(x), out=y3, out_mask=g(omk), out_gain=0.5, accumulate=True)
    gx = fc.dgrad(g(gy), (h, w)) if cout % 8 == 0 else None
    for got, want in ((y1, ref1), (y2, ref2), (y3, ref3), (gx, gref)):
        if got is None:
            continue
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        assert err < bound, err
    # the gradient mask IS the input (VGG-19: a conv reads the pre-ReLU tap of the layer below): pro(x) = max(x, 0) without a second tile stream
    xg = g(x)
    y5 = fc.forward(xg, in_mask=xg, mask=(1.0, 0.0), bias=g(bias))
    ref5 = c64(torch.relu(D(x))) + D(bias)[None, :, None, None]
    assert float((y5.double().cpu() - ref5).abs().max() / ref5.abs().max()) < bound
    # ContentLoss value fused into the epilogue (VGG taps): sum (y - reference)^2 over the launch, partial tiles and padded channels excluded
    sq_acc, fused = torch.zeros(_lib.SQ_SLOTS, device=DEV), [False]
    y4 = fc.forward(g(x), in_mask=g(msk), mask=(1.0, 0.0), bias=g(bias), sq=(g(res), sq_acc, fused))
    want_sq = float(((y4.double().cpu() - D(res)) ** 2).sum())
    assert fused[0] and abs(float(sq_acc.double().sum()) - want_sq) <= 1e-5 * want_sq
    # ... and on an unmasked launch (the F(4x4) kernel's own epilogue when it is on), with the noise / bias / residual operands beside it
    sq_acc.zero_()
    fused = [False]
    y6 = fc.forward(xg, in_mask=xg, mask=(1.0, 0.0), bias=g(bias), noise=g(nz), noise_w=0.3, residual=g(res), sq=(g(rmk), sq_acc, fused))
    ref6 = ref5 + D(nz) * 0.3 + D(res)
    assert float((y6.double().cpu() - ref6).abs().max() / ref6.abs().max()) < bound
    # the ContentLoss gradient term res_coef * res_coef_dev[0] * (residual - res_sub) with a residual mask, on an unmasked launch
    y7 = fc.forward(xg, in_mask=xg, mask=(1.0, 0.0), residual=g(res), res_sub=g(rmk), res_coef=0.25, res_coef_dev=torch.full((1,), 2.0, device=DEV), res_mask=g(omk))
    ref7 = c64(torch.relu(D(x))) + torch.where(omk > 0, 0.5 * (D(res) - D(rmk)), torch.zeros_like(D(res)))
    assert float((y7.double().cpu() - ref7).abs().max() / ref7.abs().max()) < bound
    want_sq = float(((y6.double().cpu() - D(rmk)) ** 2).sum())
    assert fused[0] and abs(float(sq_acc.double().sum()) - want_sq) <= 1e-5 * want_sq
    # the direct kernel on the same problem, for scale (and as a cross-check of the dispatch switch)
    conv.USE_WINOGRAD = False
    try:
        y0 = fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
        fused = [False]
        fc.forward(g(x), bias=g(bias), sq=(g(res), sq_acc, fused))
        assert not fused[0]                                        # other kernels leave the sum to the caller (perceptual.taps falls back to sqdiff)
    finally:
        conv.USE_WINOGRAD = True
    e0 = float((y0.double().cpu() - ref1).abs().max() / ref1.abs().max())
    assert e0 < 5e-6, e0


@pytest.mark.parametrize('cin,cout,h,w,b', [(64, 64, 64, 64, 2), (8, 64, 64, 128, 3), (40, 24, 50, 132, 2), (256, 128, 16, 64, 1), (512, 512, 8, 64, 1),
                                            (32, 32, 40, 96, 1), (128, 96, 7, 64, 2), (16, 40, 9, 72, 1),
                                            (512, 512, 32, 32, 1), (64, 96, 37, 40, 2), (32, 64, 16, 60, 1)])      # the last three: maps 32 .. 63 wide = the 32 x 16-pixel tile
def test_wino4_position_split_kernel_bit_identical_to_round4_kernel(cin, cout, h, w, b):
    """[r5] The position-split F(4x4,3x3) kernel (two waves share the 36 Winograd positions of a tile row, 32 output channels per block, 16-byte
    raw-tile DMA through a 72-column aligned window) does the round-4 kernel's arithmetic in the same order: the two must agree BIT FOR BIT on every
    instantiation (plain, style-scaled, ReLU-on-load, both) with every epilogue operand, on ragged maps (rows not a multiple of 8, widths not a
    multiple of 64, channel counts that do not fill a block) and on long K loops (both U stages, all three raw stages many times over)."""
    rs = np.random.RandomState(cin * 7 + cout + h + w)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
    bias, res, rmk, omk, rsb = T(rs.randn(cout)), *(T(rs.randn(b, cout, h, w)) for _ in range(4))
    nz = T(rs.randn(b, 1, h, w))
    g = lambda t: t.to(DEV)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    xg = g(x)
    coef = torch.full((1,), 2.0, device=DEV)
    calls = [
        dict(),
        dict(in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5),
        dict(in_mask=xg, mask=(1.0, 0.0), bias=g(bias), residual=g(res), res_mask=g(rmk), act=conv.ACT_RELU),
        dict(in_mask=xg, mask=(1.0, 0.0), in_scale=g(s), residual=g(res), res_sub=g(rsb), res_coef=0.25, res_coef_dev=coef, out_mask=g(omk), out_gain=0.5),
    ]
    prev = conv.WINO4
    try:
        for kw in calls:
            out, sums = {}, {}
            for mode in ('r4', 'all', 'tall'):             # ('tall' [r5]: the same kernel body on a 64 x 16-pixel tile / eight waves where the map has >= 16 rows and >= 64 columns)
                conv.WINO4 = mode
                conv.WINO4_R4_MIN_W = 32                   # (the round-4 kernel is the reference on the narrow maps too: 64-wide tiles, half empty)
                launched = conv.PROFILE = []
                try:
                    sq_acc, fused = torch.zeros(_lib.SQ_SLOTS, device=DEV), [False]
                    y = g(res).clone()
                    fc.forward(xg, out=y, accumulate=True, sq=(g(rmk), sq_acc, fused), **kw)
                finally:
                    conv.PROFILE = None
                assert [q[4] for q in launched] == ['l2i_conv2d_wino4_f32'] and fused[0]
                out[mode], sums[mode] = y, sq_acc.double().sum()
            assert torch.equal(out['r4'], out['all']), float((out['r4'] - out['all']).abs().max())
            assert torch.equal(out['r4'], out['tall']), float((out['r4'] - out['tall']).abs().max())
            assert abs(float(sums['tall']) - float(sums['all'])) <= 1e-5 * abs(float(sums['all']))
            want_sq = float(((out['all'].double() - g(rmk).double()) ** 2).sum())
            assert abs(float(sums['all']) - want_sq) <= 1e-5 * want_sq and abs(float(sums['r4']) - want_sq) <= 1e-5 * want_sq
    finally:
        conv.WINO4 = prev
        conv.WINO4_R4_MIN_W = 64


def test_content_loss_sum_fused_into_the_three_channel_conv():
    """VGG conv_1 (3 -> 64, 3x3) runs on the <= 3-input-channel kernel, which also sums (y - reference)^2 in its epilogue."""
    rs = np.random.RandomState(3)
    wt, bias = T(rs.randn(64, 3, 3, 3) / 5.0), T(rs.randn(64))
    x, ref = T(rs.randn(2, 3, 40, 72)), T(rs.randn(2, 64, 40, 72))
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    acc, fused = torch.zeros(_lib.SQ_SLOTS, device=DEV), [False]
    y = fc.forward(x.to(DEV), bias=bias.to(DEV), sq=(ref.to(DEV), acc, fused))
    want = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    close(y, want.float(), 1e-4, 2e-5)
    assert fused[0]
    np.testing.assert_allclose(float(acc.double().sum()), float(((want - ref.double()) ** 2).sum()), rtol=1e-5)


@pytest.mark.parametrize('cin,cout,h,w,b', [(64, 256, 32, 32, 2), (256, 64, 16, 64, 1), (1024, 256, 16, 16, 2), (128, 512, 16, 32, 3), (16, 64, 64, 64, 1)])
def test_gemm_1x1_conv(cin, cout, h, w, b):
    """Unmasked 1x1 stride-1 layers take the DMA-fed GEMM kernel (l2i_gemm.hip) inside l2i_conv2d_f32; same results as the generic
    implicit-GEMM kernel (forced with a tile hint) and as a float64 reference, with the epilogue fusions ResNet-50 uses."""
    rs = np.random.RandomState(cin + cout + h)
    wt = T(rs.randn(cout, cin, 1, 1) / np.sqrt(cin))
    x, bias = T(rs.randn(b, cin, h, w)), T(rs.randn(cout))
    res, rmk, omk, y_prev = (T(rs.randn(b, cout, h, w)) for _ in range(4))
    g = lambda t: t.to(DEV)
    fc = conv.FrozenConv2d(wt, 1, 0, device=DEV)
    D = lambda t: t.double()
    c64 = F.conv2d(D(x), D(wt))
    ref1 = torch.relu(c64 + D(bias)[None, :, None, None] + D(res))
    ref2 = F.leaky_relu(torch.where(omk > 0, c64, torch.zeros_like(c64)) + torch.where(rmk > 0, D(res), torch.zeros_like(c64)), 0.2) * 2 ** 0.5 * 0.5 + D(y_prev)
    msk = T(rs.randn(b, cin, h, w))
    xm = D(x) * torch.where(msk > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64))
    ref3 = F.conv2d(xm, D(wt)) + torch.where(rmk > 0, D(res), torch.zeros_like(c64))
    for hint in (0, 2):                  # 0: GEMM kernel, 2: generic kernel
        y1 = fc.forward(g(x), bias=g(bias), residual=g(res), act=conv.ACT_RELU, tile_hint=hint)
        y3 = fc.forward(g(x), in_mask=g(msk), mask=(1.0, 0.2), residual=g(res), res_mask=g(rmk), tile_hint=hint)
        assert float((y3.double().cpu() - ref3).abs().max() / ref3.abs().max()) < 5e-6, hint
        y2 = g(y_prev).clone()
        fc.forward(g(x), out=y2, out_mask=g(omk), residual=g(res), res_mask=g(rmk), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5, out_gain=0.5,
                   accumulate=True, tile_hint=hint)
        for got, want in ((y1, ref1), (y2, ref2)):
            err = float((got.double().cpu() - want).abs().max() / want.abs().max())
            assert err < 5e-6, (hint, err)


@pytest.mark.parametrize('cin,cout,k,stride,h,b', [(512, 512, 3, 1, 4, 8), (256, 96, 3, 1, 8, 8), (512, 64, 3, 1, 16, 8), (512, 512, 3, 2, 9, 8), (384, 128, 3, 1, 6, 3)])
def test_split_k_small_maps(cin, cout, k, stride, h, b):
    """4x4 .. 16x16 maps: Cin is cut into ranges computed by separate blocks, a second pass reduces and applies the epilogue
    (l2i.h: ksplit / ws).  Same results as the one-pass kernel and as a float64 reference, forward with the StyledConv fusions
    and masked input-gradient."""
    rs = np.random.RandomState(cin + cout + h)
    pad = 1 if stride == 1 else 0
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    x, s, d, bias = T(rs.randn(b, cin, h, h)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5), T(rs.randn(cout))
    g = lambda t: t.to(DEV)
    fc = conv.FrozenConv2d(wt, stride, pad, device=DEV)
    oh = fc.out_hw(h, h)[0]
    nz, res = T(rs.randn(b, 1, oh, oh)), T(rs.randn(b, cout, oh, oh))
    D = lambda t: t.double()
    ref = F.leaky_relu(F.conv2d(D(x) * D(s)[:, :, None, None], D(wt), stride=stride, padding=pad) * D(d)[:, :, None, None] + 0.3 * D(nz)
                       + D(bias)[None, :, None, None] + D(res), 0.2) * 2 ** 0.5
    outs = []
    for on in (True, False):
        conv.SPLIT_K = on
        try:
            outs.append(fc.forward(g(x), in_scale=g(s), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), residual=g(res),
                                   act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5))
        finally:
            conv.SPLIT_K = True
    for y in outs:
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 5e-6
    if stride == 1:
        gy, msk = T(rs.randn(b, cout, oh, oh)), T(rs.randn(b, cout, oh, oh))
        xr = D(x).clone().requires_grad_(True)
        gm = D(gy) * torch.where(msk > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64))
        gref, = torch.autograd.grad(F.conv2d(xr, D(wt), padding=pad), xr, gm)
        gx = fc.dgrad(g(gy), (h, h), in_mask=g(msk), mask=(1.0, 0.2)) if cout >= 256 else None
        if gx is not None:
            assert float((gx.double().cpu() - gref).abs().max() / gref.abs().max()) < 5e-6


def test_fused_bias_act_golden(golden):
    gd = golden('fused_bias_act')
    x, b, ref = (T(gd[k]).to(DEV) for k in ('x', 'b', 'ref'))
    for act in (1, 3):
        for grad in (0, 1, 2):
            y = kernels.fused_bias_act(x, b if grad == 0 else None, ref if grad == 1 else None, act, grad, 0.2, 2 ** 0.5)
            close(y, T(gd['y_%d%d' % (act, grad)]), 1e-6, 1e-7)
    close(kernels.fused_bias_act(T(gd['x2']).to(DEV), T(gd['b2']).to(DEV), None, 3, 0, 0.2, 2 ** 0.5), T(gd['y2_30']), 1e-6, 1e-7)
    # odd sizes take the scalar path; big maps the float4 path
    rs = np.random.RandomState(0)
    for shape in ((3, 5, 7, 9), (2, 32, 64, 64), (1, 3, 1, 1)):
        xx, bb = T(rs.randn(*shape)), T(rs.randn(shape[1]))
        close(kernels.fused_bias_act(xx.to(DEV), bb.to(DEV), None, 3, 0, 0.2, 2 ** 0.5), sg2.fused_leaky_relu(xx, bb), 1e-6, 1e-7)


def test_upfirdn2d_golden(golden):
    gd = golden('upfirdn2d')
    for i, (n, c, h, w, up, down, p0, p1, gain) in enumerate(gd['cases']):
        up, down, p0, p1 = int(up), int(down), int(p0), int(p1)
        x = T(np.random.RandomState(20 + i).randn(int(n), int(c), int(h), int(w)).astype(np.float32))
        k = T(synth.fir_kernel(gain=gain))
        y = kernels.upfirdn2d(x.to(DEV), k.to(DEV), (up, up), (down, down), (p0, p1, p0, p1))
        close(y, T(gd['y_%d' % i]), 1e-5, 1e-6)
        # backward = same op, flipped kernel, up<->down, g_pad (op/upfirdn2d.py:105-115)
        gy = T(gd['gy_%d' % i])
        g0 = 4 - p0 - 1
        g1y = int(h) * up - gy.shape[2] * down + p0 - up + 1
        g1x = int(w) * up - gy.shape[3] * down + p0 - up + 1
        gx = kernels.upfirdn2d(gy.to(DEV), torch.flip(k, [0, 1]).to(DEV), (down, down), (up, up), (g0, g1x, g0, g1y))
        close(gx, T(gd['gx_%d' % i]), 1e-5, 1e-6)


def test_upfirdn2d_fused_epilogue():
    rs = np.random.RandomState(1)
    x, nz, b, add = T(rs.randn(2, 5, 17, 17)), T(rs.randn(2, 1, 16, 16)), T(rs.randn(5)), T(rs.randn(2, 5, 16, 16))
    k = T(synth.fir_kernel(gain=4.0))
    y = kernels.upfirdn2d(x.to(DEV), k.to(DEV), pad=(1, 1, 1, 1), noise=nz.to(DEV), noise_w=0.4, bias=b.to(DEV), addend=add.to(DEV),
                          act=kernels.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    ref = F.leaky_relu(sg2.upfirdn2d(x, k, pad=(1, 1)) + 0.4 * nz + b[None, :, None, None] + add, 0.2) * 2 ** 0.5
    close(y, ref, 1e-5, 1e-6)
    # wide maps: the register-streaming 4x4 kernel (>= 192 output columns, whole 16-byte rows) with every epilogue operand, pads (1, -2) /
    # (2, 1) / (2, 5) as the generator / discriminator use them, a ragged last band and a partial last strip
    for (h, w, pad) in ((70, 264, (1, -2, 1, -2)), (40, 512, (2, 1, 2, 1)), (19, 200, (2, 5, 2, 5)), (130, 196, (1, 1, 1, 1))):
        oh, ow = kernels.upfirdn2d_out_hw(h, w, 4, 4, (1, 1), (1, 1), pad)
        x, nz, add = T(rs.randn(2, 5, h, w)), T(rs.randn(2, 1, oh, ow)), T(rs.randn(2, 5, oh, ow))
        y = kernels.upfirdn2d(x.to(DEV), k.to(DEV), pad=pad, noise=nz.to(DEV), noise_w=0.4, bias=b.to(DEV), addend=add.to(DEV), act=kernels.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
        xp = F.pad(x, [max(pad[0], 0), max(pad[1], 0), max(pad[2], 0), max(pad[3], 0)])
        xp = xp[:, :, max(-pad[2], 0):xp.shape[2] - max(-pad[3], 0), max(-pad[0], 0):xp.shape[3] - max(-pad[1], 0)]
        ref = F.conv2d(xp.reshape(-1, 1, xp.shape[2], xp.shape[3]), torch.flip(k, [0, 1])[None, None]).reshape(2, 5, oh, ow)
        close(y, F.leaky_relu(ref + 0.4 * nz + b[None, :, None, None] + add, 0.2) * 2 ** 0.5, 1e-5, 1e-5)
        close(kernels.upfirdn2d(x.to(DEV), k.to(DEV), pad=pad), ref, 1e-5, 1e-5)
    # Blur evaluated at every second pixel (down = 2, pad (1, 1): the discriminator's skip branch): the streaming kernel on wide maps, the
    # generic one below 96 output columns
    for h, w in ((64, 256), (50, 200), (16, 40)):
        x = T(rs.randn(2, 3, h, w))
        y = kernels.upfirdn2d(x.to(DEV), (k / 4).to(DEV), down=(2, 2), pad=(1, 1, 1, 1))
        close(y, sg2.upfirdn2d(x, k / 4, down=2, pad=(1, 1)), 1e-5, 1e-6)
    # the ToRGB skip upsample (up = 2, pad (2, 1), addend = the new rgb; networks.py:353-356): the 2x4-patch kernel (output width % 4 == 0)
    # and, for an odd input width, the generic one
    for h, w in ((8, 6), (33, 64), (5, 7)):
        x, add = T(rs.randn(2, 3, h, w)), T(rs.randn(2, 3, 2 * h, 2 * w))
        y = kernels.upfirdn2d(x.to(DEV), k.to(DEV), up=(2, 2), pad=(2, 1, 2, 1), addend=add.to(DEV))
        close(y, sg2.upfirdn2d(x, k, up=2, pad=(2, 1)) + add, 1e-5, 1e-6)


def test_torgb_and_act_bwd_and_reductions():
    rs = np.random.RandomState(2)
    B, C, H = 2, 24, 12
    x, wm, bias = T(rs.randn(B, C, H, H)), T(rs.randn(B, 3, C)), T(rs.randn(3))
    rgb = kernels.torgb_fwd(x.to(DEV), wm.to(DEV), bias.to(DEV))
    close(rgb, torch.einsum('bchw,boc->bohw', x, wm) + bias[None, :, None, None], 1e-4, 1e-5)
    # fused backward pass
    y, gin, gs, grgb = T(rs.randn(B, C, H, H)), T(rs.randn(B, C, H, H)), T(rs.rand(B, C) + 0.5), T(rs.randn(B, 3, H, H))
    cb, nz = T(rs.randn(C)), T(rs.randn(B, 1, H, H))
    red_q = torch.zeros(B * C, device=DEV)
    dz, red, red_rgb = kernels.sg2_act_bwd(y.to(DEV), gin.to(DEV), gs.to(DEV), grgb.to(DEV), wm.to(DEV), cb.to(DEV), nz.to(DEV), 0.3, red_q=red_q)
    close(red_q.view(B, C), (gin * y).sum((2, 3)), 1e-4, 1e-3)          # [r5] red_gin_y: the next layer's style gradient, formed in the same pass
    g = gin * gs[:, :, None, None] + torch.einsum('bohw,boc->bchw', grgb, wm)
    dz_ref = g * torch.where(y > 0, torch.tensor(2 ** 0.5), torch.tensor(0.2 * 2 ** 0.5))
    zpre = torch.where(y > 0, y / 2 ** 0.5, y / (0.2 * 2 ** 0.5)) - cb[None, :, None, None] - 0.3 * nz
    close(dz, dz_ref, 1e-5, 1e-5)
    close(red, (dz_ref * zpre).sum((2, 3)), 1e-4, 1e-3)
    close(red_rgb, torch.einsum('bchw,bohw->bco', y, grgb), 1e-4, 1e-3)
    close(kernels.dot_reduce(x.to(DEV), y.to(DEV)), (x * y).sum((2, 3)), 1e-4, 1e-3)
    close(kernels.dot_reduce(x.to(DEV)), x.sum((2, 3)), 1e-4, 1e-3)
    s, gr = kernels.sqdiff(x.to(DEV), y.to(DEV), coef=0.25, want_grad=True)
    close(s, ((y - x) ** 2).sum().reshape(1), 1e-4, 1e-3)
    close(gr, 0.25 * (y - x), 1e-6, 1e-6)
    close(kernels.relu_mask(x.to(DEV), y.to(DEV)), x * (y > 0), 0, 0)
    close(kernels.axpby(x.to(DEV), y.to(DEV), 0.5, -2.0), 0.5 * x - 2 * y, 1e-6, 1e-6)


@pytest.mark.parametrize('k,s,pad,h', [(3, 2, 1, 16), (2, 2, 0, 16), (3, 2, 1, 15), (3, 2, 1, 72), (2, 2, 0, 40), (3, 2, 1, 8), (2, 2, 0, 6)])      # W = 2 OW with OW % 4 == 0: the vectorised kernels
def test_maxpool(k, s, pad, h):
    rs = np.random.RandomState(k + h)
    x = torch.relu(T(rs.randn(2, 5, h, h))).requires_grad_(True)      # zeros create ties, as after a ReLU
    ref = F.max_pool2d(x, k, s, pad)
    y, idx = kernels.maxpool2d_fwd(x.detach().to(DEV), k, s, pad)
    close(y, ref, 0, 0)
    gy = T(rs.randn(*ref.shape))
    gref, = torch.autograd.grad(ref, x, gy)
    gx = kernels.maxpool2d_bwd(gy.to(DEV), idx, (h, h), k, s, pad)
    close(gx, gref, 1e-6, 1e-6)


@pytest.mark.parametrize('h,w', [(16, 16), (6, 40), (64, 4)])
def test_maxpool2x2_backward_fused_with_content_gradient(h, w):
    """gx = maxpool_bwd(gy) + coef * coef_dev * (b - a) in one pass == the three separate kernels (ties included)."""
    rs = np.random.RandomState(h + w)
    x = torch.relu(T(rs.randn(3, 5, h, w)))
    o = T(rs.randn(3, 5, h, w))
    y, idx = kernels.maxpool2d_fwd(x.to(DEV), 2, 2, 0)
    gy = T(rs.randn(*y.shape)).to(DEV)
    cd = torch.tensor([0.7], device=DEV)
    want = kernels.maxpool2d_bwd(gy, idx, (h, w), 2, 2, 0) + kernels.sqdiff(o.to(DEV), x.to(DEV), coef=0.3, coef_dev=cd, want_grad=True, want_sum=False)[1]
    got = kernels.maxpool2x2_bwd_add_diff(gy, idx, o.to(DEV), x.to(DEV), 0.3, cd)
    close(got, want, 1e-6, 1e-6)
    xr = x.clone().requires_grad_(True)
    ref, = torch.autograd.grad(F.max_pool2d(xr, 2, 2), xr, gy.cpu())
    close(got, ref + 0.3 * 0.7 * (x - o), 1e-5, 1e-6)


@pytest.mark.parametrize('cin,cout,k,h,w,b,precision', [(64, 64, 3, 64, 64, 2, 'f32'),      # Winograd kernel
                                                         (16, 40, 3, 20, 24, 1, 'f32'),      # implicit-GEMM kernel, wide epilogue
                                                         (16, 40, 3, 21, 23, 1, 'f32'),      # ... scalar epilogue (odd width)
                                                         (64, 64, 1, 32, 32, 2, 'f32'),      # DMA-fed 1x1 GEMM
                                                         (512, 64, 3, 8, 8, 2, 'f32'),       # split-K second pass
                                                         (64, 64, 3, 64, 64, 1, 'bf16x3')])
def test_conv_epilogue_difference_residual(cin, cout, k, h, w, b, precision, wino4_off):
    """res_sub: the residual term becomes res_coef * res_coef_dev[0] * (residual - res_sub) (the ContentLoss gradient of a VGG
    tap formed inside the gradient conv), in every kernel family's epilogue, with and without the residual mask."""
    rs = np.random.RandomState(cin + cout + h + w)
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    x, feat, org = T(rs.randn(b, cin, h, w)), T(rs.randn(b, cout, h, w)), T(rs.randn(b, cout, h, w))
    cd = torch.tensor([1.7], device=DEV)
    g = lambda t: t.to(DEV)
    old, conv.PRECISION = conv.PRECISION, precision
    try:
        fc = conv.FrozenConv2d(wt, 1, k // 2, device=DEV)
        y = fc.forward(g(x), out_mask=g(feat), residual=g(feat), res_sub=g(org), res_coef=0.25, res_coef_dev=cd)
        ref = F.conv2d(x, wt, padding=k // 2) * (feat > 0) + 0.25 * 1.7 * (feat - org)
        close(y, ref, 1e-4, 2e-5)
        y = fc.forward(g(x), residual=g(feat), res_sub=g(org), res_coef=-2.0, res_mask=g(org), act=conv.ACT_RELU)
        ref = torch.relu(F.conv2d(x, wt, padding=k // 2) - 2.0 * (feat - org) * (org > 0))
        close(y, ref, 1e-4, 2e-5)
    finally:
        conv.PRECISION = old


@pytest.mark.parametrize('cin,cout,h,w,b', [(3, 64, 64, 128, 2), (3, 64, 20, 36, 1), (1, 96, 8, 8, 2), (2, 40, 33, 12, 1)])
def test_small_cin_3x3_conv_as_one_contraction(cin, cout, h, w, b):
    """<= 3 input channels, 3x3 stride 1 (VGG conv1_1 on the image): the 27-long im2col contraction kernel; bias / activation /
    gain fused; tile_hint != 0 forces the generic kernel for comparison."""
    rs = np.random.RandomState(cin + cout + h + w)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, bias = T(rs.randn(b, cin, h, w)), T(rs.randn(cout))
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    g = lambda t: t.to(DEV)
    ref = F.conv2d(x, wt, bias, padding=1)
    close(fc.forward(g(x), bias=g(bias)), ref, 1e-4, 2e-5)
    close(fc.forward(g(x), bias=g(bias), tile_hint=2), ref, 1e-4, 2e-5)
    close(fc.forward(g(x), bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5, out_gain=0.5), F.leaky_relu(ref, 0.2) * 2 ** 0.5 * 0.5, 1e-4, 2e-5)
    close(fc.forward(g(x), act=conv.ACT_RELU), torch.relu(F.conv2d(x, wt, padding=1)), 1e-4, 2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in autograd wrappers (latent2im_amd.op == the reference's graphs/stylegan_v2_real/op package)
# ---------------------------------------------------------------------------------------------------------------------
def test_op_fused_leaky_relu_autograd_and_double_backward(golden):
    """latent2im_amd.op.fused_leaky_relu / FusedLeakyReLU through autograd (op/fused_act.py:17-86): forward == y_30, the
    backward Function == the reference's act=3, grad=1 table entry (y_31) with grad_bias = its sum over all dims but 1, and the
    double backward (gradgrad_input + gradgrad_bias through the same mask) against torch's own double backward of
    leaky_relu(x + b) * sqrt(2) on the CPU."""
    from latent2im_amd import op
    from latent2im_amd.op import fused_act
    gd = golden('fused_bias_act')
    x, b, ref = (T(gd[k]) for k in ('x', 'b', 'ref'))
    xd, bd = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = op.fused_leaky_relu(xd, bd)
    close(y, T(gd['y_30']), 1e-6, 1e-7)
    # the backward Function on the reference's own vectors: input plays grad_output, ref plays the saved output
    gi, gb = fused_act._FusedLeakyReLUBackward.apply(x.to(DEV), ref.to(DEV), 0.2, 2 ** 0.5)
    close(gi, T(gd['y_31']), 1e-6, 1e-7)
    close(gb, T(gd['y_31']).sum((0, 2, 3)), 1e-5, 1e-6)
    # autograd end to end, first and second order, against torch on the CPU
    rs = np.random.RandomState(3)
    gy, vx, vb = T(rs.randn(*x.shape)), T(rs.randn(*x.shape)), T(rs.randn(*b.shape))
    xc, bc, gyc = x.clone().requires_grad_(True), b.clone().requires_grad_(True), gy.clone().requires_grad_(True)
    yc = F.leaky_relu(xc + bc.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    gxc, gbc = torch.autograd.grad(yc, (xc, bc), gyc, create_graph=True)
    ggc, = torch.autograd.grad((gxc * vx).sum() + (gbc * vb).sum(), gyc)
    gyd = gy.to(DEV).requires_grad_(True)
    gxd, gbd = torch.autograd.grad(y, (xd, bd), gyd, create_graph=True)
    close(gxd, gxc, 1e-6, 1e-7)
    close(gbd, gbc, 1e-5, 1e-6)
    ggd, = torch.autograd.grad((gxd * vx.to(DEV)).sum() + (gbd * vb.to(DEV)).sum(), gyd)
    close(ggd, ggc, 1e-5, 1e-6)
    # the module form: bias parameter of the right shape, gradient reaches it; 2-D inputs (the mapping network's use, networks.py:151)
    m = op.FusedLeakyReLU(5).to(DEV)
    with torch.no_grad():
        m.bias.copy_(b.to(DEV))
    ym = m(x.to(DEV))
    close(ym, T(gd['y_30']), 1e-6, 1e-7)
    ym.backward(gy.to(DEV))
    close(m.bias.grad, gbc, 1e-5, 1e-6)
    x2, b2 = T(gd['x2']).to(DEV).requires_grad_(True), T(gd['b2']).to(DEV).requires_grad_(True)
    y2 = op.fused_leaky_relu(x2, b2)
    close(y2, T(gd['y2_30']), 1e-6, 1e-7)
    g2 = T(rs.randn(3, 7))
    y2.backward(g2.to(DEV))
    x2c, b2c = T(gd['x2']).requires_grad_(True), T(gd['b2']).requires_grad_(True)
    (F.leaky_relu(x2c + b2c, 0.2) * 2 ** 0.5).backward(g2)
    close(x2.grad, x2c.grad, 1e-6, 1e-7)
    close(b2.grad, b2c.grad, 1e-5, 1e-6)


def test_op_upfirdn2d_autograd_and_double_backward(golden):
    """latent2im_amd.op.upfirdn2d through autograd (op/upfirdn2d.py:17-149) on every case of the reference fixture: forward == y_i,
    x.grad for the fixture's gy_i == gx_i (the reference's own backward: flipped FIR, up <-> down, the g_pad algebra of :110-115),
    and the double backward (= the forward op applied to the cotangent, :63-84) against the op itself and the CPU oracle."""
    from latent2im_amd import op
    gd = golden('upfirdn2d')
    rs = np.random.RandomState(4)
    for i, (n, c, h, w, up, down, p0, p1, gain) in enumerate(gd['cases']):
        up, down, p0, p1 = int(up), int(down), int(p0), int(p1)
        x = T(np.random.RandomState(20 + i).randn(int(n), int(c), int(h), int(w)).astype(np.float32))
        k = T(synth.fir_kernel(gain=gain))
        xd = x.to(DEV).requires_grad_(True)
        y = op.upfirdn2d(xd, k.to(DEV), up=up, down=down, pad=(p0, p1))
        close(y, T(gd['y_%d' % i]), 1e-5, 1e-6)
        gy = T(gd['gy_%d' % i]).to(DEV).requires_grad_(True)
        gx, = torch.autograd.grad(y, xd, gy, create_graph=True)
        assert gx.shape == x.shape
        close(gx, T(gd['gx_%d' % i]), 1e-5, 1e-6)
        v = T(rs.randn(*x.shape))
        gg, = torch.autograd.grad((gx * v.to(DEV)).sum(), gy)
        close(gg, sg2.upfirdn2d(v, k, up=up, down=down, pad=(p0, p1)), 1e-5, 1e-6)
        close(gg, kernels.upfirdn2d(v.to(DEV), k.to(DEV), (up, up), (down, down), (p0, p1, p0, p1)), 1e-6, 1e-7)


@pytest.mark.parametrize('cin,cout,pad,tr,h,w,b', [(64, 32, 0, True, 64, 64, 2), (128, 64, 0, True, 32, 32, 1), (32, 48, 0, True, 40, 36, 1), (512, 256, 0, True, 32, 32, 2),
                                                   (64, 64, 1, False, 64, 64, 2), (48, 96, 1, False, 66, 72, 1), (32, 64, 0, False, 65, 65, 2), (64, 128, 0, False, 129, 129, 1)])
def test_bf16x3_transposed_conv(cin, cout, pad, tr, h, w, b):
    """l2i_conv_transpose2d_bf16x3_f32 (all four output parities of a 3x3 stride-2 transposed conv from one staged tile, 3-term bf16 split)
    against float64: the generator's up layers (pad 0, style scale in, demod scale out) and the input-gradient of 3x3 stride-2 convs
    (pad 1: ResNet-50; pad 0: the discriminator's blurred down convs) with the activation-gradient mask."""
    rs = np.random.RandomState(cin + cout + h + pad)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    fc = conv.FrozenConv2d(wt, 2, pad, transposed=tr, device=DEV)
    g = lambda t: t.to(DEV)
    launched = []
    conv.PRECISION, conv.PROFILE = 'bf16x3', launched
    try:
        if tr:
            x, s, d = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cin) + 0.5), T(rs.rand(b, cout) + 0.5)
            ref = F.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.double().transpose(0, 1), stride=2, padding=pad) * d.double()[:, :, None, None] * 0.7
            got = fc.forward(g(x), in_scale=g(s), out_scale=g(d), out_gain=0.7)
        else:
            oh, ow = fc.out_hw(h, w)
            gy, y = T(rs.randn(b, cout, oh, ow)), T(rs.randn(b, cout, oh, ow))
            xr = T(rs.randn(b, cin, h, w)).double().requires_grad_(True)
            gm = gy.double() * torch.where(y > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64))
            ref, = torch.autograd.grad(F.conv2d(xr, wt.double(), stride=2, padding=pad), xr, gm)
            got = fc.dgrad(g(gy), (h, w), in_mask=g(y), mask=(1.0, 0.2))
    finally:
        conv.PRECISION, conv.PROFILE = 'f32', None
    torch.cuda.synchronize()
    assert [q[4] for q in launched] == ['l2i_conv_transpose2d_bf16x3_f32'], [q[4] for q in launched]
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_half_precision_entry_points_of_the_reference_ops():
    """l2i_fused_bias_act_f16 / l2i_upfirdn2d_f16 (the reference dispatches both ops for half too: fused_bias_act_kernel.cu:79,
    upfirdn2d_kernel.cu:225) against a scalar numpy float16 execution of the reference kernel bodies: every c10::Half binary op is the
    float op rounded to half, so the expected values are computed op by op in np.float16 — bit-exact comparison."""
    import ctypes
    from latent2im_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    f16, f32 = np.float16, np.float32
    hr = lambda a: np.asarray(a, dtype=f32).astype(f16).astype(f32)
    x = rs.randn(2, 5, 6, 7).astype(f16)
    b = rs.randn(5).astype(f16)
    ref = rs.randn(2, 5, 6, 7).astype(f16)
    xd, bd, rd = (torch.from_numpy(a).to(DEV) for a in (x, b, ref))
    ah, sh = hr(0.2), hr(2 ** 0.5)
    for act in (1, 3):
        for grad in (0, 1, 2):
            y = torch.empty_like(xd)
            _lib.check(lib.l2i_fused_bias_act_f16(_lib.ptr(y), _lib.ptr(xd), _lib.ptr(bd) if grad == 0 else None, _lib.ptr(rd) if grad == 1 else None,
                                                  x.size, 6 * 7, 5, act, grad, 0.2, 2 ** 0.5, _lib.stream_ptr()), 'l2i_fused_bias_act_f16')
            v = x.astype(f32)
            if grad == 0:
                v = hr(v + b.astype(f32)[None, :, None, None])
            code = act * 10 + grad
            if code in (10, 11):
                o = v
            elif code in (12, 32):
                o = np.zeros_like(v)
            elif code == 30:
                o = np.where(v > 0, v, hr(v * ah))
            else:
                o = np.where(ref.astype(f32) > 0, v, hr(v * ah))
            want = hr(o * sh).astype(f16)
            assert np.array_equal(y.cpu().numpy().view(np.uint16), want.view(np.uint16)), code
    # upfirdn2d: the three modes of the path (blur, upsample-by-2, downsample-by-2) + a padded / cropped case
    k = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0).astype(f16)
    for (h, w, up, down, p0, p1, gain) in ((9, 9, 1, 1, 1, 1, 4.0), (8, 6, 2, 1, 2, 1, 4.0), (8, 8, 1, 2, 1, 1, 1.0), (7, 10, 1, 1, 2, -1, 1.0)):
        kk = (k.astype(f32) * gain).astype(f16)
        xi = rs.randn(3, h, w).astype(f16)
        oh, ow = (h * up + p0 + p1 - 4) // down + 1, (w * up + p0 + p1 - 4) // down + 1
        y = torch.empty(3, oh, ow, dtype=torch.float16, device=DEV)
        xi_d, kk_d = torch.from_numpy(xi).to(DEV), torch.from_numpy(kk).to(DEV)          # (named: the caller owns the buffers until the kernel ran)
        _lib.check(lib.l2i_upfirdn2d_f16(_lib.ptr(y), _lib.ptr(xi_d), _lib.ptr(kk_d), 3, h, w, 4, 4,
                                         up, up, down, down, p0, p1, p0, p1, _lib.stream_ptr()), 'l2i_upfirdn2d_f16')
        kf = kk[::-1, ::-1].astype(f32)                               # sk[ky][kx] = kernel[kh-1-ky][kw-1-kx]
        want = np.zeros((3, oh, ow), dtype=f16)
        for m in range(3):
            for oy in range(oh):
                for ox in range(ow):
                    v = f32(0)
                    for ky in range(4):
                        uy = oy * down - p0 + ky
                        if uy < 0 or uy % up:
                            continue
                        for kx in range(4):
                            ux = ox * down - p0 + kx
                            if ux < 0 or ux % up:
                                continue
                            iy, ix = uy // up, ux // up
                            xv = f32(xi[m, iy, ix]) if (iy < h and ix < w) else f32(0)
                            v = f32(f16(v + xv * kf[ky, kx]))
                    want[m, oy, ox] = f16(v)
        assert np.array_equal(y.cpu().numpy().view(np.uint16), want.view(np.uint16)), (h, w, up, down)


@pytest.mark.parametrize('cin,cout,k,stride,pad,h,w,b', [(64, 64, 3, 1, 1, 16, 16, 4), (3, 64, 7, 2, 3, 64, 64, 2), (256, 64, 1, 1, 0, 8, 8, 4),
                                                         (40, 72, 3, 2, 1, 17, 19, 3), (128, 256, 1, 2, 0, 16, 16, 2), (512, 512, 3, 1, 1, 2, 2, 8)])
def test_conv_weight_gradient(cin, cout, k, stride, pad, h, w, b):
    """l2i_conv2d_wgrad_f32 against torch autograd (float64): every conv shape class of ResNet-50 (7x7 stem on 3 channels, 1x1, 3x3, strides
    1 / 2, channel counts that are not multiples of 32, tiny maps), and accumulation into a non-zero dw."""
    from latent2im_amd import regressor_train as RT
    rs = np.random.RandomState(cin + cout + k)
    x = T(rs.randn(b, cin, h, w))
    wt = T(rs.randn(cout, cin, k, k)).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, stride=stride, padding=pad)
    gy = T(rs.randn(*y.shape))
    ref, = torch.autograd.grad(y, wt, gy.double())
    dw = RT.conv_wgrad(x.to(DEV), gy.to(DEV), k, stride, pad)
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_batchnorm_training_kernels():
    """l2i_bn_stats / apply / bwd_reduce / bwd_apply through regressor_train._BN against torch's training-mode batch_norm + relu (+ residual)
    and autograd, including the running-statistics update."""
    from latent2im_amd import regressor_train as RT
    rs = np.random.RandomState(1)
    B, Cn, H, W = 4, 24, 9, 7
    x, res, gy = T(rs.randn(B, Cn, H, W) * 2 + 0.5), T(rs.randn(B, Cn, H, W)), T(rs.randn(B, Cn, H, W))
    P = {'bn.weight': rs.rand(Cn).astype(np.float32) + 0.5, 'bn.bias': rs.randn(Cn).astype(np.float32), 'bn.running_mean': rs.randn(Cn).astype(np.float32),
         'bn.running_var': rs.rand(Cn).astype(np.float32) + 0.5}
    bn = RT._BN(P, 'bn', DEV)
    xr = x.double().requires_grad_(True)
    gam, bet = T(P['bn.weight']).double().requires_grad_(True), T(P['bn.bias']).double().requires_grad_(True)
    rm, rv = T(P['bn.running_mean']).double(), T(P['bn.running_var']).double()
    yr = torch.relu(F.batch_norm(xr, rm, rv, gam, bet, training=True, momentum=0.1, eps=1e-5) + res.double())
    gx_r, gg_r, gb_r = torch.autograd.grad(yr, (xr, gam, bet), gy.double())
    y, saved = bn.forward(x.to(DEV), residual=res.to(DEV), relu=True)
    close(y, yr, 1e-5, 1e-5)
    close(bn.running_mean, rm, 1e-5, 1e-6)
    close(bn.running_var, rv, 1e-5, 1e-6)
    dx, dg, db, gm = bn.backward(gy.to(DEV), saved, out_mask=y, want_masked=True)
    close(dx, gx_r, 1e-4, 1e-5)
    close(dg, gg_r, 1e-4, 1e-4)
    close(db, gb_r, 1e-4, 1e-4)
    close(gm, gy.double() * (yr > 0), 0, 0)


@pytest.mark.parametrize('size,batch', [(32, 3), (256, 8), (1024, 16)])
def test_segmented_matvec_kernel_vs_table_emulation(size, batch):
    """l2i_segmented_matvec_f32 (every modulation / demodulation / ToRGB weight of a generator pass in two launches, d s and the latent
    gradient in two more) against a float64 numpy execution of the SAME segment tables (tests/emu.py; the tables themselves are checked
    against the layer-by-layer formulas in tests/test_modplan_cpu.py).  Batches above 8 take a second accumulator pass."""
    from latent2im_amd import generator
    from tests import emu
    G = generator.Generator(synth.generator_state(size, seed=100), size, device=DEV)
    plan = G.modplan
    rs = np.random.RandomState(size + batch)
    B = batch
    lat = T(rs.randn(B, G.n_latent, 512)).to(DEV)
    s_all, d_all, w_all = plan.forward(lat)
    red_dz, q_all, red_rgb = plan.reductions(B, DEV)
    red_dz.copy_(T(rs.randn(red_dz.numel())))
    q_all.copy_(T(rs.randn(q_all.numel())))
    red_rgb.copy_(T(rs.randn(red_rgb.numel())))
    g = plan.backward(B, s_all, d_all, red_dz, q_all, red_rgb)
    torch.cuda.synchronize()
    real = kernels.segmented_matvec
    try:
        kernels.segmented_matvec = emu.emulate_segmented_matvec
        cpu = lambda t: t.detach().cpu()
        plan_c = generator.Generator(synth.generator_state(size, seed=100), size, device='cpu').modplan
        s_c, d_c, w_c = plan_c.forward(cpu(lat))
        g_c = plan_c.backward(B, s_c, d_c, cpu(red_dz), cpu(q_all), cpu(red_rgb))
    finally:
        kernels.segmented_matvec = real
    close(s_all, s_c, 1e-5, 1e-5)
    close(d_all, d_c, 1e-5, 1e-6)
    close(w_all, w_c, 1e-5, 1e-5)
    assert float((g.cpu() - g_c).abs().max()) < 1e-5 * float(g_c.abs().max())


@pytest.mark.parametrize('cin,cout,h,w,b,mode', [(64, 64, 64, 64, 2, 'all'), (16, 40, 18, 72, 1, 'all'), (32, 64, 32, 40, 2, 'all'), (64, 64, 48, 128, 1, 'tall')])
def test_wino4_fused_maxpool_output(cin, cout, h, w, b, mode):
    """[r5] l2i_conv_params::pool_out / pool_idx: the position-split F(4x4) kernel also writes MaxPool2d(2, 2) of its output (VGG-19 pool1 after
    conv1_2, transform_base.py:426-454) from the 4x4 tile a lane holds: values and arg-max bytes BIT-IDENTICAL to l2i_maxpool2d_fwd_f32 on the
    stored map (ReLU-on-load launch with bias and the ContentLoss sum, as the product issues it; ties through a ReLU'd input included)."""
    from latent2im_amd import kernels
    rs = np.random.RandomState(cin + cout + h)
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x = T(np.round(rs.randn(b, cin, h, w) * 2) / 2).to(DEV)                 # coarse values: exact ties in the windows
    bias = T(rs.randn(cout)).to(DEV)
    fc = conv.FrozenConv2d(wt, 1, 1, device=DEV)
    prev = conv.WINO4
    conv.WINO4 = mode
    try:
        y0 = fc.forward(x, in_mask=x, mask=(1.0, 0.0), bias=bias)
        pool = (torch.full((b, cout, h // 2, w // 2), float('nan'), device=DEV), torch.full((b, cout, h // 2, w // 2), 255, device=DEV, dtype=torch.uint8), [False])
        ref = T(rs.randn(b, cout, h, w)).to(DEV)
        sq = (ref, torch.zeros(_lib.SQ_SLOTS, device=DEV), [False])
        y1 = fc.forward(x, in_mask=x, mask=(1.0, 0.0), bias=bias, sq=sq, pool=pool)
    finally:
        conv.WINO4 = prev
    torch.cuda.synchronize()
    assert pool[2][0] and sq[2][0] and torch.equal(y0, y1)
    want_p, want_i = kernels.maxpool2d_fwd(y1, 2, 2, 0)
    assert torch.equal(pool[0], want_p) and torch.equal(pool[1], want_i)
    # a constant map: every window is a four-way tie -> arg-max 0 everywhere
    z = torch.zeros(b, cin, h, w, device=DEV)
    conv.WINO4 = mode
    try:
        fc.forward(z, pool=pool)
    finally:
        conv.WINO4 = prev
    assert int(pool[1].max()) == 0 and float(pool[0].abs().max()) == 0.0


def test_abi5_fields_are_refused_by_the_entries_that_do_not_implement_them():
    """[r5] rgb_* belong to l2i_conv2d_h8, pool_* to l2i_conv2d_wino4_f32, in_h8 to the 7x7 kernel of l2i_conv_transpose2d_f32: every other conv entry
    returns L2I_E_UNSUPPORTED for a struct that carries them instead of silently ignoring the request."""
    lib = _lib.load()
    x = torch.randn(1, 8, 16, 64, device=DEV)
    fc = conv.FrozenConv2d(torch.randn(8, 8, 3, 3), 1, 1, device=DEV)
    L = fc.fwd[0]
    y = torch.empty(1, 8, 16, 64, device=DEV)
    buf = torch.zeros(1, 8, 8, 32, device=DEV)
    idx = torch.zeros(1, 8, 8, 32, device=DEV, dtype=torch.uint8)

    def params():
        p = _lib.ConvParams()
        p.x, p.w, p.y = _lib.fptr(x), _lib.fptr(L.w), _lib.fptr(y)
        p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = 1, 8, 16, 64, 8, L.w.shape[2]
        p.KH = p.KW = 3
        p.stride, p.pad_y, p.pad_x = 1, 1, 1
        p.OH, p.OW, p.OHf, p.OWf = 16, 64, 16, 64
        p.oy_step = p.ox_step = 1
        p.act_gain = p.out_gain = 1.0
        return p
    p = params()
    assert lib.l2i_conv2d_f32(p, _lib.stream_ptr()) == 0
    for field, val in (('rgb_w', _lib.fptr(buf)), ('pool_out', _lib.fptr(buf)), ('pool_idx', _lib.ptr(idx)), ('in_h8', 1)):
        p = params()
        setattr(p, field, val)
        assert lib.l2i_conv2d_f32(p, _lib.stream_ptr()) < 0, field
        assert lib.l2i_conv2d_wino_f32(p, _lib.stream_ptr()) < 0, field
    torch.cuda.synchronize()
