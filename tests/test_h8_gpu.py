"""The 16-bit path (BASELINE config 5): kernels on bf16 tensors in the channel-blocked h8 layout [B, C/8, H, W, 8] (include/l2i.h), through the
C ABI, against float64 torch on the SAME bf16-rounded operands.  Tolerances: the products and the fp32 accumulation are exact to fp32 rounding,
the only 16-bit rounding is the output's (relative 2^-9), so outputs are held to 2^-8 of their magnitude plus a small absolute term."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from latent2im_amd import _lib, conv

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(autouse=True, params=['bf16', 'f16'])
def elem(request):
    """[r5] Every KERNEL test of this file runs on both element types of the h8 path: bf16 (l2i_*_h8) and IEEE fp16 (l2i_*_h8_f16, the same sources
    compiled with another element type).  conv.PRECISION selects the type new h8 tensors / weight planes get; the references round with the same
    type.  The network- and step-level tests below carry their precision in their name and run once."""
    if request.param == 'f16' and request.node.name.startswith(('test_bf16_', 'test_fp16_', 'test_train_multi_attr_cli')):
        pytest.skip('named-precision test: runs once')
    old, conv.PRECISION = conv.PRECISION, request.param
    yield request.param
    conv.PRECISION = old


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def bf(x):
    return x.to(conv.h8_dtype()).to(torch.float64)


def close16(got, want, what=''):
    got, want = got.double().cpu(), want.double().cpu()
    tol = 2.0 ** -8 * want.abs() + 2e-3 * float(want.abs().max())
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - want).abs().max()), float(want.abs().max()))


H8_CASES = [  # cin, cout, k, stride, pad, transposed, h, w, batch
    (32, 32, 3, 1, 1, False, 16, 32, 1), (64, 64, 3, 1, 1, False, 37, 45, 2), (32, 40, 3, 1, 1, False, 9, 70, 1), (128, 96, 1, 1, 0, False, 33, 40, 2),
    (64, 128, 3, 2, 1, False, 33, 31, 2), (32, 64, 3, 2, 0, False, 68, 68, 1), (64, 64, 1, 2, 0, False, 40, 36, 2), (256, 512, 1, 2, 0, False, 16, 16, 1),
    (64, 32, 3, 2, 0, True, 32, 32, 2), (32, 64, 3, 2, 0, True, 17, 40, 1), (64, 64, 3, 2, 1, True, 20, 33, 1), (512, 512, 3, 1, 1, False, 8, 8, 2),
]


@pytest.mark.parametrize('case', H8_CASES)
def test_conv_h8_forward_and_dgrad(case):
    cin, cout, k, stride, pad, tr, h, w, b = case
    rs = np.random.RandomState(cin + 3 * cout + k + stride + h)
    wt = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    x = T(rs.randn(b, cin, h, w))
    xr = bf(x).requires_grad_(True)
    wr = bf(wt)
    ref = F.conv_transpose2d(xr, wr.transpose(0, 1), stride=2, padding=pad) if tr else F.conv2d(xr, wr, stride=stride, padding=pad)
    hc = conv.H8Conv(wt, stride, pad, transposed=tr, device=DEV)
    y = hc.forward(conv.to_h8(x.to(DEV), 32))
    torch.cuda.synchronize()
    assert y.dtype == conv.h8_dtype() and tuple(y.shape) == (b, (cout + 7) // 8, ref.shape[2], ref.shape[3], 8)
    close16(conv.from_h8(y, cout), ref.detach(), 'forward')
    if cout % 32 == 0 and not (k == 1 and stride == 2):       # the gradient's input tensor carries the conv's Cout channels: whole 32-channel chunks
                                                              # (a strided 1x1's gradient is a compact 1x1 conv + zero insertion: regressor16)
        gy = T(rs.randn(*ref.shape))
        gref, = torch.autograd.grad(ref, xr, bf(gy))
        gx = hc.dgrad(conv.to_h8(gy.to(DEV), 32), (h, w))
        close16(conv.from_h8(gx, cin), gref, 'dgrad')


IMG_CASES = [  # cin, cout, k, stride, pad, h, w, batch, act           (the three image-side convs of the 16-bit path, odd sizes, > 64 channels)
    (3, 64, 7, 2, 3, 64, 96, 2, 'relu'), (3, 32, 1, 1, 0, 40, 64, 2, 'lrelu'), (3, 128, 1, 1, 0, 33, 37, 1, 'lrelu'), (3, 64, 3, 1, 1, 32, 64, 2, 'none'),
    (3, 64, 3, 1, 1, 21, 36, 1, 'none'), (3, 64, 7, 2, 3, 31, 45, 1, 'relu'), (3, 64, 7, 2, 3, 256, 256, 1, 'relu'), (1, 40, 3, 1, 1, 19, 70, 2, 'relu'),
    (4, 96, 3, 1, 0, 18, 67, 1, 'lrelu'),
]


@pytest.mark.parametrize('case', IMG_CASES)
def test_conv_img_h8(case):
    """[r5] l2i_conv_img_h8 (ResNet-50 stem, discriminator from-RGB, VGG conv1_1 of the 16-bit path: transform_base.py:396-403, networks.py:568-575,
    transform_base.py:426-454): fp32 image in, h8 out, against float64 torch on the operands rounded to the element type (the kernel rounds them
    itself); the fused (y - ref)^2 sum against float64 on the rounded output."""
    from latent2im_amd import kernels16 as K16
    cin, cout, k, stride, pad, h, w, b, act = case
    rs = np.random.RandomState(cin + cout + k + h)
    wt, bias = T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)), T(rs.randn(cout))
    x = T(rs.randn(b, cin, h, w))
    ic = conv.ImgConvH8(wt, stride, pad, device=DEV)
    gain, og = 2.0 ** 0.5, (1.0 if act != 'none' else 0.75)
    ref = F.conv2d(bf(x), bf(wt), stride=stride, padding=pad) + bias.double().reshape(1, -1, 1, 1)
    ref = {'relu': torch.relu(ref), 'lrelu': F.leaky_relu(ref, 0.2) * gain, 'none': ref}[act] * og
    ref16 = K16.cast_to_h8(T(rs.randn(*ref.shape)).to(DEV), dtype=conv.h8_dtype())
    sq = (ref16, torch.zeros(_lib.SQ_SLOTS, device=DEV), [False])
    got = ic.forward(x.to(DEV), bias=bias.to(DEV), act={'relu': conv.ACT_RELU, 'lrelu': conv.ACT_LRELU, 'none': conv.ACT_NONE}[act], slope=0.2, gain=gain, out_gain=og, sq=sq)
    torch.cuda.synchronize()
    assert got.dtype == conv.h8_dtype() and tuple(got.shape) == (b, cout // 8, ref.shape[2], ref.shape[3], 8)
    close16(conv.from_h8(got, cout), ref, 'image conv')
    exact = float(((got.double() - ref16.double()) ** 2).sum())
    assert sq[2][0] and abs(float(sq[1].sum()) - exact) <= 1e-4 * exact


def test_stem_input_gradient_from_h8():
    """[r5] l2i_conv_params::in_h8: the 7x7 / stride-2 stem's input gradient reads the 16-bit pool gradient and ReLU mask (regressor backward,
    transform_base.py:396-403).  Same FMAs on the same values as the fp32 launch of the same kernel on the casts: bit-identical."""
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(3)
    wt = T(rs.randn(64, 3, 7, 7) / np.sqrt(147))
    fc = conv.FrozenConv2d(wt, 2, 3, device=DEV)
    for b, h, w in ((2, 64, 96), (1, 38, 52)):
        oh, ow = fc.out_hw(h, w)
        g16 = K16.cast_to_h8(T(rs.randn(b, 64, oh, ow)).to(DEV), dtype=conv.h8_dtype())
        m16 = K16.cast_to_h8(T(rs.randn(b, 64, oh, ow)).to(DEV).clamp_(min=0), dtype=conv.h8_dtype())
        want = fc.dgrad(K16.cast_from_h8(g16), (h, w), in_mask=K16.cast_from_h8(m16), mask=(1.0, 0.0), out_gain=0.5)
        got = fc.dgrad_h8in(g16, (h, w), in_mask=m16, mask=(1.0, 0.0), out_gain=0.5)
        torch.cuda.synchronize()
        if ow % 4 == 0:                                       # (odd widths: the fp32 launch is four per-parity launches, another summation order)
            assert torch.equal(got, want), float((got - want).abs().max())
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        ref = torch.autograd.grad(F.conv2d(xr := torch.zeros(b, 3, h, w, dtype=torch.float64, requires_grad=True), wt.double(), stride=2, padding=3), xr,
                                  (g16.double().cpu().permute(0, 1, 4, 2, 3).reshape(b, 64, oh, ow) * (m16.double().cpu().permute(0, 1, 4, 2, 3).reshape(b, 64, oh, ow) > 0)))[0] * 0.5
        assert float((got.double().cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize('cout, cin, h, w', [(32, 64, 24, 40), (64, 64, 17, 70), (32, 32, 40, 33)])
def test_conv_h8_fused_torgb(cout, cin, h, w):
    """[r5] l2i_conv_params::rgb_w: the modulated 3x3 conv of the 512^2 / 1024^2 generator blocks also writes the ToRGB image of its output
    (networks.py:349-358 after 503-505): rgb = bias + sum_c wmod[b, o, c] * y[c] with y the conv's epilogue value (demodulation, noise, bias, leaky
    ReLU) in fp32, before the 16-bit store; the h8 output itself is unchanged (bit-identical to the launch without rgb_w)."""
    rs = np.random.RandomState(cout + h)
    b = 2
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, d, bias, nz = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cout) + 0.5), T(rs.randn(cout)), T(rs.randn(b, 1, h, w))
    wmod, rb = T(rs.randn(b, 3, cout) / np.sqrt(cout)), T(rs.randn(3))
    g = lambda t: t.to(DEV)
    hc = conv.H8Conv(wt, 1, 1, device=DEV)
    xh = conv.to_h8(g(x), 32)
    kw = dict(out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    y0 = hc.forward(xh, **kw)
    rgb = torch.full((b, 3, h, w), float('nan'), device=DEV)
    y1 = hc.forward(xh, rgb=(g(wmod), g(rb), rgb), **kw)
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    ref = F.leaky_relu(F.conv2d(bf(x), bf(wt), padding=1) * d.double()[:, :, None, None] + nz.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    want = torch.einsum('boc,bchw->bohw', wmod.double(), ref) + rb.double()[None, :, None, None]
    assert float((rgb.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    with pytest.raises(_lib.L2IError):                       # the lean epilogue only: a per-pixel operand map is refused, not silently dropped
        hc.forward(xh, rgb=(g(wmod), g(rb), rgb), out_mask=y0, **kw)
    # [r6] Two things this test pins on purpose.  (1) The fused image is formed from the fp32 epilogue values BEFORE the 16-bit store, while the unfused
    # l2i_torgb_fwd_h8 and the backward (l2i_sg2_act_bwd_h8: sum_o wmod * grgb against the stored y) see the rounded map: forward and backward of the
    # ToRGB branch differ by one 16-bit rounding of y, inside the 16-bit step contract (2^-9 / 2^-12 relative per element), which is why `want` above is
    # built from the unrounded reference.  (2) Cout must be a whole number of 32-channel tiles: the epilogue fetches rgb_w rows for every channel group
    # of the tile before masking (a 40-channel layer would read past its row).
    if cout == 32:
        hc40 = conv.H8Conv(T(rs.randn(40, cin, 3, 3)), 1, 1, device=DEV)
        with pytest.raises((AssertionError, _lib.L2IError)):
            hc40.forward(xh, rgb=(g(T(rs.randn(b, 3, 40))), g(rb), rgb))


def test_conv_h8_epilogue_fusions_and_fp32_output():
    rs = np.random.RandomState(5)
    b, cin, cout, h, w = 2, 64, 64, 24, 40
    wt = T(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    x, d, bias = T(rs.randn(b, cin, h, w)), T(rs.rand(b, cout) + 0.5), T(rs.randn(cout))
    nz, res, rmk, omk, rsub = T(rs.randn(b, 1, h, w)), T(rs.randn(b, cout, h, w)), T(rs.randn(b, cout, h, w)), T(rs.randn(b, cout, h, w)), T(rs.randn(b, cout, h, w))
    g = lambda t: t.to(DEV)
    H = lambda t: conv.to_h8(g(t), 32)
    hc = conv.H8Conv(wt, 1, 1, device=DEV)
    c64 = F.conv2d(bf(x), bf(wt), padding=1)
    # modulated-conv epilogue: demod scale, noise, bias, leaky ReLU * sqrt(2)
    y1 = hc.forward(H(x), out_scale=g(d), noise=g(nz), noise_w=0.3, bias=g(bias), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    ref1 = F.leaky_relu(c64 * d.double()[:, :, None, None] + nz.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    close16(conv.from_h8(y1, cout), ref1, 'modconv epilogue')
    # ResNet-style: bias + masked residual + ReLU
    y2 = hc.forward(H(x), bias=g(bias), residual=H(res), res_mask=H(rmk), act=conv.ACT_RELU)
    ref2 = torch.relu(c64 + bias.double()[None, :, None, None] + torch.where(bf(rmk) > 0, bf(res), torch.zeros_like(bf(res))))
    close16(conv.from_h8(y2, cout), ref2, 'residual epilogue')
    # gradient conv: output mask, ContentLoss direct term res_coef * coef_dev * (residual - res_sub), gain
    cdev = torch.full((1,), 0.7, device=DEV)
    y3 = hc.forward(H(x), out_mask=H(omk), residual=H(res), res_sub=H(rsub), res_coef=0.25, res_coef_dev=cdev, out_gain=0.5)
    ref3 = (torch.where(bf(omk) > 0, c64, torch.zeros_like(c64)) + 0.25 * 0.7 * (bf(res) - bf(rsub))) * 0.5
    close16(conv.from_h8(y3, cout), ref3, 'gradient epilogue')
    # ContentLoss value in the epilogue: sum (y - ref)^2 on the rounded output
    sq_acc, fused = torch.zeros(_lib.SQ_SLOTS, device=DEV), [False]
    y4 = hc.forward(H(x), bias=g(bias), sq=(H(res), sq_acc, fused))
    want = float(((conv.from_h8(y4, cout).double().cpu() - bf(res)) ** 2).sum())
    assert fused[0] and abs(float(sq_acc.double().sum()) - want) <= 1e-4 * want
    # fp32 NCHW output onto 3 channels (a gradient landing on an image) and its fused terms
    w3 = T(rs.randn(3, cin, 3, 3) / np.sqrt(cin * 9))
    h3 = conv.H8Conv(w3, 1, 1, device=DEV)
    y5 = h3.forward(H(x), out_f32=True, out_gain=2.0)
    assert y5.dtype == torch.float32 and tuple(y5.shape) == (b, 3, h, w)
    ref5 = F.conv2d(bf(x), bf(w3), padding=1) * 2.0
    assert float((y5.double().cpu() - ref5).abs().max()) < 1e-4 * float(ref5.abs().max())           # fp32 output: no 16-bit rounding at all
    # per-sample weight planes (the generator's modulated convs): sample i uses plane set i
    s = T(rs.rand(b, cin) + 0.5)
    planes = torch.stack([conv.pack_weight_h8(wt * s[i][None, :, None, None]) for i in range(b)]).to(DEV)
    y6 = hc.forward(H(x), planes=planes, w_bstride=planes[0].numel() * 2)
    ref6 = torch.stack([F.conv2d(bf(x[i:i + 1]), bf(wt * s[i][None, :, None, None]), padding=1)[0] for i in range(b)])
    close16(conv.from_h8(y6, cout), ref6, 'per-sample weights')


# ---- streaming companions (csrc/l2i_stream_h8.hip) against torch on the bf16-rounded operands ---------------------------------------------------
def rb(x):
    """what an h8 tensor holds of x."""
    return x.to(conv.h8_dtype()).float()


def test_h8_layout_casts():
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(0)
    for B, C, H, W, cpad in ((2, 3, 9, 11, 32), (1, 40, 5, 70, 40), (2, 64, 16, 16, 64)):
        x = T(rs.randn(B, C, H, W)).to(DEV)
        t = K16.cast_to_h8(x, cpad)
        assert torch.equal(t, conv.to_h8(x, cpad))                                     # == the torch reshuffle (round to nearest even)
        assert torch.equal(K16.cast_from_h8(t, C), rb(x))


def test_h8_upfirdn2d_all_path_geometries():
    from latent2im_amd import kernels16 as K16
    from oracle import sg2
    rs = np.random.RandomState(1)
    k1 = np.asarray([1.0, 3.0, 3.0, 1.0])
    k = T(np.outer(k1, k1) / 64.0)
    for (B, C, H, W), up, down, pad in (((2, 16, 33, 33), 1, 1, (1, 1)),           # generator up-layer blur on the (2H+1)^2 map
                                        ((1, 8, 36, 36), 1, 1, (1, -2)),            # ... produced (2H+4)^2, cropped by the negative far pad
                                        ((2, 8, 16, 16), 1, 1, (2, 2)),             # discriminator blur in front of a stride-2 3x3 conv
                                        ((1, 16, 20, 20), 1, 2, (1, 1)),            # discriminator skip: blur + every second pixel
                                        ((1, 8, 10, 10), 2, 1, (2, 1)),             # its adjoint family: zero insertion
                                        ((1, 8, 17, 17), 1, 1, (2, 5))):            # generator backward: the blur gradient padded to (2H+4)^2
        x = T(rs.randn(B, C, H, W))
        kk = k * (up * up)
        ref = sg2.upfirdn2d(rb(x).double(), kk.double(), up=up, down=down, pad=pad)
        for sep in (None, K16.separable(kk)):                                             # the generic kernel, and the separable one where it applies
            y = K16.upfirdn2d(conv.to_h8(x.to(DEV)), kk.to(DEV), up=up, down=down, pad=(pad[0], pad[1], pad[0], pad[1]), sep=sep)
            close16(conv.from_h8(y, C), ref, 'upfirdn %s %s' % (pad, sep is not None))
    # a tall map: several 16-row bands per column, ragged last band
    x = T(rs.randn(1, 8, 70, 19))
    y = K16.upfirdn2d(conv.to_h8(x.to(DEV)), k.to(DEV), pad=(2, 2, 2, 2), sep=K16.separable(k))
    close16(conv.from_h8(y, 8), sg2.upfirdn2d(rb(x).double(), k.double(), pad=(2, 2)), 'upfirdn tall')
    # the discriminator's skip pair on maps spanning several 62-column chunks and 8-row bands (ragged ends): blur + every second pixel, and its
    # adjoint (zero insertion + blur) with the skip sum as addend
    x = T(rs.randn(1, 8, 140, 150))
    y = K16.upfirdn2d(conv.to_h8(x.to(DEV)), k.to(DEV), down=2, pad=(1, 1, 1, 1), sep=K16.separable(k))
    close16(conv.from_h8(y, 8), sg2.upfirdn2d(rb(x).double(), k.double(), down=2, pad=(1, 1)), 'upfirdn down 2, wide')
    x, add = T(rs.randn(2, 8, 70, 75)), T(rs.randn(2, 8, 140, 150))
    kk = k * 4
    for a in (None, add):
        y = K16.upfirdn2d(conv.to_h8(x.to(DEV)), kk.to(DEV), up=2, pad=(2, 1, 2, 1), addend=None if a is None else conv.to_h8(a.to(DEV)), sep=K16.separable(kk))
        ref = sg2.upfirdn2d(rb(x).double(), kk.double(), up=2, pad=(2, 1)) + (0 if a is None else rb(a).double())
        close16(conv.from_h8(y, 8), ref, 'upfirdn up 2, wide, addend %s' % (a is not None))
    # fused epilogue of the generator's up layers: noise, bias, leaky ReLU * sqrt(2)
    x, nz, bias = T(rs.randn(2, 16, 33, 33)), T(rs.randn(2, 1, 32, 32)), T(rs.randn(16))
    kk = k * 4
    ref = F.leaky_relu(sg2.upfirdn2d(rb(x).double(), kk.double(), pad=(1, 1)) + 0.3 * nz.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    msk, add = T(rs.randn(2, 16, 32, 32)), T(rs.randn(2, 16, 32, 32))
    ref2 = ref * torch.where(rb(msk).double() > 0, 1.5, 0.25) + rb(add).double()
    for sep in (None, K16.separable(kk)):
        y = K16.upfirdn2d(conv.to_h8(x.to(DEV)), kk.to(DEV), pad=(1, 1, 1, 1), noise=nz.to(DEV), noise_w=0.3, bias=bias.to(DEV), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5, sep=sep)
        close16(conv.from_h8(y, 16), ref, 'upfirdn epilogue')
        y2 = K16.upfirdn2d(conv.to_h8(x.to(DEV)), kk.to(DEV), pad=(1, 1, 1, 1), noise=nz.to(DEV), noise_w=0.3, bias=bias.to(DEV), act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5,
                           mask=conv.to_h8(msk.to(DEV)), mask_vals=(1.5, 0.25), addend=conv.to_h8(add.to(DEV)), sep=sep)
        close16(conv.from_h8(y2, 16), ref2, 'upfirdn epilogue + mask + addend')


def test_h8_torgb_act_bwd_reductions_sqdiff():
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(2)
    B, C, H = 2, 24, 20
    x, wm, bias = T(rs.randn(B, C, H, H)), T(rs.randn(B, 3, C)), T(rs.randn(3))
    g = lambda t: t.to(DEV)
    H8 = lambda t: conv.to_h8(g(t))
    rgb = K16.torgb_fwd(H8(x), g(wm), g(bias))
    ref = torch.einsum('bchw,boc->bohw', rb(x).double(), wm.double()) + bias.double()[None, :, None, None]
    assert float((rgb.double().cpu() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    y, gin, gs, grgb = T(rs.randn(B, C, H, H)), T(rs.randn(B, C, H, H)), T(rs.rand(B, C) + 0.5), T(rs.randn(B, 3, H, H))
    cb, nz = T(rs.randn(C)), T(rs.randn(B, 1, H, H))
    red, red_rgb, red_q = torch.zeros(B, C, device=DEV), torch.zeros(B, C, 3, device=DEV), torch.zeros(B * C, device=DEV)
    dz = K16.sg2_act_bwd(H8(y), H8(gin), g(gs), g(grgb), g(wm), g(cb), g(nz), 0.3, 0.2, 2 ** 0.5, red, red_rgb, red_q=red_q)
    yb, gb = rb(y).double(), rb(gin).double()
    # [r5] red_gin_y: the next layer's style gradient sum_p gin * y (what l2i_dot_reduce_h8 formed in a separate pass)
    assert float((red_q.view(B, C).double().cpu() - (gb * yb).sum((2, 3))).abs().max()) < 1e-4 * float((gb * yb).abs().sum((2, 3)).max())
    gg = gb * gs.double()[:, :, None, None] + torch.einsum('bohw,boc->bchw', grgb.double(), wm.double())
    dz_ref = gg * torch.where(yb > 0, torch.tensor(2 ** 0.5, dtype=torch.float64), torch.tensor(0.2 * 2 ** 0.5, dtype=torch.float64))
    zpre = torch.where(yb > 0, yb / 2 ** 0.5, yb / (0.2 * 2 ** 0.5)) - cb.double()[None, :, None, None] - 0.3 * nz.double()
    close16(conv.from_h8(dz, C), dz_ref, 'dz')
    assert float((red.double().cpu() - (dz_ref * zpre).sum((2, 3))).abs().max()) < 1e-4 * float((dz_ref * zpre).abs().sum((2, 3)).max())
    want = torch.einsum('bchw,bohw->bco', yb, grgb.double())
    assert float((red_rgb.double().cpu() - want).abs().max()) < 1e-4 * float(torch.einsum('bchw,bohw->bco', yb.abs(), grgb.double().abs()).max())
    # without the ToRGB branch / without an incoming gradient
    red2 = torch.zeros(B, C, device=DEV)
    dz2 = K16.sg2_act_bwd(H8(y), H8(gin), g(gs), None, None, g(cb), None, 0.0, 0.2, 2 ** 0.5, red2, None)
    close16(conv.from_h8(dz2, C), gb * gs.double()[:, :, None, None] * torch.where(yb > 0, 2 ** 0.5, 0.2 * 2 ** 0.5), 'dz without rgb')
    d = K16.dot_reduce(H8(x), H8(y))
    want = (rb(x).double() * yb).sum((2, 3))
    assert float((d.double().cpu() - want).abs().max()) < 1e-4 * float((rb(x).double() * yb).abs().sum((2, 3)).max())
    s, gr = K16.sqdiff(H8(x), H8(y), coef=0.25, want_grad=True)
    assert abs(float(s) - float(((yb - rb(x).double()) ** 2).sum())) < 1e-4 * float(s)
    close16(conv.from_h8(gr, C), 0.25 * (yb - rb(x).double()), 'sqdiff grad')


@pytest.mark.parametrize('k,s,pad,h,w', [(3, 2, 1, 16, 16), (2, 2, 0, 16, 24), (3, 2, 1, 15, 21), (2, 2, 0, 40, 8), (3, 2, 1, 9, 72)])
def test_h8_maxpool_forward_backward(k, s, pad, h, w):
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(k + h)
    x = T(np.round(rs.randn(2, 16, h, w) * 4) / 4)                   # coarse values: ties inside windows
    xr = rb(x).double().requires_grad_(True)
    ref = F.max_pool2d(xr, k, s, pad)
    y, idx = K16.maxpool2d_fwd(conv.to_h8(x.to(DEV)), k, s, pad)
    assert torch.equal(conv.from_h8(y, 16).cpu().double(), ref.detach())
    gy = T(rs.randn(*ref.shape))
    gref, = torch.autograd.grad(ref, xr, rb(gy).double())
    gx = K16.maxpool2d_bwd(conv.to_h8(gy.to(DEV)), idx, (h, w), k, s, pad)
    close16(conv.from_h8(gx, 16), gref, 'pool backward')               # first maximum in row-major order, like ATen
    yr, _ = K16.maxpool2d_fwd(conv.to_h8(x.to(DEV)), k, s, pad, relu=True)
    assert torch.equal(conv.from_h8(yr, 16).cpu().double(), torch.relu(ref.detach()))
    if k == 2:                                                          # fused ContentLoss term of the pooled tap
        a, b = T(rs.randn(2, 16, h, w)), T(rs.randn(2, 16, h, w))
        cd = torch.full((1,), 0.5, device=DEV)
        gx2 = K16.maxpool2d_bwd(conv.to_h8(gy.to(DEV)), idx, (h, w), k, s, pad, a=conv.to_h8(a.to(DEV)), b=conv.to_h8(b.to(DEV)), coef=0.3, coef_dev=cd)
        close16(conv.from_h8(gx2, 16), gref + 0.15 * (rb(b).double() - rb(a).double()), 'pool backward + diff')


def test_h8_zero_insert_and_modulated_planes():
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(4)
    y, c = T(rs.randn(2, 16, 10, 14)), T(rs.randn(2, 16, 5, 7))
    yh = conv.to_h8(y.to(DEV))
    K16.add_zero_insert(yh, conv.to_h8(c.to(DEV)))
    want = rb(y).double()
    want[:, :, ::2, ::2] += rb(c).double()
    close16(conv.from_h8(yh, 16), want, 'zero insert')
    w = T(rs.randn(40, 64, 3, 3))
    s = T(rs.rand(3, 64) + 0.5)
    planes = K16.modulate_planes(conv.pack_weight_h8_f32(w).to(DEV), s.to(DEV))
    for i in range(3):
        assert torch.equal(planes[i].cpu(), conv.pack_weight_h8(w * s[i][None, :, None, None]))


# ---- the networks and the whole step on the 16-bit path against the FLOAT64 oracle ----------------------------------------------------------------
# Tolerance contract of BASELINE config 5.  bf16 storage rounds every feature map to 2^-9 relative, so the fp32 bars of the
# north star (rtol 1e-3 / atol 1e-4) do not apply to images and gradients.  Two kinds of bound, kept apart since round 4:
#   (A) KERNEL CONSISTENCY — against an oracle that restates the path's storage rounding (oracle/nets16.py, ResNet-50: the network whose gradient
#       is hurt most).  A network with bf16 feature maps is chaotic in its rounding decisions: that oracle against ITSELF with values perturbed by
#       1e-7 before rounding (an fp32 summation-order difference) agrees only to an output error of 5e-4, a gradient cosine of 0.992 and a relative
#       L2 of 0.125 (the "floor", measured in the same call).  The kernels are held to 1.35 x that floor — measured AT the floor (0.992 / 0.125 at
#       64^2, 0.990 / 0.138 at 256^2).  Element-exact checks of every kernel are test_conv_h8_* above (2^-8 relative per element).
#   (B) PRICE OF THE FORMAT — against the exact float64 oracle (measured by tools/bf16_study.py into profiles/r04_bf16_tolerance.json; bounds =
#       measured value with a real margin, at 64^2 AND 256^2):
#         images <= 4e-2 of the largest pixel (measured 0.8 - 1.9e-2); regressor outputs / alpha_org <= 2e-3 absolute (2.4 - 5.9e-4);
#         total and regressor loss <= 5e-3 relative (1e-5 - 4e-4), GAN term <= 2e-2 (4e-4 - 1.3e-2); per-attribute regressor loss <= 1e-3 absolute
#         (the north star's "<= 1e-3 per-attr regressor-loss delta": 1.6 - 6.5e-4);
#         gradient cosines: generator latent >= 0.997 (0.9985), discriminator >= 0.99 (0.995 - 0.997), VGG >= 0.99 (0.995 - 0.996), ResNet-50 >= 0.94
#         (0.972 at 64^2, 0.953 at 256^2 — and the bf16-storage oracle itself sits at 0.95 - 0.98 against the exact one: fifty ReLU layers whose
#         masks flip where a rounded feature map crosses zero; the fp32 residual trunk of the round-3 review was built, changed nothing and was removed);
#         walk gradient of the whole step: cosine >= 0.97, relative L2 <= 0.27 (0.977 - 0.997 / 0.09 - 0.25).
# The content term is the mean squared DIFFERENCE of two feature maps that are each rounded to bf16: at walk initialisation (|w| ~ 0.02) the true
# difference is below the rounding step, so its relative error is unbounded by construction and only its absolute size is held.
def _study():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bf16_study', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'bf16_study.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize('size', [64, 256])
def test_bf16_networks_vs_float64_oracle(size):
    r = _study().networks(size, 2)
    print(r)
    # (A) kernels: ResNet-50 against the oracle with the same storage rounding
    assert r['R16_out_relmax'] < 2e-3 and r['R16_grad_cos'] > 0.985, {k: v for k, v in r.items() if k.startswith('R16')}
    assert r['R16_grad_l2'] < 1.35 * r['R16floor_grad_l2'] + 0.01, {k: v for k, v in r.items() if k.startswith('R16')}
    # (B) the format, against the exact oracle
    assert r['G_img_relmax'] < 2.5e-2 and r['G_grad_cos'] > 0.997 and r['G_grad_l2'] < 0.08
    assert r['R_out_relmax'] < 4e-3 and r['R_grad_cos'] > 0.94 and r['R_grad_l2'] < 0.36
    assert r['R16fmt_grad_cos'] < 0.995                     # the storage rounding alone explains the distance (if this ever fails, tighten R_grad_cos)
    assert r['D_out_relmax'] < 1e-2 and r['D_grad_cos'] > 0.99 and r['D_grad_l2'] < 0.13
    assert max(r['V_loss_rel']) < 4e-3 and r['V_grad_cos'] > 0.99 and r['V_grad_l2'] < 0.13      # (content terms: 3e-4 at 64^2, 1.9e-3 at 256^2)


def _cached_step(precision, size, batch, hipgraph=False):
    """tools/bf16_study.py:step for BASELINE config 5's flow against the float64 oracle evaluation of tests/golden/oracle_1024.npz (cases s64 / s256 /
    s1024: the same seeds; [r5] the oracle used to be re-evaluated on every run, 8 - 36 s per case; [r6] s1024b8: the per-GPU batch bench.py runs,
    float32 oracle — 1e-3 of the float64 one in relative L2 on the cases that hold both)."""
    import gc
    from latent2im_amd import conv
    from tests import oracle_cache
    from tests.conftest import GOLDEN
    st = _study()
    name = 's%d' % size + ('b%d' % batch if batch == 8 else '')
    case = oracle_cache.CASES[name]
    assert (case['batch'], case['z_seed']) == (batch, 21)
    old, st.PRECISION = conv.PRECISION, precision
    try:
        return st.step(size, batch, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], True, 'scene', cached=oracle_cache.load(GOLDEN, name), hipgraph=hipgraph)
    finally:
        conv.PRECISION = old
        gc.collect()
        torch.cuda.empty_cache()


@pytest.mark.parametrize('size,batch', [(64, 4), (256, 2), (1024, 1)])
def test_bf16_training_step_vs_float64_oracle(size, batch):
    """BASELINE config 5's step (SceneGraph, five attributes, clamp flow) on the 16-bit path at 64^2, 256^2 and 1024^2 against the float64 oracle."""
    r = _cached_step('bf16', size, batch)
    print(r)
    assert r['x0_relmax'] < 4e-2 and r['x1_relmax'] < 4e-2
    assert r['a0_absmax'] < 2e-3 and r['eps_absmax'] < 2e-3
    assert r["loss_rel"] < 5e-3 and r["reg_rel"] < 5e-3 and r["gan_rel"] < 2e-2          # (the GAN term passes nine bf16 residual blocks at 1024^2: measured 8e-3 there)
    assert max(r['per_attr_reg_loss_delta']) < 1e-3
    assert r['grad_cos'] > 0.97 and r['grad_l2'] < 0.27, (r['grad_cos'], r['grad_l2'])


@pytest.mark.parametrize('size,batch', [(64, 4), (256, 2), (1024, 1)])
def test_fp16_training_step_vs_float64_oracle(size, batch):
    """[r5] The same step with IEEE fp16 h8 maps (conv.PRECISION = 'f16': BASELINE configs[4] says "fp16 MFMA") and the static power-of-two gradient
    scales of nets16.loss_scale_for.  Three more mantissa bits than bf16: the contract is an order of magnitude tighter — measured (profiles/
    r05_fp16_tolerance.json) images 1.1e-3 / 1.4e-3 / 2.2e-3 of the largest pixel, alpha 5e-5, losses 1e-4, GAN term 1e-3, per-attribute regressor
    loss 5e-5, walk gradient cosine 0.9997 / 0.9999 / 0.9999 and relative L2 0.025 / 0.017 / 0.031 (bf16: 0.997 / 0.993 / 0.983 and 0.09 / 0.12 / 0.19)."""
    r = _cached_step('f16', size, batch)
    print(r)
    assert r['finite']
    assert r['x0_relmax'] < 6e-3 and r['x1_relmax'] < 6e-3
    assert r['a0_absmax'] < 2e-4 and r['eps_absmax'] < 2e-4
    assert r["loss_rel"] < 5e-4 and r["reg_rel"] < 5e-4 and r["gan_rel"] < 4e-3
    assert max(r['per_attr_reg_loss_delta']) < 2e-4
    assert r['grad_cos'] > 0.998 and r['grad_l2'] < 0.06, (r['grad_cos'], r['grad_l2'])


@pytest.mark.parametrize('hipgraph', [False, True], ids=['eager', 'hipgraph'])
def test_fp16_config5_as_benched_1024_batch8_vs_oracle(hipgraph):
    """[r6] BASELINE config 5 EXACTLY as `bench.py --config c5` times it — SceneGraph, five scene attributes, clamp flow, IEEE fp16 h8 maps with the
    gradient scales of nets16.loss_scale_for(1024, 8), 1024^2, per-GPU batch 8, launched eagerly and REPLAYED from the hipGraph — against the oracle
    (cached case s1024b8).  The exponents depend on batch and resolution, so this is the leg that checks the batch-8 values for range; same contract as
    test_fp16_training_step_vs_float64_oracle (the oracle here is its float32 evaluation: within 1e-3 relative L2 of float64 on every case that holds
    both, against a bound of 0.06)."""
    r = _cached_step('f16', 1024, 8, hipgraph=hipgraph)
    print(r)
    assert r['finite'] and r['hipgraph'] == hipgraph
    assert r['x0_relmax'] < 6e-3 and r['x1_relmax'] < 6e-3
    assert r['a0_absmax'] < 2e-4 and r['eps_absmax'] < 2e-4
    assert r["loss_rel"] < 5e-4 and r["reg_rel"] < 5e-4 and r["gan_rel"] < 4e-3
    assert max(r['per_attr_reg_loss_delta']) < 2e-4
    assert r['grad_cos'] > 0.998 and r['grad_l2'] < 0.06, (r['grad_cos'], r['grad_l2'])


def test_fp16_hipgraph_captured_steps_follow_eager_steps():
    """[r6] tests/test_networks_gpu.py::test_hipgraph_captured_step_matches_eager on fp16 elements: three optimiser steps (GuardedAdam, dynamic loss
    scale) replayed from the graph against the same steps launched eagerly.  Same kernels on the same inputs: only the fp32 atomics of the reductions
    reorder, and 16-bit rounding may turn that into a one-step difference of an activation."""
    from latent2im_amd import capture, constants, selfcheck, synth
    old = conv.PRECISION
    try:
        conv.PRECISION = 'f16'
        attrs = ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk']
        size, batch = 64, 4
        rs = np.random.RandomState(3)
        zs = [synth.z_sample(batch, seed=40 + i) for i in range(3)]
        al = [np.ones((batch, 5)) * rs.uniform(-1, 1, 5) for _ in range(3)]
        ge = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        assert type(ge.optimizers).__name__ == 'GuardedAdam' and ge.loss_scaler is not None
        eager = [selfcheck.run_step(ge, zs[i], al[i], clamp=True) for i in range(3)]
        gc_ = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        step = capture.CapturedStep(gc_, batch, 5, clamp=True)
        for i in range(3):
            r = step(zs[i], al[i])
            torch.cuda.synchronize()
            e = eager[i]
            assert float((r['x1'].float() - e['x1'].float()).abs().max()) <= 2.0 ** -8 * float(e['x1'].abs().max())
            assert abs(float(r['loss']) - float(e['loss'])) <= 1e-4 * abs(float(e['loss']))
            g, ge_ = r['grad'].float(), e['grad'].float()
            cos = float((g * ge_).sum() / (g.norm() * ge_.norm()))
            assert cos > 0.9995, (i, cos)
        w0 = torch.from_numpy(synth.walk_init(5, 10, seed=7))
        assert not torch.equal(gc_.walk.w.detach().cpu(), w0)
        assert float((gc_.walk.w - ge.walk.w).abs().max()) <= 2.5e-3          # three Adam steps of lr 1e-3: entries whose gradient is rounding noise may step apart
        st = gc_.loss_scaler.stats()
        assert st['steps'] == 3 and st['skipped'] == 0 and st['scale'] == 1.0, st
    finally:
        conv.PRECISION = old
        constants.resolution, constants.BATCH_SIZE = 256, 4


def test_bf16_hipgraph_replay_matches_eager_1024_batch8():
    """BASELINE config 5 exactly as bench.py --config c5 times it: SceneGraph, five scene attributes, clamp flow, the 16-bit path, forward +
    backward REPLAYED from one hipGraph (capture.CapturedStep) at 1024^2, batch 8 — against the same step launched eagerly from the same state.
    The forward has no atomics (images agree to the last bf16 bit or one rounding step); the reductions' atomics reorder fp32 sums, so the loss
    is held to 1e-4 relative and the walk gradient to a cosine of 0.9999 and 2e-2 of its largest entry."""
    import gc
    from latent2im_amd import capture, constants, selfcheck, synth
    old = conv.PRECISION
    try:
        conv.PRECISION = 'bf16'
        attrs = ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk']
        size, batch = 1024, 8
        zs = synth.z_sample(batch, seed=17)
        alpha = np.ones((batch, 5)) * np.random.RandomState(18).uniform(-1, 1, 5)
        ge = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        assert type(ge.module.netG).__module__.endswith('nets16')
        e = selfcheck.run_step(ge, zs, alpha, clamp=True, optimize=False)
        e = {k: (v.detach().float().clone() if torch.is_tensor(v) else v) for k, v in e.items()}
        del ge
        gc.collect()
        torch.cuda.empty_cache()
        gr = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        step = capture.CapturedStep(gr, batch, 5, clamp=True)
        r = step(zs, alpha, optimize=False)
        r2 = step(zs, alpha, optimize=False)                     # a second replay of the same inputs: the graph holds no state of its own
        torch.cuda.synchronize()
        for res in (r, r2):
            x1, g = res['x1'].detach().float(), res['grad'].detach().float()
            assert float((x1 - e['x1']).abs().max()) <= 2.0 ** -7 * float(e['x1'].abs().max())
            assert abs(float(res['loss']) - float(e['loss'])) <= 1e-4 * abs(float(e['loss']))
            assert float((res['eps'].float() - e['eps']).abs().max()) < 1e-4
            cos = float((g * e['grad']).sum() / (g.norm() * e['grad'].norm()))
            assert cos > 0.9999, cos
            assert float((g - e['grad']).abs().max()) < 2e-2 * float(e['grad'].abs().max())
        del step, gr
        gc.collect()
        torch.cuda.empty_cache()
    finally:
        conv.PRECISION = old
        constants.resolution, constants.BATCH_SIZE = 256, 4


@pytest.mark.parametrize('prec', ['bf16', 'f16'])
def test_bf16_data_parallel_matches_single_process(tmp_path, prec):
    """SURVEY 8(e) on the 16-bit path: two rank processes (one GPU here, process group over gloo; RCCL on a node) run the product graph with
    L2I_PRECISION=bf16 for two optimizeParametersAll steps on strided shards of a global batch of 8, full loss; a single process runs the
    global batch.  Sharding changes which samples share a launch, not a sample's arithmetic (bf16 rounding is per element, the discriminator's
    minibatch-stddev groups stay intact), so the total, regressor and GAN terms agree to 1e-3 (the content term, a difference of rounded maps, to 5e-2) and the all-reduced walk
    gradient to a cosine of 0.999."""
    import os
    import subprocess
    import sys
    from latent2im_amd import dist
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_worker.py')
    args = ['64', '8', '2', '0']
    env = dict(os.environ, L2I_DIST_BACKEND='gloo', L2I_PRECISION=prec)       # [r6] f16: per-RANK batch in the loss scales, GuardedAdam after the all-reduce
    env.pop('WORLD_SIZE', None)
    single, multi = str(tmp_path / 'single.npz'), str(tmp_path / 'dp2.npz')
    # (one after the other: three processes time-slicing the box's one GPU at once were measured 2x SLOWER than the two runs in sequence)
    r = subprocess.run([sys.executable, worker, single] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    codes, _ = dist.spawn_local(2, [sys.executable, worker, multi] + args, env=env, timeout=600)
    assert codes == [0, 0], codes
    a, b = np.load(single), np.load(multi)
    assert int(a['world']) == 1 and int(b['world']) == 2
    assert str(a['precision']) == prec and str(b['precision']) == prec
    la, lb = a['losses'][0], b['losses'][0]                            # total, regressor, content, GAN
    print('bf16 DP losses: single', la, 'two ranks', lb)
    np.testing.assert_allclose(lb[[0, 1, 3]], la[[0, 1, 3]], rtol=1e-3, atol=1e-5)
    # the content term is a mean squared DIFFERENCE of bf16-rounded feature maps: a different launch shape rounds differently
    np.testing.assert_allclose(lb[2], la[2], rtol=5e-2, atol=1e-5)
    ga, gb = a['grads'][0].reshape(-1), b['grads'][0].reshape(-1)
    cos = float((ga * gb).sum() / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    print('bf16 DP walk-gradient cosine', cos)
    assert cos > 0.999, cos


def test_train_multi_attr_cli_config5_flow_bf16_hipgraph(tmp_path):
    """BASELINE config 5 through the drop-in driver: train_multi_attr.py --transform scene with five transient-scene attributes,
    --precision bf16 --hip_graph (16-bit path, forward + backward replayed from the captured graph), 64^2, two iterations: log and checkpoints
    under the reference's names, a finite walk that moved."""
    import os
    from latent2im_amd import constants, trainer
    os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    models = str(tmp_path / 'models')
    argv = ['--model', 'stylegan_v2_real', '--transform', 'scene', '--num_samples', '8', '--learning_rate', '1e-3', '--latent', 'w',
            '--walk_type', 'linear', '--loss', 'l2', '--attrList', 'dirty,daylight,night,sunrisesunset,dawndusk',
            '--attrPath', './dataset/attributes_scene.txt', '--models_dir', models, '--overwrite_config', '--resolution', '64',
            '--batch_size', '4', '--n_epoch', '1', '--seed', '3', '--model_save_freq', '1', '--synthetic_weights', '--precision', 'bf16', '--hip_graph']
    old = conv.PRECISION
    try:
        trainer.main(multi_attr=True, argv=argv)
        out = os.path.join(models, 'stylegan_v2_real_scene_linear_lr0.001_l2_w')
        last = torch.load(os.path.join(out, 'model_w_1_final_walk_module.ckpt'), map_location='cpu', weights_only=False)
        assert tuple(last.w.shape)[0] == 5 and torch.isfinite(last.w).all()
        assert float(last.w.detach().std()) > 0.01
        log = open(os.path.join(out, 'log.txt')).read()
        assert 'T, epc, bst, lss, alpha:' in log
        import yaml
        opt = yaml.safe_load(open(os.path.join(out, 'opt.yml')))
        assert opt.get('precision') == 'bf16' and opt.get('hip_graph') is True
    finally:
        conv.PRECISION = old
        constants.resolution, constants.BATCH_SIZE = 256, 4


def test_modulate_planes_multi_matches_per_layer():
    """[r5] l2i_modulate_planes_multi_h8 (every modulated conv of a pass in one launch; networks.py:234-235 per layer) against the per-layer
    l2i_modulate_planes_h8: bit-identical planes."""
    from latent2im_amd import kernels16 as K16
    rs = np.random.RandomState(9)
    shapes = [(64, 32, 9), (32, 96, 9), (128, 64, 1), (32, 32, 9)]                 # cin, cout, taps
    w32s = [conv.pack_weight_h8_f32(T(rs.randn(co, ci, int(kk ** 0.5), int(kk ** 0.5)))).to(DEV) for ci, co, kk in shapes]
    B = 3
    rows = [ci for ci, _, _ in shapes]
    offs = [int(v) for v in np.concatenate([[0], np.cumsum(rows)[:-1]])]
    s_all = T(rs.rand(B * sum(rows)) + 0.5).to(DEV)
    plan = K16.ModulatePlan(w32s, offs, DEV)
    views = plan.run(s_all, B)
    torch.cuda.synchronize()
    for w32, off, ci, v in zip(w32s, offs, rows, views):
        want = K16.modulate_planes(w32, s_all[B * off:B * (off + ci)].view(B, ci).contiguous())
        assert v.shape == want.shape and torch.equal(v, want)


# ---- [r6] the optimiser tail of the fp16 path: guarded Adam + dynamic loss scale on the device (csrc/l2i_optim.hip, latent2im_amd/optim.py) ---------
def test_fp16_guarded_adam_matches_torch_adam_and_skips_nonfinite_steps():
    """l2i_adam_guarded_f32 against torch.optim.Adam(lr, betas=(0.5, 0.99)) (transform_base.py:329-331) over six steps of random gradients on a
    walk-shaped tensor: parameter and both moments to float32 rounding.  Then GradScaler semantics: a gradient with one inf (and one with a NaN) leaves
    parameter, moments and step counter bit-identical, halves the dynamic scale, resets the growth tracker, counts the skip; `interval` clean steps
    double the scale again, never past its cap."""
    from latent2im_amd import optim
    rs = np.random.RandomState(0)
    w0 = T(rs.randn(5, 18, 512) * 0.02)
    pa, pb = torch.nn.Parameter(w0.clone().to(DEV)), torch.nn.Parameter(w0.clone().to(DEV))
    ref = torch.optim.Adam([pa], lr=1e-3, betas=(0.5, 0.99))
    sc = optim.LossScaler(dict(R=0, V=0, D=0, G=0), DEV, growth_interval=3, max_log2=1)
    opt = optim.GuardedAdam([pb], lr=1e-3, betas=(0.5, 0.99), scaler=sc)
    for i in range(6):
        g = T(rs.randn(5, 18, 512) * 10.0 ** rs.uniform(-6, 0)).to(DEV)
        pa.grad, pb.grad = g.clone(), g.clone()
        ref.step()
        opt.step()
    torch.cuda.synchronize()
    assert float((pa - pb).abs().max()) < 2e-7, float((pa - pb).abs().max())
    for key in ('exp_avg', 'exp_avg_sq'):                                            # (m + (g - m) / 2 cancels: absolute bound at the moment's own scale)
        got, want = opt.state[pb][key].cpu().numpy(), ref.state[pa][key].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6 * float(np.abs(want).max()))
    st = sc.stats()
    assert st == dict(scale=2.0, tracker=0, skipped=0, steps=6), st                 # grew once after 3 clean steps; the cap 2^1 stopped the second doubling
    before = [pb.detach().clone(), opt.state[pb]['exp_avg'].clone(), opt.state[pb]['exp_avg_sq'].clone(), opt.state[pb]['step'].clone()]
    for bad in (float('inf'), float('nan')):
        g = T(rs.randn(5, 18, 512)).to(DEV)
        g[3, 7, 100] = bad
        pb.grad = g
        opt.step()
    torch.cuda.synchronize()
    after = [pb.detach(), opt.state[pb]['exp_avg'], opt.state[pb]['exp_avg_sq'], opt.state[pb]['step']]
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    st = sc.stats()
    assert st == dict(scale=0.5, tracker=0, skipped=2, steps=8), st
    assert float(sc.inv_dyn) == 2.0
    pb.grad = T(rs.randn(5, 18, 512)).to(DEV)
    opt.step()
    assert not torch.equal(pb.detach(), before[0]) and float(opt.state[pb]['step']) == 7.0
    # several parameter tensors (the MLP walks): one inf in ANY of them skips ALL of them
    ps = [torch.nn.Parameter(T(rs.randn(64, 32)).to(DEV)), torch.nn.Parameter(T(rs.randn(32)).to(DEV))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref2, opt2 = torch.optim.Adam(qs, lr=1e-2, betas=(0.5, 0.99)), optim.GuardedAdam(ps, lr=1e-2, betas=(0.5, 0.99))
    for i in range(3):
        gs = [T(rs.randn(*p.shape)).to(DEV) for p in ps]
        for p, q, g in zip(ps, qs, gs):
            p.grad, q.grad = g.clone(), g.clone()
        if i == 1:
            ps[1].grad[5] = float('inf')                        # the SECOND tensor: the first one's update must be skipped too
            keep = [p.detach().clone() for p in ps]
            opt2.step()
            assert all(torch.equal(p.detach(), k) for p, k in zip(ps, keep))
            continue
        ref2.step()
        opt2.step()
    for p, q in zip(ps, qs):
        assert float((p - q).abs().max()) < 1e-6


@pytest.mark.parametrize('hipgraph', [False, True], ids=['eager', 'hipgraph'])
def test_fp16_overflow_guard_skips_steps_and_backs_the_scale_off(monkeypatch, hipgraph):
    """The whole fp16 training step with the static exponents forced 13 octaves too high (L2I_F16_SCALES; the calibrated ones leave eleven octaves of
    headroom): the scaled gradient maps overflow fp16, the walk gradient comes back non-finite, and — instead of inf / NaN going straight into Adam as in
    round 5 — the step is skipped (walk and moments untouched), the dynamic factor halves step by step until the maps fit, and from then on the walk
    trains on gradients that agree with a run at the calibrated exponents.  No host read anywhere in the step (the same code path replays from the
    hipGraph: test_fp16_hipgraph_captured_steps_follow_eager_steps)."""
    from latent2im_amd import capture, constants, nets16, selfcheck, synth
    old = conv.PRECISION
    attrs = ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk']
    size, batch = 64, 4
    zs = synth.z_sample(batch, seed=21)
    alpha = np.ones((batch, 5)) * np.random.RandomState(22).uniform(-1, 1, 5)
    try:
        conv.PRECISION = 'f16'
        good = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        base = nets16.loss_scale_for(size, batch)
        assert good.loss_scaler.log2 == base
        rg = selfcheck.run_step(good, zs, alpha, clamp=True, optimize=False)
        assert bool(torch.isfinite(rg['grad']).all())
        monkeypatch.setenv('L2I_F16_SCALES', ','.join(str(base[k] + 13) for k in 'RVDG'))
        hot = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform='scene')
        monkeypatch.delenv('L2I_F16_SCALES')
        assert hot.loss_scaler.log2 == {k: v + 13 for k, v in base.items()}
        w0 = hot.walk.w.detach().clone()
        skipped, first_clean = 0, None
        # hipgraph: forward + backward replayed from ONE captured graph — the dynamic factor is a device tensor the captured multiplies read, so every
        # replay sees the value the guarded optimiser tail (outside the graph) left behind
        cs = capture.CapturedStep(hot, batch, 5, clamp=True) if hipgraph else None
        for i in range(16):
            r = cs(zs, alpha) if hipgraph else selfcheck.run_step(hot, zs, alpha, clamp=True)
            torch.cuda.synchronize()
            if hipgraph:
                r = dict(r, grad=r['grad'].detach().clone())
            finite = bool(torch.isfinite(r['grad']).all())
            assert bool(torch.isfinite(hot.walk.w).all())                           # whatever the gradient held, the walk never sees it
            if not finite:
                assert first_clean is None, 'a non-finite step after a clean one at a smaller scale'
                assert torch.equal(hot.walk.w.detach(), w0)
                skipped += 1
            elif first_clean is None:
                first_clean = r
                break
        st = hot.loss_scaler.stats()
        print('skipped %d steps; scaler %s' % (skipped, st))
        assert 1 <= skipped <= 14 and first_clean is not None
        assert st['skipped'] == skipped and st['scale'] == 2.0 ** -skipped and st['steps'] == skipped + 1
        assert not torch.equal(hot.walk.w.detach(), w0)                              # the first finite step trained
        g, gg = first_clean['grad'].float(), rg['grad'].float()
        cos = float((g * gg).sum() / (g.norm() * gg.norm()))
        assert cos > 0.999 and abs(float(g.norm() / gg.norm()) - 1) < 0.03, (cos, float(g.norm() / gg.norm()))
    finally:
        conv.PRECISION = old
        constants.resolution, constants.BATCH_SIZE = 256, 4


@pytest.mark.parametrize('name,up', [('same512', False), ('up512', True)])
def test_modulated_conv_layer_fixture_h8(name, up):
    """[r6] The reference's ModulatedConv2d layer fixtures (tests/golden/modconv.npz; networks.py:231-272; 48 -> 40 and 40 -> 24 channels: ragged channel
    groups on both sides) through the 16-bit path's call sequence — style folded into per-sample weight planes (l2i_modulate_planes_h8), demodulation as
    out_scale, the separable blur, the gradient conv on planes carrying the demodulation, the style gradient from dot_reduce — under the 16-bit
    contract (outputs and gradients to 2^-7 of their largest entry: three roundings on the way)."""
    import math
    from latent2im_amd import kernels16 as K16
    from latent2im_amd.generator import _Mod
    from tests.conftest import GOLDEN
    import os
    g = np.load(os.path.join(GOLDEN, 'modconv.npz'))
    P = {'m.' + k[len(name) + 3:]: g[k] for k in g.files if k.startswith(name + '.P.')}
    W = T(P['m.weight'])[0]
    cout, cin, k, _ = W.shape
    ws = W * (1.0 / math.sqrt(cin * k * k))
    mod = _Mod(P, 'm', DEV)
    x, w, gy = (T(g['%s.%s' % (name, n)]).to(DEV) for n in ('x', 'w', 'gy'))
    B = x.shape[0]
    s = mod(w)
    Tm = (ws * ws).sum((2, 3)).to(DEV)
    demod = torch.rsqrt((s * s) @ Tm.t() + 1e-8)
    pad32 = lambda v: torch.cat([v, v.new_zeros(B, (v.shape[1] + 31) // 32 * 32 - v.shape[1])], 1).contiguous()
    hc = conv.H8Conv(ws, stride=2 if up else 1, padding=0 if up else 1, transposed=up, device=DEV)
    wt = ws.transpose(0, 1).contiguous()
    w32_fwd = conv.pack_weight_h8_f32(ws).to(DEV)
    w32_bwd = conv.pack_weight_h8_f32(wt if up else torch.flip(wt, [2, 3])).to(DEV)
    planes = K16.modulate_planes(w32_fwd, pad32(s))
    xh = conv.to_h8(x, 32)
    t = hc.forward(xh, planes=planes, w_bstride=planes[0].numel() * 2, out_scale=demod.contiguous())
    bk = None
    if up:
        bk = T(P['m.blur.kernel']).to(DEV)
        yh = K16.upfirdn2d(t, bk, pad=(1, 1, 1, 1), sep=K16.separable(P['m.blur.kernel']))
    else:
        yh = t
    torch.cuda.synchronize()
    want = T(g[name + '.y']).double()
    y = conv.from_h8(yh, cout)
    assert float((y.double().cpu() - want).abs().max()) < 2.0 ** -7 * float(want.abs().max())
    gyh = conv.to_h8(gy, 32)
    dt = K16.upfirdn2d(gyh, torch.flip(bk, [0, 1]).contiguous(), pad=(2, 2, 2, 2), sep=K16.separable(np.asarray(P['m.blur.kernel'])[::-1, ::-1])) if up else gyh
    planes_b = K16.modulate_planes(w32_bwd, pad32(demod))
    dxmod = hc.dgrad(dt, (x.shape[2], x.shape[3]), planes=planes_b, w_bstride=planes_b[0].numel() * 2)
    gx = conv.from_h8(dxmod, cin) * s[:, :, None, None]
    want = T(g[name + '.gx']).double()
    assert float((gx.double().cpu() - want).abs().max()) < 2.0 ** -7 * float(want.abs().max())
    q = K16.dot_reduce(dxmod, conv.to_h8(x, 8))[:, :cin]               # (maps of equal channel-group count)
    red = K16.dot_reduce(conv.to_h8(gy, 8), yh)[:, :cout]
    ds = q - s * ((red * demod * demod) @ Tm)
    gw = (ds @ mod.A).double().cpu()
    want = T(g[name + '.gw']).double()
    assert float((gw - want).abs().max()) < 2.0 ** -6 * float(want.abs().max()), float((gw - want).abs().max()) / float(want.abs().max())


def test_conv_h8_sign_planes_written_and_read():
    """[r6] l2i_conv_params::mask_out / mask_bits: a launch writes the sign plane of its h8 output (one byte per 16-byte pixel slot, bit e = the stored element
    e > 0) from both epilogues (lean: bias + ReLU / leaky ReLU; general: residual + ReLU), bit-exactly what the stored map says; and a gradient launch
    that reads its out_mask / res_mask as such planes returns the same bits as the launch that reads the maps."""
    rs = np.random.RandomState(5)
    g = lambda t: t.to(DEV)
    b, cin, cout, h, w = 2, 64, 96, 19, 45
    x = conv.to_h8(g(T(rs.randn(b, cin, h, w))), 32)
    bias = g(T(rs.randn(cout)))

    def want_bits(y):                                   # [B, G, H, W, 8] 16-bit -> [B, G, H, W] uint8
        pos = (y.float() > 0).to(torch.int32)
        return sum(pos[..., e] << e for e in range(8)).to(torch.uint8)

    for k, kw in ((3, dict(bias=bias, act=conv.ACT_RELU)), (3, dict(bias=bias, act=conv.ACT_LRELU, slope=0.2, gain=2 ** 0.5)), (1, dict(bias=bias, act=conv.ACT_RELU))):
        hc = conv.H8Conv(T(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)), 1, k // 2, device=DEV)
        y0 = hc.forward(x, **kw)
        planes = torch.full((b, cout // 8, h, w), 0xAA, device=DEV, dtype=torch.uint8)
        y1 = hc.forward(x, mask_out=planes, **kw)                                        # lean epilogue
        res = conv.to_h8(g(T(rs.randn(b, cout, h, w))), 8)
        planes_r = torch.full_like(planes, 0x55)
        y2 = hc.forward(x, residual=res, mask_out=planes_r, **kw)                        # general epilogue
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        assert torch.equal(planes, want_bits(y1)) and torch.equal(planes_r, want_bits(y2))
        assert 0.2 < float((planes != 0).float().mean()) and int(planes_r.max()) > 127 and int((planes_r & 1).sum()) > 0
    # consumers: out_mask / res_mask as maps vs as planes
    hc = conv.H8Conv(T(rs.randn(cout, cin, 1, 1) / np.sqrt(cin)), 1, 0, device=DEV)
    gy = conv.to_h8(g(T(rs.randn(b, cout, h, w))), 32)
    act = conv.to_h8(g(T(rs.randn(b, cin, h, w))), 8)                                    # the activation whose sign masks the gradient
    res = conv.to_h8(g(T(rs.randn(b, cin, h, w))), 8)
    bits = want_bits(act)
    for kw in (dict(out_mask=act), dict(out_mask=act, res_mask=act, residual=res), dict(out_mask=act, mask=(2 ** 0.5, 0.2 * 2 ** 0.5))):
        a = hc.dgrad(gy, (h, w), **kw)
        kb = {k_: (bits if k_ in ('out_mask', 'res_mask') else v) for k_, v in kw.items()}
        bq = hc.dgrad(gy, (h, w), mask_bits=True, **kb)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16), bq.view(torch.int16)), list(kw)
    with pytest.raises((AssertionError, _lib.L2IError)):     # planes are an h8-epilogue feature: refused with the fp32 output
        hc.dgrad(gy, (h, w), out_f32=True, out_mask=bits, mask_bits=True)


def test_resnet16_backward_on_sign_planes_is_bit_identical():
    """[r6] nets16.ResNet50 with the backward's ReLU masks as one-bit sign planes (written by the forward convs) against the round-5 form that reads the
    activation maps: same masks, same launches otherwise — the image gradient must be bit-identical, at a shape with and one without stride-2 tails."""
    from latent2im_amd import nets16, synth
    net = nets16.ResNet50(synth.resnet50_state(seed=300), device=DEV)
    rs = np.random.RandomState(11)
    for size, batch in ((64, 4), (128, 2)):
        x = T(rs.randn(batch, 3, size, size) * 0.5).to(DEV)
        gy = T(rs.randn(batch, 40)).to(DEV)
        grads = []
        old = nets16.SIGN_PLANES
        try:
            for flag in (True, False):
                nets16.SIGN_PLANES = flag
                xg = x.clone().requires_grad_(True)
                net(xg).backward(gy)
                grads.append(xg.grad.detach().clone())
        finally:
            nets16.SIGN_PLANES = old
        torch.cuda.synchronize()
        assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


def test_discriminator16_backward_on_sign_planes_is_bit_identical():
    """[r6] nets16.Discriminator with the leaky-ReLU masks of its backward as sign planes (conv1 / conv2 of every block write them; the masked blur and
    mask_mul read them) against the round-5 form on the maps: bit-identical image gradient; and the two stream kernels against their map forms."""
    from latent2im_amd import kernels16 as K16, nets16, synth
    rs = np.random.RandomState(3)
    g = lambda t: t.to(DEV)
    x = conv.to_h8(g(T(rs.randn(2, 64, 21, 37))), 8)
    ref = conv.to_h8(g(T(rs.randn(2, 64, 21, 37))), 8)
    pos = (ref.float() > 0).to(torch.int32)
    bits = sum(pos[..., e] << e for e in range(8)).to(torch.uint8).contiguous()
    assert torch.equal(K16.mask_mul(x, ref, 1.5, 0.3).view(torch.int16), K16.mask_mul(x, bits, 1.5, 0.3).view(torch.int16))
    k = T(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0).to(DEV)
    sep = K16.separable(k)
    a = K16.upfirdn2d(x, k, pad=(1, 2, 2, 1), mask=ref, mask_vals=(2 ** 0.5, 0.2 * 2 ** 0.5), sep=sep)
    b = K16.upfirdn2d(x, k, pad=(1, 2, 2, 1), mask=bits, mask_vals=(2 ** 0.5, 0.2 * 2 ** 0.5), sep=sep, mask_bits=True)
    assert a.shape == ref.shape and torch.equal(a.view(torch.int16), b.view(torch.int16))
    for size, batch in ((64, 4), (128, 4)):
        net = nets16.Discriminator(synth.discriminator_state(size, seed=200), size, device=DEV)
        img = T(rs.randn(batch, 3, size, size) * 0.5).to(DEV)
        grads = []
        old = nets16.SIGN_PLANES
        try:
            for flag in (True, False):
                nets16.SIGN_PLANES = flag
                xg = img.clone().requires_grad_(True)
                net(xg).sum().backward()
                grads.append(xg.grad.detach().clone())
        finally:
            nets16.SIGN_PLANES = old
        torch.cuda.synchronize()
        assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


@pytest.mark.parametrize('c1,c2,c3,h,w,batch', [(64, 256, 64, 16, 32, 2), (128, 512, 128, 16, 16, 3), (64, 256, 128, 8, 32, 2), (64, 256, 64, 32, 40, 1), (256, 1024, 256, 16, 16, 2),
                                                  (128, 512, 256, 8, 16, 3)])
def test_conv1x1_pair_h8_is_bit_identical_to_two_launches(c1, c2, c3, h, w, batch):
    """[r6] l2i_conv1x1_pair_h8 (csrc/l2i_pair_h8.hip): two chained 1x1 convs in one launch, the wide map handed from the first conv's epilogue to the second
    conv's MFMAs in registers.  Forward form (bias + identity + ReLU + sign plane -> bias + ReLU + sign plane) and backward form (sign-plane mask on the
    accumulator and on the trunk gradient -> sign-plane mask), both tile variants: every output and every sign plane equals the two l2i_conv2d_h8 launches'
    bit for bit (the second conv sees the rounded 16-bit map in both, K steps accumulate in the same order)."""
    rs = np.random.RandomState(c1 + h)
    g = lambda t: t.to(DEV)
    A = conv.H8Conv(T(rs.randn(c2, c1, 1, 1) / np.sqrt(c1)), 1, 0, device=DEV)
    Bc = conv.H8Conv(T(rs.randn(c3, c2, 1, 1) / np.sqrt(c2)), 1, 0, device=DEV)
    x = conv.to_h8(g(T(rs.randn(batch, c1, h, w))), 32)
    res = conv.to_h8(g(T(rs.randn(batch, c2, h, w))), 8)
    ba, bb = g(T(rs.randn(c2))), g(T(rs.randn(c3) * 0.3))
    plane = lambda c, fill: torch.full((batch, c // 8, h, w), fill, device=DEV, dtype=torch.uint8)
    assert conv.pair_h8_shapes_ok(c1, c2, c3, h * w)
    # ---- forward form ----
    m_mid, m_out = plane(c2, 0xAA), plane(c3, 0x55)
    mid0 = A.forward(x, bias=ba, residual=res, act=conv.ACT_RELU, mask_out=m_mid)
    out0 = Bc.forward(mid0, bias=bb, act=conv.ACT_RELU, mask_out=m_out)
    for variant in (0, 1):
        if variant == 0 and (h * w) % 256:
            continue
        pm_mid, pm_out = plane(c2, 0x11), plane(c3, 0x22)
        d = []
        mid1 = A.forward(x, bias=ba, residual=res, act=conv.ACT_RELU, mask_out=pm_mid, _defer=d)
        out1 = Bc.forward(mid1, bias=bb, act=conv.ACT_RELU, mask_out=pm_out, _defer=d)
        conv.launch_pair_h8(d, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(mid0.view(torch.int16), mid1.view(torch.int16)) and torch.equal(out0.view(torch.int16), out1.view(torch.int16)), variant
        assert torch.equal(m_mid, pm_mid) and torch.equal(m_out, pm_out), variant
        assert float(out1.float().abs().max()) > 0 and 0.1 < float((out1.float() > 0).float().mean()) < 0.9
    # ---- backward form: G_prev = (conv of g + G) * [m > 0]; then (conv of G_prev) * [y2 > 0], no biases, no activation ----
    bits1 = torch.from_numpy(rs.randint(0, 256, size=(batch, c2 // 8, h, w)).astype(np.uint8)).to(DEV)
    bits2 = torch.from_numpy(rs.randint(0, 256, size=(batch, c3 // 8, h, w)).astype(np.uint8)).to(DEV)
    mid0 = A.forward(x, residual=res, out_mask=bits1, res_mask=bits1, mask_bits=True)
    out0 = Bc.forward(mid0, out_mask=bits2, mask_bits=True)
    for variant in (0, 1):
        if variant == 0 and (h * w) % 256:
            continue
        d = []
        mid1 = A.forward(x, residual=res, out_mask=bits1, res_mask=bits1, mask_bits=True, _defer=d)
        out1 = Bc.forward(mid1, out_mask=bits2, mask_bits=True, _defer=d)
        conv.launch_pair_h8(d, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(mid0.view(torch.int16), mid1.view(torch.int16)) and torch.equal(out0.view(torch.int16), out1.view(torch.int16)), variant
    # refusals: a 3x3 first conv, a residual-less first conv
    with pytest.raises((AssertionError, _lib.L2IError)):
        d = []
        A.forward(x, bias=ba, act=conv.ACT_RELU, _defer=d)
        Bc.forward(mid1, bias=bb, act=conv.ACT_RELU, _defer=d)
        conv.launch_pair_h8(d)


def test_resnet16_pair_launches_are_bit_identical():
    """[r6] nets16.ResNet50 with the trunk's chained convs as l2i_conv_chain3_h8 / l2i_conv1x1_pair_h8 launches (nets16.CHAIN3 / PAIR) against the separate launches:
    predictions of the no-grad pass, predictions and image gradient of the differentiable pass — bit-identical; and the fused launches are really taken."""
    from latent2im_amd import nets16, synth
    net = nets16.ResNet50(synth.resnet50_state(seed=301), device=DEV)
    rs = np.random.RandomState(12)
    x = T(rs.randn(2, 3, 256, 256) * 0.5).to(DEV)
    gy = T(rs.randn(2, 40)).to(DEV)
    res, taken = [], []
    old = nets16.PAIR, nets16.CHAIN3
    try:
        for pair, chain in ((True, True), (True, False), (False, False)):
            nets16.PAIR, nets16.CHAIN3 = pair, chain
            conv.PROFILE = []
            with torch.no_grad():
                p0 = net(x).clone()
            xg = x.clone().requires_grad_(True)
            p1 = net(xg)
            p1.backward(gy)
            torch.cuda.synchronize()
            taken.append((sum(1 for q in conv.PROFILE if q[4].startswith('l2i_conv_chain3_h8')), sum(1 for q in conv.PROFILE if q[4].startswith('l2i_conv1x1_pair_h8'))))
            conv.PROFILE = None
            res.append((p0, p1.detach().clone(), xg.grad.detach().clone()))
    finally:
        (nets16.PAIR, nets16.CHAIN3), conv.PROFILE = old, None
    # 64^2 / 32^2 / 16^2 trunks at this input: layer1 + layer2 chains of two forwards and one backward, the wider shapes as pairs
    assert taken[0][0] >= 12 and taken[0][1] >= 3 and taken[1][0] == 0 and taken[1][1] >= 15 and taken[2] == (0, 0), taken
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a, b) and float(a.abs().max()) > 0


@pytest.mark.parametrize('c,c2,c3,h,w,batch', [(64, 256, 64, 16, 32, 2), (128, 512, 128, 8, 32, 2), (64, 256, 128, 24, 64, 1), (64, 256, 64, 12, 32, 3)])
def test_conv_chain3_h8_is_bit_identical_to_three_launches(c, c2, c3, h, w, batch):
    """[r6] l2i_conv_chain3_h8: the 3x3 stride-1 conv in front of the pair in the same launch (one launch per ResNet-50 bottleneck: conv2 -> conv3 + identity -> next
    conv1; backwards: conv2's input gradient -> conv1's + trunk gradient -> conv3's of the block below).  The 3x3 conv's map is never written; its sign plane, the wide
    map and the last conv's output (and their planes) equal the three l2i_conv2d_h8 launches' bit for bit, on tiles that cross the image border (zero padding)
    and on both tile variants."""
    rs = np.random.RandomState(c + h)
    g = lambda t: t.to(DEV)
    H3 = conv.H8Conv(T(rs.randn(c, c, 3, 3) / np.sqrt(9 * c)), 1, 1, device=DEV, cin_pad=16)
    A = conv.H8Conv(T(rs.randn(c2, c, 1, 1) / np.sqrt(c)), 1, 0, device=DEV)
    Bc = conv.H8Conv(T(rs.randn(c3, c2, 1, 1) / np.sqrt(c2)), 1, 0, device=DEV)
    x = conv.to_h8(g(T(rs.randn(batch, c, h, w))), 16)
    res = conv.to_h8(g(T(rs.randn(batch, c2, h, w))), 8)
    b0, ba, bb = g(T(rs.randn(c) * 0.3)), g(T(rs.randn(c2))), g(T(rs.randn(c3) * 0.3))
    plane = lambda ch, fill: torch.full((batch, ch // 8, h, w), fill, device=DEV, dtype=torch.uint8)
    assert conv.chain3_h8_shapes_ok(c, c2, c3, h, w)
    # ---- forward form ----
    m0, m_mid, m_out = plane(c, 0xAA), plane(c2, 0xAA), plane(c3, 0x55)
    y0 = H3.forward(x, bias=b0, act=conv.ACT_RELU, mask_out=m0)
    mid0 = A.forward(y0, bias=ba, residual=res, act=conv.ACT_RELU, mask_out=m_mid)
    out0 = Bc.forward(mid0, bias=bb, act=conv.ACT_RELU, mask_out=m_out)
    for variant in (0, 1):
        if variant == 0 and h % 8:
            continue
        q0, qm, qo = plane(c, 0x11), plane(c2, 0x22), plane(c3, 0x33)
        d = []
        y1 = H3.forward(x, bias=b0, act=conv.ACT_RELU, mask_out=q0, _defer=d)
        mid1 = A.forward(y1, bias=ba, residual=res, act=conv.ACT_RELU, mask_out=qm, _defer=d)
        out1 = Bc.forward(mid1, bias=bb, act=conv.ACT_RELU, mask_out=qo, _defer=d)
        d[0][0].y = None                                   # the 3x3 conv's map is not wanted
        conv.launch_pair_h8(d, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(mid0.view(torch.int16), mid1.view(torch.int16)) and torch.equal(out0.view(torch.int16), out1.view(torch.int16)), variant
        assert torch.equal(m0, q0) and torch.equal(m_mid, qm) and torch.equal(m_out, qo), variant
    # ---- backward form ----
    bits0 = torch.from_numpy(rs.randint(0, 256, size=(batch, c // 8, h, w)).astype(np.uint8)).to(DEV)
    bits1 = torch.from_numpy(rs.randint(0, 256, size=(batch, c2 // 8, h, w)).astype(np.uint8)).to(DEV)
    bits2 = torch.from_numpy(rs.randint(0, 256, size=(batch, c3 // 8, h, w)).astype(np.uint8)).to(DEV)
    y0 = H3.forward(x, out_mask=bits0, mask_bits=True)
    mid0 = A.forward(y0, residual=res, out_mask=bits1, res_mask=bits1, mask_bits=True)
    out0 = Bc.forward(mid0, out_mask=bits2, mask_bits=True)
    for variant in (0, 1):
        if variant == 0 and h % 8:
            continue
        d = []
        y1 = H3.forward(x, out_mask=bits0, mask_bits=True, _defer=d)
        mid1 = A.forward(y1, residual=res, out_mask=bits1, res_mask=bits1, mask_bits=True, _defer=d)
        out1 = Bc.forward(mid1, out_mask=bits2, mask_bits=True, _defer=d)
        d[0][0].y = None
        conv.launch_pair_h8(d, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(mid0.view(torch.int16), mid1.view(torch.int16)) and torch.equal(out0.view(torch.int16), out1.view(torch.int16)), variant
