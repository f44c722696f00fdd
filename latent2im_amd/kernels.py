"""Python call wrappers for the streaming kernels of libl2i_hip.so (include/l2i.h).  Tensors in, tensors out;
all GPU float32 contiguous; no autograd here (see op/ and the network modules for the differentiable forms)."""
import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU  # noqa: F401

SQRT2 = 2 ** 0.5


def fused_bias_act(x, b, ref, act, grad, alpha, scale, out=None):
    """Reference signature fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
    (op/fused_bias_act.cpp:11-21); empty tensors / None mean "absent"."""
    lib = _lib.load()
    x = x.contiguous()
    b = None if (b is None or b.numel() == 0) else b.contiguous()
    ref = None if (ref is None or ref.numel() == 0) else ref.contiguous()
    y = torch.empty_like(x) if out is None else out
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.shape[i]
    size_b = b.numel() if b is not None else 0
    if ref is not None:
        assert ref.shape == x.shape
    _lib.check(lib.l2i_fused_bias_act_f32(_lib.fptr(y), _lib.fptr(x), _lib.fptr(b), _lib.fptr(ref), x.numel(),
                                          step_b, size_b, int(act), int(grad), float(alpha), float(scale),
                                          _lib.stream_ptr()), 'l2i_fused_bias_act_f32')
    return y


def upfirdn2d_out_hw(h, w, kh, kw, up, down, pad):
    px0, px1, py0, py1 = pad
    return (h * up[1] + py0 + py1 - kh) // down[1] + 1, (w * up[0] + px0 + px1 - kw) // down[0] + 1


def upfirdn2d(x, kernel, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0), noise=None, noise_w=0.0, bias=None, addend=None,
              act=ACT_NONE, slope=0.2, gain=1.0, out=None, mask=None, mask_vals=(1.0, 0.0)):
    """x [N, C, H, W]; up/down = (x, y); pad = (x0, x1, y0, y1) as in op/upfirdn2d.cpp:12-23.  Optional fused
    epilogue: act(fir(x) + noise*noise_w + bias[c] + addend) * gain, then [r5] * (mask > 0 ? mask_vals[0] : mask_vals[1]) (mask shaped like the output)."""
    lib = _lib.load()
    x = x.contiguous()
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    oh, ow = upfirdn2d_out_hw(h, w, kh, kw, up, down, pad)
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32) if out is None else out
    assert y.shape == (n, c, oh, ow), (y.shape, (n, c, oh, ow))
    if addend is not None:
        assert addend.shape == y.shape
    if mask is not None:
        assert mask.shape == y.shape
        _lib.check(lib.l2i_upfirdn2d_masked_f32(_lib.fptr(y), _lib.fptr(x), _lib.fptr(kernel.contiguous()), n * c, h, w, kh, kw,
                                                up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3], c,
                                                _lib.fptr(noise), float(noise_w), _lib.fptr(bias), _lib.fptr(addend),
                                                int(act), float(slope), float(gain), _lib.fptr(mask.contiguous()), float(mask_vals[0]), float(mask_vals[1]),
                                                _lib.stream_ptr()), 'l2i_upfirdn2d_masked_f32')
        return y
    _lib.check(lib.l2i_upfirdn2d_f32(_lib.fptr(y), _lib.fptr(x), _lib.fptr(kernel.contiguous()), n * c, h, w, kh, kw,
                                     up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3], c,
                                     _lib.fptr(noise), float(noise_w), _lib.fptr(bias), _lib.fptr(addend),
                                     int(act), float(slope), float(gain), _lib.stream_ptr()), 'l2i_upfirdn2d_f32')
    return y


def torgb_fwd(x, wmod, bias):
    """x [B,C,H,W], wmod [B,3,C], bias [3] -> rgb [B,3,H,W]."""
    lib = _lib.load()
    b, c, h, w = x.shape
    rgb = torch.empty(b, 3, h, w, device=x.device, dtype=torch.float32)
    _lib.check(lib.l2i_torgb_fwd_f32(_lib.fptr(rgb), _lib.fptr(x), _lib.fptr(wmod.contiguous()), _lib.fptr(bias), b, c,
                                     h * w, _lib.stream_ptr()), 'l2i_torgb_fwd_f32')
    return rgb


def sg2_act_bwd(y, gin=None, gin_scale=None, grgb=None, wmod_rgb=None, bias=None, noise=None, noise_w=0.0,
                slope=0.2, gain=SQRT2, want_rgb_red=True, red=None, red_rgb=None, red_q=None):
    """Fused StyledConv elementwise backward (see l2i.h).  Returns (dz, red_dz_z [B,C], red_x_grgb [B,C,3] or None).  ``red`` / ``red_rgb`` /
    ``red_q`` ([r5] [B*C]: sum_p gin * y, the NEXT layer's style gradient): ZEROED destination buffers of the reductions (views of one buffer zeroed once per backward pass); allocated here when absent."""
    lib = _lib.load()
    b, c, h, w = y.shape
    dz = torch.empty_like(y)
    if red is None:
        red = torch.zeros(b, c, device=y.device, dtype=torch.float32)
    if red_rgb is None and grgb is not None and want_rgb_red:
        red_rgb = torch.zeros(b, c, 3, device=y.device, dtype=torch.float32)
    if grgb is None or not want_rgb_red:
        red_rgb = None
    _lib.check(lib.l2i_sg2_act_bwd_f32(_lib.fptr(dz), _lib.fptr(gin), _lib.fptr(gin_scale), _lib.fptr(grgb),
                                       _lib.fptr(wmod_rgb), _lib.fptr(y), _lib.fptr(bias), _lib.fptr(noise),
                                       float(noise_w), float(slope), float(gain), _lib.fptr(red), _lib.fptr(red_rgb), _lib.fptr(red_q),
                                       b, c, h * w, _lib.stream_ptr()), 'l2i_sg2_act_bwd_f32')
    return dz, red, red_rgb


def dot_reduce(a, b=None, out=None):
    """a, b [..., P] viewed as [rows, cols] with cols = prod(last 2 dims) for 4-D maps: returns sum over pixels.  ``out``: a ZEROED
    contiguous destination of ``rows`` floats (a view of a buffer zeroed once per backward pass); allocated here when absent."""
    lib = _lib.load()
    if a.dim() == 4:
        rows, cols = a.shape[0] * a.shape[1], a.shape[2] * a.shape[3]
        shape = a.shape[:2]
    else:
        rows, cols = a.shape[0], a.numel() // a.shape[0]
        shape = (rows,)
    if out is None:
        out = torch.zeros(rows, device=a.device, dtype=torch.float32)
    assert out.numel() == rows and out.is_contiguous()
    _lib.check(lib.l2i_dot_reduce_f32(_lib.fptr(out), _lib.fptr(a), _lib.fptr(b), rows, cols, _lib.stream_ptr()),
               'l2i_dot_reduce_f32')
    return out.reshape(shape)


def maxpool2d_fwd(x, k, s, pad):
    lib = _lib.load()
    n, c, h, w = x.shape
    oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
    idx = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.uint8)
    _lib.check(lib.l2i_maxpool2d_fwd_f32(_lib.fptr(y), _lib.ptr(idx), _lib.fptr(x), n * c, h, w, k, s, pad, oh, ow,
                                         _lib.stream_ptr()), 'l2i_maxpool2d_fwd_f32')
    return y, idx


def maxpool2d_bwd(gy, idx, in_hw, k, s, pad):
    lib = _lib.load()
    n, c, oh, ow = gy.shape
    gx = torch.empty(n, c, in_hw[0], in_hw[1], device=gy.device, dtype=torch.float32)
    _lib.check(lib.l2i_maxpool2d_bwd_f32(_lib.fptr(gx), _lib.fptr(gy), _lib.ptr(idx), n * c, in_hw[0], in_hw[1], k, s,
                                         pad, oh, ow, _lib.stream_ptr()), 'l2i_maxpool2d_bwd_f32')
    return gx


def maxpool2x2_bwd_add_diff(gy, idx, a, b, coef, coef_dev=None):
    """maxpool(2,2) backward of ``gy`` plus coef*coef_dev*(b - a) on the pool's input, in one pass (VGG conv1_2 tap)."""
    lib = _lib.load()
    n, c, oh, ow = gy.shape
    assert a.shape == b.shape == (n, c, 2 * oh, 2 * ow)
    gx = torch.empty_like(b)
    _lib.check(lib.l2i_maxpool2x2_bwd_add_diff_f32(_lib.fptr(gx), _lib.fptr(gy), _lib.ptr(idx), _lib.fptr(a), _lib.fptr(b), float(coef),
                                                   _lib.fptr(coef_dev), n * c, oh, ow, _lib.stream_ptr()), 'l2i_maxpool2x2_bwd_add_diff_f32')
    return gx


def sqdiff(a, b, coef=0.0, want_grad=False, coef_dev=None, want_sum=True):
    """sum((b-a)^2) as a 1-element tensor, and optionally coef*coef_dev*(b-a) (coef_dev: 1-element device tensor)."""
    lib = _lib.load()
    s = torch.zeros(1, device=a.device, dtype=torch.float32) if want_sum else None
    g = torch.empty_like(b) if want_grad else None
    _lib.check(lib.l2i_sqdiff_f32(_lib.fptr(s), _lib.fptr(g), _lib.fptr(a), _lib.fptr(b), a.numel(), float(coef),
                                  _lib.fptr(coef_dev), _lib.stream_ptr()), 'l2i_sqdiff_f32')
    return s, g


def axpby(a, b=None, alpha=1.0, beta=1.0, out=None):
    lib = _lib.load()
    y = torch.empty_like(a) if out is None else out
    _lib.check(lib.l2i_axpby_f32(_lib.fptr(y), _lib.fptr(a), _lib.fptr(b), float(alpha), float(beta), a.numel(),
                                 _lib.stream_ptr()), 'l2i_axpby_f32')
    return y


def relu_mask(g, ref):
    lib = _lib.load()
    y = torch.empty_like(g)
    _lib.check(lib.l2i_relu_mask_f32(_lib.fptr(y), _lib.fptr(g), _lib.fptr(ref), g.numel(), _lib.stream_ptr()),
               'l2i_relu_mask_f32')
    return y


def pixelnorm_act(x, slope=0.2, eps=1e-8):
    """lrelu(x / sqrt(mean_c x^2 + eps), slope) on [B,C,...] (model_256.py:78-84 + the LeakyReLU(0.2) that follows; slope 1: PixelNorm alone)."""
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    hw = x.numel() // (x.shape[0] * x.shape[1])
    _lib.check(lib.l2i_pixelnorm_act_f32(_lib.fptr(y), _lib.fptr(x), x.shape[0], x.shape[1], hw, float(eps), float(slope), _lib.stream_ptr()),
               'l2i_pixelnorm_act_f32')
    return y


def pixelnorm_act_bwd(gy, x, slope=0.2, eps=1e-8):
    lib = _lib.load()
    gy, x = gy.contiguous(), x.contiguous()
    assert gy.shape == x.shape
    dx = torch.empty_like(x)
    hw = x.numel() // (x.shape[0] * x.shape[1])
    _lib.check(lib.l2i_pixelnorm_act_bwd_f32(_lib.fptr(dx), _lib.fptr(gy), _lib.fptr(x), x.shape[0], x.shape[1], hw, float(eps), float(slope),
                                             _lib.stream_ptr()), 'l2i_pixelnorm_act_bwd_f32')
    return dx


def upsample2x_nearest(x, scale=1.0):
    lib = _lib.load()
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty(n, c, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    _lib.check(lib.l2i_upsample2x_nearest_f32(_lib.fptr(y), _lib.fptr(x), n * c, h, w, float(scale), _lib.stream_ptr()), 'l2i_upsample2x_nearest_f32')
    return y


def pool2x2(x, scale=0.25):
    """scale * (sum of every 2x2 window): 0.25 = bilinear halving (align_corners=False), 1 = adjoint of the nearest 2x upsample."""
    lib = _lib.load()
    x = x.contiguous()
    n, c, h, w = x.shape
    assert h % 2 == 0 and w % 2 == 0
    y = torch.empty(n, c, h // 2, w // 2, device=x.device, dtype=torch.float32)
    _lib.check(lib.l2i_pool2x2_f32(_lib.fptr(y), _lib.fptr(x), n * c, h // 2, w // 2, float(scale), _lib.stream_ptr()), 'l2i_pool2x2_f32')
    return y


def segmented_matvec(out, inp, w, segs, block_seg, nblocks, B, in2=None, bias=None, e1=None, e2=None, wmod=None, wrgb=None):
    """l2i_segmented_matvec_f32 (include/l2i.h): every layer's style-dependent vectors of one kind in one launch.  ``segs`` / ``block_seg``
    are device tensors holding the segment tables (latent2im_amd/generator.py:_ModPlan)."""
    lib = _lib.load()
    _lib.check(lib.l2i_segmented_matvec_f32(_lib.fptr(out), _lib.fptr(inp), _lib.fptr(in2), _lib.fptr(w), _lib.fptr(bias), _lib.fptr(e1), _lib.fptr(e2),
                                            _lib.fptr(wmod), _lib.fptr(wrgb), _lib.ptr(segs), _lib.ptr(block_seg), int(nblocks), int(B),
                                            _lib.stream_ptr()), 'l2i_segmented_matvec_f32')
    return out

