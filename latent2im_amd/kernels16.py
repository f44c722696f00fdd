"""Python call wrappers for the streaming kernels of the 16-bit path (csrc/l2i_stream_h8.hip, include/l2i.h): bf16 tensors in the
channel-blocked h8 layout [B, C/8, H, W, 8] in, out; fp32 for images, noise, bias, per-sample vectors and reductions.  No autograd here."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU  # noqa: F401

BF = torch.bfloat16
H8_DTYPES = (torch.bfloat16, torch.float16)


def h8_dtype():
    """Element type of the 16-bit path's h8 maps: bf16 (conv.PRECISION 'bf16') or IEEE fp16 ('f16', [r5])."""
    from . import conv
    return conv.h8_dtype()


def _fn(lib, name, dtype):
    """The entry point of `name` for h8 maps of `dtype` (csrc/l2i_*_h8.hip are compiled once per element type)."""
    return getattr(lib, name + '_f16' if dtype == torch.float16 else name)


def separable(kernel):
    """(k1y, k1x) with kernel == outer(k1y, k1x) (to float32 rounding), or None: computed ONCE from a host copy of a constant FIR kernel (the
    networks do it at construction: [1,3,3,1] x [1,3,3,1] / 64 times the up-sampling gain) and handed to ``upfirdn2d(..., sep=...)``, which then
    takes the register-streaming separable kernel; nothing is read back from the device per call."""
    import numpy as np
    k = np.asarray(kernel.detach().float().cpu().numpy() if torch.is_tensor(kernel) else kernel, dtype=np.float64)
    if k.shape != (4, 4):
        return None
    i, j = divmod(int(np.abs(k).argmax()), 4)
    if k[i, j] == 0:
        return None
    ky, kx = k[:, j] / k[i, j], k[i, :].copy()
    if not np.allclose(np.outer(ky, kx), k, rtol=1e-6, atol=0):
        return None
    return [float(v) for v in ky], [float(v) for v in kx]


def _h8(t):
    assert t.dtype in H8_DTYPES and t.dim() == 5 and t.shape[4] == 8 and t.is_contiguous(), (t.dtype, t.shape)
    return _lib.ptr(t)


def cast_to_h8(x, cpad=None, dtype=None):
    """fp32 NCHW -> 16-bit h8 with ``cpad`` channels (zero filled above C)."""
    lib = _lib.load()
    B, C, H, W = x.shape
    cpad = (C + 7) // 8 * 8 if cpad is None else cpad
    dtype = dtype or h8_dtype()
    y = torch.empty(B, cpad // 8, H, W, 8, device=x.device, dtype=dtype)
    _lib.check(_fn(lib, 'l2i_cast_f32_to_h8', dtype)(_lib.ptr(y), _lib.fptr(x.contiguous()), B, C, cpad, H * W, _lib.stream_ptr()), 'l2i_cast_f32_to_h8')
    return y


def cast_from_h8(t, channels=None):
    lib = _lib.load()
    B, G8, H, W, _ = t.shape
    C = G8 * 8 if channels is None else channels
    y = torch.empty(B, C, H, W, device=t.device, dtype=torch.float32)
    _lib.check(_fn(lib, 'l2i_cast_h8_to_f32', t.dtype)(_lib.fptr(y), _h8(t), B, C, G8 * 8, H * W, _lib.stream_ptr()), 'l2i_cast_h8_to_f32')
    return y


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0, 0, 0), noise=None, noise_w=0.0, bias=None, act=ACT_NONE, slope=0.2, gain=1.0, mask=None, mask_vals=(1.0, 0.0),
              addend=None, sep=None, mask_bits=False):
    """x h8; pad = (x0, x1, y0, y1) as in op/upfirdn2d.cpp:12-23; optional fused epilogue act(fir(x) + noise*noise_w + bias[c]) * gain, then
    * (mask > 0 ? mask_vals[0] : mask_vals[1]) and + addend (both h8, shaped like the output).  ``sep`` = separable(kernel) (4x4, no resampling): the
    register-streaming separable kernel."""
    lib = _lib.load()
    B, G8, H, W, _ = x.shape
    kh, kw = kernel.shape
    oh = (H * up + pad[2] + pad[3] - kh) // down + 1
    ow = (W * up + pad[0] + pad[1] - kw) // down + 1
    y = torch.empty(B, G8, oh, ow, 8, device=x.device, dtype=x.dtype)
    use_sep = sep is not None and kh == 4 and kw == 4 and not os.environ.get('L2I_H8_NOSEP')            # the library picks its separable register-streaming kernels (no resampling, down 2, up 2) where they apply
    k1y = (ctypes.c_float * 4)(*sep[0]) if use_sep else None
    k1x = (ctypes.c_float * 4)(*sep[1]) if use_sep else None
    _lib.check(_fn(lib, 'l2i_upfirdn2d_h8', x.dtype)(_lib.ptr(y), _h8(x), _lib.fptr(kernel.contiguous()), B * G8, G8 * 8, H, W, kh, kw, up, down, pad[0], pad[1], pad[2], pad[3],
                                    _lib.fptr(noise), float(noise_w), _lib.fptr(bias), int(act), float(slope), float(gain), None if mask is None else (_lib.ptr(mask) if mask_bits else _h8(mask)),
                                    float(mask_vals[0]), float(mask_vals[1]), None if addend is None else _h8(addend), k1y, k1x, int(bool(mask_bits)), _lib.stream_ptr()), 'l2i_upfirdn2d_h8')
    assert (mask is None or tuple(mask.shape) == (tuple(y.shape[:4]) if mask_bits else tuple(y.shape))) and (addend is None or addend.shape == y.shape)
    return y


def torgb_fwd(x, wmod, bias):
    """x h8 [B,C/8,H,W,8], wmod [B,3,C] fp32, bias [3] -> rgb fp32 [B,3,H,W]."""
    lib = _lib.load()
    B, G8, H, W, _ = x.shape
    rgb = torch.empty(B, 3, H, W, device=x.device, dtype=torch.float32)
    _lib.check(_fn(lib, 'l2i_torgb_fwd_h8', x.dtype)(_lib.fptr(rgb), _h8(x), _lib.fptr(wmod.contiguous()), _lib.fptr(bias), B, G8 * 8, H * W, _lib.stream_ptr()), 'l2i_torgb_fwd_h8')
    return rgb


def sg2_act_bwd(y, gin, gin_scale, grgb, wmod_rgb, bias, noise, noise_w, slope, gain, red, red_rgb=None, red_q=None):
    """Fused StyledConv elementwise backward on h8 maps (l2i_sg2_act_bwd_h8): returns dz (h8); ``red`` [B,C] / ``red_rgb`` [B,C,3] / ``red_q`` ([r5] [B*C]: sum_p gin * y, the next layer's style gradient) are ZEROED fp32 buffers."""
    lib = _lib.load()
    B, G8, H, W, _ = y.shape
    dz = torch.empty_like(y)
    _lib.check(_fn(lib, 'l2i_sg2_act_bwd_h8', y.dtype)(_lib.ptr(dz), None if gin is None else _h8(gin), _lib.fptr(gin_scale), _lib.fptr(grgb), _lib.fptr(wmod_rgb), _h8(y), _lib.fptr(bias),
                                      _lib.fptr(noise), float(noise_w), float(slope), float(gain), _lib.fptr(red), _lib.fptr(red_rgb), _lib.fptr(red_q), B, G8 * 8, H * W,
                                      _lib.stream_ptr()), 'l2i_sg2_act_bwd_h8')
    return dz


def dot_reduce(a, b=None, out=None):
    """out[b,c] += sum_p a*b on h8 maps; ``out``: zeroed fp32 [B*C] (allocated when absent).  Returns [B, C]."""
    lib = _lib.load()
    B, G8, H, W, _ = a.shape
    if out is None:
        out = torch.zeros(B * G8 * 8, device=a.device, dtype=torch.float32)
    _lib.check(_fn(lib, 'l2i_dot_reduce_h8', a.dtype)(_lib.fptr(out), _h8(a), None if b is None else _h8(b), B, G8 * 8, H * W, _lib.stream_ptr()), 'l2i_dot_reduce_h8')
    return out.view(B, G8 * 8)


def maxpool2d_fwd(x, k, s, pad, relu=False):
    lib = _lib.load()
    B, G8, H, W, _ = x.shape
    oh, ow = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    y = torch.empty(B, G8, oh, ow, 8, device=x.device, dtype=x.dtype)
    idx = torch.empty(B, G8, oh, ow, 8, device=x.device, dtype=torch.uint8)
    _lib.check(_fn(lib, 'l2i_maxpool2d_fwd_h8', x.dtype)(_lib.ptr(y), _lib.ptr(idx), _h8(x), B * G8, H, W, k, s, pad, oh, ow, int(relu), _lib.stream_ptr()), 'l2i_maxpool2d_fwd_h8')
    return y, idx


def maxpool2d_bwd(gy, idx, in_hw, k, s, pad, a=None, b=None, coef=0.0, coef_dev=None):
    """Pool backward on h8 maps; a / b (h8, input-sized): + coef * coef_dev[0] * (b - a) in the same pass."""
    lib = _lib.load()
    B, G8, oh, ow, _ = gy.shape
    gx = torch.empty(B, G8, in_hw[0], in_hw[1], 8, device=gy.device, dtype=gy.dtype)
    _lib.check(_fn(lib, 'l2i_maxpool2d_bwd_h8', gy.dtype)(_lib.ptr(gx), _h8(gy), _lib.ptr(idx), None if a is None else _h8(a), None if b is None else _h8(b), float(coef), _lib.fptr(coef_dev),
                                        B * G8, in_hw[0], in_hw[1], k, s, pad, oh, ow, _lib.stream_ptr()), 'l2i_maxpool2d_bwd_h8')
    return gx


def sqdiff(a, b, coef=0.0, want_grad=False, coef_dev=None, want_sum=True):
    lib = _lib.load()
    s = torch.zeros(1, device=a.device, dtype=torch.float32) if want_sum else None
    g = torch.empty_like(b) if want_grad else None
    _lib.check(_fn(lib, 'l2i_sqdiff_h8', a.dtype)(_lib.fptr(s), None if g is None else _lib.ptr(g), _h8(a), _h8(b), a.numel() // 8, float(coef), _lib.fptr(coef_dev), _lib.stream_ptr()), 'l2i_sqdiff_h8')
    return s, g


def add_zero_insert(y, c, mask=None):
    """y[.., 2oy, 2ox, :] += c[.., oy, ox, :] * (mask[.., 2oy, 2ox, :] > 0 if a mask is given) in place (h8)."""
    lib = _lib.load()
    B, G8, H, W, _ = y.shape
    _lib.check(_fn(lib, 'l2i_add_zero_insert_h8', y.dtype)(_h8(y), _h8(c), None if mask is None else _h8(mask), B * G8, H, W, c.shape[2], c.shape[3], _lib.stream_ptr()), 'l2i_add_zero_insert_h8')
    return y


def mask_mul(g, ref, pos=1.0, neg=0.0):
    """g * (ref > 0 ? pos : neg) on h8 maps; ``ref`` may be the map or ([r6]) its sign plane (uint8 [B, C/8, H, W]: l2i_conv_params::mask_out)."""
    lib = _lib.load()
    y = torch.empty_like(g)
    if ref.dtype == torch.uint8:
        assert tuple(ref.shape) == tuple(g.shape[:4]) and ref.is_contiguous()
        _lib.check(_fn(lib, 'l2i_mask_mul_bits_h8', g.dtype)(_lib.ptr(y), _h8(g), _lib.ptr(ref), float(pos), float(neg), g.numel() // 8, _lib.stream_ptr()), 'l2i_mask_mul_bits_h8')
        return y
    _lib.check(_fn(lib, 'l2i_mask_mul_h8', g.dtype)(_lib.ptr(y), _h8(g), _h8(ref), float(pos), float(neg), g.numel() // 8, _lib.stream_ptr()), 'l2i_mask_mul_h8')
    return y


def modulate_planes(w32, s, dtype=None):
    """w32: fp32 weights in plane order [Cin/16, KK, 2, CoutP, 8] (conv.pack_weight_h8_f32); s [B, Cin] fp32 -> 16-bit planes [B, Cin/16, KK, 2, CoutP, 8]
    (int16 view) in the element type of the path (or ``dtype``)."""
    lib = _lib.load()
    dtype = dtype or h8_dtype()
    B = s.shape[0]
    c16, kk, _, coutp, _ = w32.shape
    assert s.shape[1] == c16 * 16 and s.is_contiguous()
    planes = torch.empty((B,) + tuple(w32.shape), device=w32.device, dtype=torch.int16)
    _lib.check(_fn(lib, 'l2i_modulate_planes_h8', dtype)(_lib.ptr(planes), _lib.fptr(w32), _lib.fptr(s), B, c16 * 16, c16 * 16, kk, coutp, _lib.stream_ptr()), 'l2i_modulate_planes_h8')
    return planes


class ModulatePlan:
    """[r5] The weight planes of EVERY modulated conv of a generator pass in one launch (l2i_modulate_planes_multi_h8): ``w32s`` = the layers' fp32
    weights in plane order (conv.pack_weight_h8_f32), ``s_offsets`` = rows before each layer's block in the per-layer-contiguous scale buffer
    ([layer][B][C], generator._ModPlan: s_all for the forward, demod_all for the backward).  ``run(scales, B)`` returns one int16 view
    [B, Cin/16, KK, 2, CoutP, 8] per layer of one buffer."""
    BLOCK_SLOTS = 256 * 8                      # slots a block handles at most per pass of its grid-stride loop

    def __init__(self, w32s, s_offsets, device, widths=None):
        self.shapes = [tuple(w.shape) for w in w32s]
        if widths is not None:                   # rows of the scale buffer per layer: the kernel reads c16 * 16 scales per sample, like modulate_planes asserts
            for w, n in zip(w32s, widths):
                assert n == w.shape[0] * 16, 'scale row of %d entries for a weight plane set of %d padded input channels' % (n, w.shape[0] * 16)
        self.w32 = torch.cat([w.reshape(-1) for w in w32s]).contiguous().to(device)
        self.w_off = [0]
        for w in w32s[:-1]:
            self.w_off.append(self.w_off[-1] + w.numel())
        self.s_off = list(s_offsets)
        self.sps = [w.numel() // 8 for w in w32s]
        self.device = device
        self._tables = {}

    def _table(self, B):
        if B not in self._tables:
            rows, out_off, first = [], 0, 0
            for (c16, kk, _, coutp, _), w_off, s_off, sps in zip(self.shapes, self.w_off, self.s_off, self.sps):
                rows.append([w_off, s_off * B, out_off, sps, kk, coutp, c16 * 16, first])
                out_off += sps * B
                first += max(1, min((sps * B + self.BLOCK_SLOTS - 1) // self.BLOCK_SLOTS, 1024))
            self._tables[B] = (torch.tensor(rows, dtype=torch.int64).to(self.device), out_off, first)
        return self._tables[B]

    def run(self, scales, B, dtype=None):
        lib = _lib.load()
        dtype = dtype or h8_dtype()
        table, slots, nblocks = self._table(B)
        planes = torch.empty(slots * 8, device=self.w32.device, dtype=torch.int16)
        _lib.check(_fn(lib, 'l2i_modulate_planes_multi_h8', dtype)(_lib.ptr(planes), _lib.fptr(self.w32), _lib.fptr(scales), _lib.ptr(table), len(self.shapes), B, nblocks, _lib.stream_ptr()),
                   'l2i_modulate_planes_multi_h8')
        views, off = [], 0
        for shp, sps in zip(self.shapes, self.sps):
            views.append(planes[off * 8:(off + sps * B) * 8].view((B,) + shp))
            off += sps * B
        return views
