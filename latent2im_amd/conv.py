"""Frozen 2-D convolutions on the MI355X matrix cores: weight packing, launch plans (forward and input-gradient)
and the thin Python call into ``l2i_conv2d_f32``.

Every dense contraction of the walk-training path is a *frozen-weight* convolution (the only trainable tensor
is the walk, reference transform_base.py:329-331), so weights are packed once at construction into the
K-major layout the kernel streams (``[Cin][KH*KW][CoutP]``) — for the forward pass and, separately, for the
input-gradient pass (transposed / flipped / split into stride-2 phases).  No weight-gradient is ever computed.
"""

import numpy as np
import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ConvParams  # noqa: F401


import os as _os

PRECISION = _os.environ.get('L2I_PRECISION', 'f32')      # 'bf16x3': eligible stride-1 layers take the split-precision bf16 MFMA kernel (opt-in);
                                                          # 'bf16' / 'f16': the 16-bit path (nets16.py) with bf16 / IEEE fp16 h8 feature maps
H8_PRECISIONS = ('bf16', 'f16')


def h8_dtype():
    """Element type of the 16-bit path's h8 tensors under the current PRECISION ([r5]: 'f16' = IEEE fp16, everything else bf16)."""
    return torch.float16 if PRECISION == 'f16' else torch.bfloat16

USE_WINOGRAD = _os.environ.get('L2I_WINOGRAD', '1') != '0'    # 3x3 stride-1 layers on maps >= 32 wide take the F(2x2,3x3) fp32 kernel
# [r4] Winograd F(4x4,3x3) (csrc/l2i_wino4.hip: 1.78x fewer MFMAs than F(2x2), error ~1e-6..1e-5 of max|y| instead of 3e-7) for the unmasked 3x3
# stride-1 launches on maps >= 32 wide ([r5]: 32 x 16-pixel tiles below 64): 'all' = every eligible launch on the [r5] position-split kernel, 'r4' = those >= 64 wide on the round-4
# kernel (kept for A/B runs: the two are bit-identical), 'off' = F(2x2) everywhere.  The parity suite runs 'all' and 'off'.
WINO4_MODES = ('all', 'tall', 'r4', 'off')        # 'tall' ([r5], A/B): the position-split kernel on 64 x 16-pixel tiles / eight waves where the map has >= 16 rows
WINO4 = _os.environ.get('L2I_WINO4', 'all')
WINO4_R4_MIN_W = 64      # narrowest map the 'r4' mode sends to the round-4 kernel (its tile is 64 wide; the bit-identity test lowers this)
if WINO4 not in WINO4_MODES:
    raise ValueError('L2I_WINO4 must be one of %s, got %r' % (WINO4_MODES, WINO4))
SPLIT_K = _os.environ.get('L2I_SPLIT_K', '1') != '0'    # 4x4 .. 16x16 maps: cut Cin into ranges computed by separate blocks (l2i.h: ksplit / ws)
_WS = {}            # split-K workspaces, one per (device, stream)
USE_FUSED_TRANSPOSED = True     # False: issue stride-2 transposed convs as four per-parity launches
PROFILE = None      # bench.py sets this to a list: every launch then appends (start_event, end_event, algorithmic_flops, shape, entry point, kernel family)
FAMILIES = ('implicit_gemm_f32', 'gemm1x1_f32', 'cin3_f32', 'direct_small_valu')      # l2i.h: L2I_FAMILY_* of l2i_conv2d_f32
# kernel family -> (kernel name in rocprof, MFMA FLOPs executed per algorithmic FLOP, peak TFLOP/s of the instruction it runs on)
FAMILY_INFO = {
    'winograd_f32': ('conv_wino_kernel', 16.0 / 36.0, 157.3),
    'winograd4_f32': ('conv_wino4s_kernel', 36.0 / 144.0, 157.3),
    'implicit_gemm_f32': ('conv_mfma_kernel / conv3x3s2_dma_kernel', 1.0, 157.3),
    'gemm1x1_f32': ('gemm1x1_kernel', 1.0, 157.3),
    'transposed_f32': ('convt_mfma_kernel', 1.0, 157.3),
    'cin3_f32': ('conv_cin3_kernel', 28.0 / 27.0, 157.3),
    'direct_small_valu': ('conv_direct_small_kernel', 1.0, 157.3),
    'implicit_gemm_bf16x3': ('conv_bf16x3_pipe_kernel', 3.0, 2500.0),
    'transposed_bf16x3': ('conv_bf16x3_pipe_kernel<TR>', 3.0, 2500.0),
    'conv_h8': ('conv_h8_kernel', 1.0, 2500.0),
    'transposed_h8': ('conv_h8_kernel<TR>', 1.0, 2500.0),
}


def pack_weight(w):
    """[Cout, Cin, KH, KW] -> packed [Cin, KH*KW, CoutP] (CoutP = Cout rounded up to 32, zero padded)."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    coutp = (cout + 31) // 32 * 32
    p = torch.zeros(cin, kh * kw, coutp, dtype=torch.float32, device=w.device)      # packs where the weight lives (trainable nets repack on the GPU)
    p[:, :, :cout] = w.permute(1, 2, 3, 0).reshape(cin, kh * kw, cout)
    return p.contiguous()


def pack_weight_bf16x3(w):
    """[Cout, Cin, KH, KW] fp32 -> (hi, lo) bf16 planes laid out [Cin/16][KH*KW][2 (8-channel half)][CoutP][8] (as int16 tensors):
    hi = bf16(w) (round to nearest even), lo = bf16(w - hi).  This is the LDS image order of csrc/l2i_conv16.hip: the rows of a
    (16-channel group, tap, half) are CoutP consecutive 16-byte slots, so a block's slice goes global -> LDS by DMA."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    assert cin % 16 == 0
    coutp = (cout + 31) // 32 * 32
    full = torch.zeros(coutp, cin, kh, kw, dtype=torch.float32, device=w.device)
    full[:cout] = w
    hi = full.to(torch.bfloat16)
    lo = (full - hi.float()).to(torch.bfloat16)

    def lay(t):          # [CoutP, Cin, KH, KW] -> [Cin/16, KH*KW, 2, CoutP, 8]
        t = t.reshape(coutp, cin // 16, 2, 8, kh * kw).permute(1, 4, 2, 0, 3).contiguous()
        return t.view(torch.int16)
    return lay(hi), lay(lo)


_WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def pack_weight_wino(w):
    """[Cout, Cin, 3, 3] -> Winograd F(2x2,3x3) weights U = G g G^T (computed in float64, stored fp32) laid out
    [Cin, 4 (i), CoutP, 4 (j)]: the kernel reads one float4 = the four j positions of row i for one (cin, cout)."""
    w = torch.as_tensor(w, dtype=torch.float64)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    G = torch.as_tensor(_WINO_G, device=w.device)
    U = torch.einsum('ik,ockl,jl->ocij', G, w, G)                 # [Cout, Cin, 4, 4]
    coutp = (cout + 31) // 32 * 32
    p = torch.zeros(cin, 4, coutp, 4, dtype=torch.float32, device=w.device)
    p[:, :, :cout, :] = U.permute(1, 2, 0, 3).float()
    return p.contiguous()


_WINO4_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                    dtype=np.float64)


def pack_weight_wino4(w):
    """[Cout, Cin, 3, 3] -> Winograd F(4x4,3x3) weights U = G g G^T (6x6, computed in float64, stored fp32) in the LDS image order of
    csrc/l2i_wino4.hip: [Cin/4][CoutP/16][ [6 i][4 cin][16 cout][4 (j = 0..3)] ++ [6 i][4 cin][16 cout][2 (j = 4, 5)] ], CoutP = Cout rounded
    up to 32 — a block's slice of a 4-channel chunk (two consecutive 16-channel images: 18432 contiguous bytes) goes global -> LDS by DMA."""
    w = torch.as_tensor(w, dtype=torch.float64)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 4 == 0
    G = torch.as_tensor(_WINO4_G, device=w.device)
    coutp = (cout + 31) // 32 * 32
    U = torch.zeros(coutp, cin, 6, 6, dtype=torch.float64, device=w.device)
    U[:cout] = torch.einsum('ik,ockl,jl->ocij', G, w, G)
    U = U.reshape(coutp // 16, 16, cin // 4, 4, 6, 6).permute(2, 0, 4, 3, 1, 5)          # [c4, mb, i, k, m, j]
    a = U[..., :4].reshape(cin // 4, coutp // 16, -1)
    b = U[..., 4:].reshape(cin // 4, coutp // 16, -1)
    return torch.cat([a, b], dim=2).float().contiguous()


class Launch:
    """One call of the kernel: a stride-1/2 correlation writing every (oy_step, ox_step)-th output pixel."""
    __slots__ = ('w', 'cin', 'cout', 'kh', 'kw', 'stride', 'pad_y', 'pad_x', 'step', 'off_y', 'off_x', 'w16', 'w_src', 'wino', 'w4', 'wino4')

    def __init__(self, w_oihw, stride, pad_y, pad_x, step=1, off_y=0, off_x=0, device=None):
        self.cout, self.cin, self.kh, self.kw = w_oihw.shape
        self.w = pack_weight(w_oihw).to(device) if device is not None else pack_weight(w_oihw)
        self.stride, self.pad_y, self.pad_x = stride, pad_y, pad_x
        self.step, self.off_y, self.off_x = step, off_y, off_x
        self.w16 = None                                        # (hi, lo) planes, built on first bf16x3 use
        self.w4 = None                                         # <= 4 output channels: dense [Cin][KH*KW][4] pack for the direct VALU kernel
        if self.cout <= 4 and stride == 1:
            self.w4 = torch.zeros(self.cin, self.kh * self.kw, 4, dtype=torch.float32, device=self.w.device)
            self.w4[:, :, :self.cout] = self.w[:, :, :self.cout]
        self.wino = None                                       # Winograd pack, built on first use
        self.wino4 = None                                      # F(4x4,3x3) pack, built on first use
        self.w_src = torch.as_tensor(w_oihw, dtype=torch.float32) if (self.cin % 8 == 0 and self.kh <= 3 and self.kw <= 3) else None

    def bf16x3_planes(self):
        if self.w16 is None and self.w_src is not None:
            hi, lo = pack_weight_bf16x3(self.w_src)
            self.w16 = (hi.to(self.w.device), lo.to(self.w.device))
        return self.w16

    def wino_pack(self):
        if self.wino is None and self.w_src is not None and self.kh == 3 and self.kw == 3:
            self.wino = pack_weight_wino(self.w_src).to(self.w.device)
        return self.wino

    def wino4_pack(self):
        if self.wino4 is None and self.w_src is not None and self.kh == 3 and self.kw == 3:
            self.wino4 = pack_weight_wino4(self.w_src).to(self.w.device)
        return self.wino4

    def to(self, device):
        self.w = self.w.to(device)
        if self.w4 is not None:
            self.w4 = self.w4.to(device)
        if self.wino is not None:
            self.wino = self.wino.to(device)
        if self.wino4 is not None:
            self.wino4 = self.wino4.to(device)
        if self.w16 is not None:
            self.w16 = tuple(t.to(device) for t in self.w16)
        return self


def correlation_plan(w_oihw, stride, pad):
    """y[co,o] = sum x[ci, o*stride - pad + k] * w[co,ci,k]."""
    return [Launch(torch.as_tensor(w_oihw, dtype=torch.float32), stride, pad, pad)]


def _phase_axis(K, pad, pi):
    """Taps of a stride-2 transposed conv (o = 2i + k - pad) that land on outputs of parity ``pi``:
    returns (tap indices in correlation order, correlation padding) or (None, 0) when the phase is empty."""
    k0 = (pi + pad) % 2
    ks = list(range(k0, K, 2))
    if not ks:
        return None, 0
    A = len(ks)
    d = (pi + pad - k0) // 2
    taps = [k0 + 2 * (A - 1 - a) for a in range(A)]     # w'[a] = w[k0 + 2(A-1-a)]
    return taps, (A - 1) - d


def transposed_plan(w_oihw, pad):
    """Stride-2 transposed convolution y[co, 2i+k-pad] += x[ci,i]*w[co,ci,k] as <= 4 stride-1 correlations, one per
    output parity (same MAC count as the dense form; nothing multiplies an inserted zero)."""
    w = torch.as_tensor(w_oihw, dtype=torch.float32)
    K = w.shape[2]
    assert w.shape[3] == K
    plan = []
    for py in (0, 1):
        ty, pad_y = _phase_axis(K, pad, py)
        for px in (0, 1):
            tx, pad_x = _phase_axis(K, pad, px)
            if ty is None or tx is None:
                plan.append(None)
                continue
            sub = w[:, :, ty, :][:, :, :, tx].contiguous()
            plan.append(Launch(sub, 1, pad_y, pad_x, step=2, off_y=py, off_x=px))
    return plan


FUSED_TRANSPOSED_SHAPES = ((3, 0), (3, 1), (7, 3))        # (K, pad) instantiated in csrc/l2i_convt.hip
_FUSED_KW = {'in_scale', 'in_mask', 'mask', 'out_scale', 'out_gain', 'tile_hint'}


def _tr_geom(K, pad):
    k0 = [(pi + pad) % 2 for pi in (0, 1)]
    A = [(K - k0[pi] + 1) // 2 for pi in (0, 1)]
    d = [(pi + pad - k0[pi]) // 2 for pi in (0, 1)]
    cp = [A[pi] - 1 - d[pi] for pi in (0, 1)]
    P = max(cp)
    Q = max(A[pi] - 1 - cp[pi] for pi in (0, 1))
    return k0, A, cp, P, Q


def fused_transposed_taps(K, pad):
    """[(ky, kx)] in the order the fused kernel walks the taps: for dy, dx (input offsets) / for py, px (output parities)."""
    k0, A, cp, P, Q = _tr_geom(K, pad)
    taps = []
    for dy in range(-P, Q + 1):
        for dx in range(-P, Q + 1):
            for py in (0, 1):
                for px in (0, 1):
                    ay, ax = dy + cp[py], dx + cp[px]
                    if 0 <= ay < A[py] and 0 <= ax < A[px]:
                        taps.append((k0[py] + 2 * (A[py] - 1 - ay), k0[px] + 2 * (A[px] - 1 - ax)))
    assert len(taps) == K * K and len(set(taps)) == K * K
    return taps


class FusedTransposed:
    """One-launch stride-2 transposed conv (l2i_conv_transpose2d_f32): weights packed in the kernel's tap order; for the split-precision
    path (l2i_conv_transpose2d_bf16x3_f32) the bf16 planes of the plain [Cout, Cin, 3, 3] weight, built on first use."""
    __slots__ = ('w', 'cin', 'cout', 'k', 'pad', 'w_src', 'w16')

    def __init__(self, w_oihw, pad):
        w = torch.as_tensor(w_oihw, dtype=torch.float32)
        self.cout, self.cin, self.k, _ = w.shape
        self.pad = pad
        taps = fused_transposed_taps(self.k, pad)
        sel = torch.stack([w[:, :, ky, kx] for ky, kx in taps], 2)               # [Cout, Cin, K*K]
        self.w = pack_weight(sel.reshape(self.cout, self.cin, self.k * self.k, 1))
        self.w_src = w if (self.k == 3 and pad in (0, 1) and self.cin % 16 == 0) else None
        self.w16 = None

    def bf16x3_planes(self):
        if self.w16 is None and self.w_src is not None:
            hi, lo = pack_weight_bf16x3(self.w_src)
            self.w16 = (hi.to(self.w.device), lo.to(self.w.device))
        return self.w16

    def to(self, device):
        self.w = self.w.to(device)
        return self


class SmallTransposed:
    """Stride-2 transposed 7x7 / pad 3 conv onto <= 3 channels in one launch (csrc/l2i_convt_small.hip: the ResNet-50 stem's input-gradient):
    weights [Cin][49][4] (three output channels + pad)."""
    __slots__ = ('w', 'cin', 'cout', 'k', 'pad')

    def __init__(self, w_oihw, pad):
        w = torch.as_tensor(w_oihw, dtype=torch.float32)
        self.cout, self.cin, self.k, _ = w.shape
        assert self.k == 7 and pad == 3 and self.cout <= 3
        self.pad = pad
        pk = torch.zeros(self.cin, 49, 4, dtype=torch.float32)
        pk[:, :, :self.cout] = w.permute(1, 2, 3, 0).reshape(self.cin, 49, self.cout)
        self.w = pk.contiguous()

    def to(self, device):
        self.w = self.w.to(device)
        return self


def run_small_transposed(F, x, y, in_mask=None, mask=(1.0, 0.0), out_gain=1.0):
    lib = _lib.load()
    h8 = x.dim() == 5                                      # [r5] 16-bit h8 gradient (and mask) in: l2i_conv_params::in_h8
    if h8:
        B, cg, H, W, _ = x.shape
        cin = cg * 8
        assert x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous() and (in_mask is None or (in_mask.dtype == x.dtype and in_mask.is_contiguous()))
    else:
        B, cin, H, W = x.shape
    assert cin == F.cin and y.shape[0] == B and y.shape[1] == F.cout
    p = ConvParams()
    p.x, p.w, p.y = (_lib.ptr(x) if h8 else _lib.fptr(x)), _lib.fptr(F.w), _lib.fptr(y)
    p.in_h8 = 0 if not h8 else (2 if x.dtype == torch.float16 else 1)
    p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = B, cin, H, W, F.cout, 4
    p.KH = p.KW = F.k
    p.stride, p.pad_y, p.pad_x = 2, F.pad, F.pad
    p.OHf, p.OWf = y.shape[2], y.shape[3]
    p.OH, p.OW = (p.OHf + 1) // 2, (p.OWf + 1) // 2
    p.oy_step = p.ox_step = 2
    p.in_mask = _lib.ptr(in_mask) if h8 else _lib.fptr(in_mask)
    p.mask_pos, p.mask_neg = mask
    p.act_gain, p.out_gain = 1.0, out_gain
    if in_mask is not None:
        assert in_mask.shape == x.shape
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.l2i_conv_transpose2d_f32(p, _lib.stream_ptr()), 'l2i_conv_transpose2d_f32')
        e1.record()
        PROFILE.append((e0, e1, 2.0 * B * F.cout * cin * F.k * F.k * H * W,
                        (B, cin, F.cout, F.k, F.k, 2, H, W, int(p.OHf), int(p.OWf), 2, in_mask is not None, False), 'l2i_conv_transpose2d_f32', 'direct_small_valu'))
        return y
    _lib.check(lib.l2i_conv_transpose2d_f32(p, _lib.stream_ptr()), 'l2i_conv_transpose2d_f32')
    return y


def _split_k(p, px, cin, y):
    """Small maps (<= 2048 positions per launch) with many input channels: cut Cin into ranges computed by separate blocks."""
    if not (SPLIT_K and px <= 2048 and cin >= 256):
        return
    ks = 16 if px <= 128 else (8 if px <= 512 else 4)
    while ks > 1 and (cin % ks or (cin // ks) % 2 or cin // ks < 16):
        ks //= 2
    if ks > 1:
        need = ks * y.numel()
        key = (y.device, torch.cuda.current_stream(y.device).cuda_stream)      # the loss branches run on their own streams
        ws = _WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _WS[key] = torch.empty(max(need, 1 << 22), device=y.device, dtype=torch.float32)
        p.ws, p.ksplit = _lib.fptr(ws), ks


def run_fused_transposed(F, x, y, in_scale=None, in_mask=None, mask=(1.0, 0.0), out_scale=None, out_gain=1.0, tile_hint=0):
    lib = _lib.load()
    B, cin, H, W = x.shape
    assert cin == F.cin and y.shape[0] == B and y.shape[1] == F.cout
    p = ConvParams()
    p.x, p.w, p.y = _lib.fptr(x), _lib.fptr(F.w), _lib.fptr(y)
    p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = B, cin, H, W, F.cout, F.w.shape[2]
    p.KH = p.KW = F.k
    p.stride, p.pad_y, p.pad_x = 2, F.pad, F.pad
    p.OHf, p.OWf = y.shape[2], y.shape[3]
    p.OH, p.OW = (p.OHf + 1) // 2, (p.OWf + 1) // 2
    p.oy_step = p.ox_step = 2
    p.in_scale, p.in_mask = _lib.fptr(in_scale), _lib.fptr(in_mask)
    p.mask_pos, p.mask_neg = mask
    p.out_scale = _lib.fptr(out_scale)
    p.act_gain, p.out_gain = 1.0, out_gain
    p.tile_hint = tile_hint
    if in_mask is not None:
        assert in_mask.shape == x.shape
    entry, name, family = lib.l2i_conv_transpose2d_f32, 'l2i_conv_transpose2d_f32', 'transposed_f32'
    nat_h, nat_w = (H - 1) * 2 - 2 * F.pad + F.k, (W - 1) * 2 - 2 * F.pad + F.k
    if (PRECISION == 'bf16x3' and tile_hint == 0 and F.w_src is not None and W % 4 == 0 and W >= 32 and x.data_ptr() % 16 == 0
            and (in_mask is None or in_mask.data_ptr() % 16 == 0) and 0 <= p.OHf - nat_h <= 8 and 0 <= p.OWf - nat_w <= 8):
        planes = F.bf16x3_planes()
        p.w_hi, p.w_lo = _lib.ptr(planes[0]), _lib.ptr(planes[1])
        entry, name, family = lib.l2i_conv_transpose2d_bf16x3_f32, 'l2i_conv_transpose2d_bf16x3_f32', 'transposed_bf16x3'
    else:
        _split_k(p, B * H * W, cin, y)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(entry(p, _lib.stream_ptr()), name)
        e1.record()
        PROFILE.append((e0, e1, 2.0 * B * F.cout * cin * F.k * F.k * H * W,
                        (B, cin, F.cout, F.k, F.k, 2, H, W, int(p.OHf), int(p.OWf), 2, in_mask is not None, in_scale is not None), name, family))
        return y
    _lib.check(entry(p, _lib.stream_ptr()), name)
    return y


def run_launch(L, x, y, out_hw=None, in_scale=None, in_mask=None, mask=(1.0, 0.0), out_scale=None, noise=None,
               noise_w=0.0, bias=None, residual=None, res_mask=None, out_mask=None, act=ACT_NONE, slope=0.2, gain=1.0,
               out_gain=1.0, accumulate=False, tile_hint=0, res_sub=None, res_coef=1.0, res_coef_dev=None, sq=None, pool=None, _defer=None):
    """Enqueue one kernel call on the current stream.  ``_defer`` (a list): a plain 1x1 stride-1 launch appends its struct instead (``launch_pair_f32``).  ``y`` is the full output tensor [B, Cout, OHf, OWf].  ``pool`` ([r5]) = (pooled [B, Cout, OHf/2, OWf/2]
    fp32, arg-max bytes of the same shape, [fused flag]): where the launch takes the position-split F(4x4) kernel it also writes MaxPool2d(2, 2) of y
    (l2i.h: pool_out / pool_idx) and sets the flag; otherwise the caller runs the pool kernel."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    assert cin == L.cin, (cin, L.cin)
    assert y.shape[0] == B and y.shape[1] == L.cout, (y.shape, B, L.cout)
    OHf, OWf = y.shape[2], y.shape[3]
    if L.step == 1:
        OH, OW = OHf, OWf
    else:
        OH = (OHf - L.off_y + L.step - 1) // L.step
        OW = (OWf - L.off_x + L.step - 1) // L.step
    if OH <= 0 or OW <= 0:
        return
    p = ConvParams()
    p.x, p.w, p.y = _lib.fptr(x), _lib.fptr(L.w), _lib.fptr(y)
    p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = B, cin, H, W, L.cout, L.w.shape[2]
    p.KH, p.KW, p.stride, p.pad_y, p.pad_x = L.kh, L.kw, L.stride, L.pad_y, L.pad_x
    p.OH, p.OW, p.OHf, p.OWf = OH, OW, OHf, OWf
    p.oy_step = p.ox_step = L.step
    p.oy_off, p.ox_off = L.off_y, L.off_x
    p.in_scale, p.in_mask = _lib.fptr(in_scale), _lib.fptr(in_mask)
    p.mask_pos, p.mask_neg = mask
    p.out_scale, p.noise, p.noise_w, p.bias = _lib.fptr(out_scale), _lib.fptr(noise), float(noise_w), _lib.fptr(bias)
    p.residual, p.res_mask, p.out_mask = _lib.fptr(residual), _lib.fptr(res_mask), _lib.fptr(out_mask)
    if res_sub is not None:                      # residual term = res_coef * res_coef_dev[0] * (residual - res_sub)
        assert residual is not None and res_sub.shape == residual.shape
        p.res_sub, p.res_coef, p.res_coef_dev = _lib.fptr(res_sub), float(res_coef), _lib.fptr(res_coef_dev)
    p.act, p.act_slope, p.act_gain, p.out_gain = act, slope, gain, out_gain
    p.accumulate, p.tile_hint = int(accumulate), tile_hint
    if in_mask is not None:
        assert in_mask.shape == x.shape
    if residual is not None:
        assert residual.shape == y.shape
    if out_mask is not None:
        assert out_mask.shape == y.shape
    entry, name = lib.l2i_conv2d_f32, 'l2i_conv2d_f32'
    if _defer is not None:
        assert L.kh == 1 and L.kw == 1 and L.stride == 1 and L.step == 1
        _defer.append((p, (x, y, L.w, bias, residual)))
        return
    if (L.w4 is not None and tile_hint == 0 and out_scale is None and noise is None and bias is None and residual is None and out_mask is None
            and act == ACT_NONE):                          # the launch takes the direct VALU kernel (l2i_conv2d_family): hand it the dense pack
        p.w, p.CoutP = _lib.fptr(L.w4), 4
    if PRECISION == 'bf16x3' and tile_hint == 0 and _bf16x3_eligible(L, x, in_mask, OW):
        planes = L.bf16x3_planes()
        p.w_hi, p.w_lo = _lib.ptr(planes[0]), _lib.ptr(planes[1])
        entry, name = lib.l2i_conv2d_bf16x3_f32, 'l2i_conv2d_bf16x3_f32'
        if (sq is not None and L.step == 1 and OWf % 4 == 0 and OW % 4 == 0 and sq[0].data_ptr() % 16 == 0
                and _wino_aligned(y, residual, res_mask, out_mask, noise, res_sub)):      # the vectorised epilogue (l2i_epilogue_vec_ok) sums (y - ref)^2 too
            assert sq[0].shape == y.shape and sq[1].numel() == _lib.SQ_SLOTS
            p.sq_ref, p.sq_out = _lib.fptr(sq[0]), _lib.fptr(sq[1])
            sq[2][0] = True
    elif (USE_WINOGRAD and L.kh == 3 and L.kw == 3 and L.stride == 1 and L.step == 1 and L.w_src is not None and L.cout > 4 and OW >= 32
          and OW % 4 == 0 and tile_hint == 0 and _wino_aligned(y, residual, res_mask, out_mask, noise, res_sub)):
        relu_in = in_mask is not None and in_mask.data_ptr() == x.data_ptr() and tuple(mask) == (1.0, 0.0)
        if (WINO4 != 'off' and OW >= (WINO4_R4_MIN_W if WINO4 == 'r4' else 32) and (in_mask is None or relu_in) and L.pad_x == 1 and W % 4 == 0 and cin % 4 == 0
                and cin * H * W * 4 < 0x7FFF0000):             # (one sample below 2 GiB: the kernel's out-of-range sentinel)
            pk = L.wino4_pack()
            p.w, p.CoutP = _lib.fptr(pk), pk.shape[1] * 16
            p.tile_hint = 1 if WINO4 == 'r4' else (2 if WINO4 == 'tall' else 0)          # (A/B: the round-4 kernel / the eight-wave tile on the same pack)
            if (pool is not None and WINO4 != 'r4' and L.step == 1 and OHf % 2 == 0 and OWf % 4 == 0 and OH == OHf and OW == OWf and not accumulate
                    and pool[0].data_ptr() % 8 == 0):
                assert tuple(pool[0].shape) == (B, L.cout, OHf // 2, OWf // 2) and pool[1].shape == pool[0].shape and pool[1].dtype == torch.uint8
                p.pool_out, p.pool_idx = _lib.fptr(pool[0]), _lib.ptr(pool[1])
                pool[2][0] = True
            entry, name = lib.l2i_conv2d_wino4_f32, 'l2i_conv2d_wino4_f32'
        else:
            p.w = _lib.fptr(L.wino_pack())
            entry, name = lib.l2i_conv2d_wino_f32, 'l2i_conv2d_wino_f32'
        if sq is not None and sq[0].data_ptr() % 16 == 0:    # sq = (reference like y, [SQ_SLOTS] zeroed accumulator, [fused flag]): sum (y - ref)^2 in the epilogue
            assert sq[0].shape == y.shape and sq[1].numel() == _lib.SQ_SLOTS
            p.sq_ref, p.sq_out = _lib.fptr(sq[0]), _lib.fptr(sq[1])
            sq[2][0] = True
    elif L.kh * L.kw > 1 and L.cout > 4:
        _split_k(p, B * OH * OW, cin, y)                  # small maps of the generic kernel (the Winograd / split-precision kernels take maps >= 32 wide)
    if (sq is not None and name == 'l2i_conv2d_f32' and tile_hint == 0 and sq[0].data_ptr() % 16 == 0 and p.ksplit <= 1
            and lib.l2i_conv2d_family(p) == 2):           # L2I_FAMILY_CIN3: the <= 3-input-channel kernel sums (y - ref)^2 in its epilogue too
        assert sq[0].shape == y.shape and sq[1].numel() == _lib.SQ_SLOTS
        p.sq_ref, p.sq_out = _lib.fptr(sq[0]), _lib.fptr(sq[1])
        sq[2][0] = True
    if PROFILE is not None:
        family = {'l2i_conv2d_wino_f32': 'winograd_f32', 'l2i_conv2d_wino4_f32': 'winograd4_f32', 'l2i_conv2d_bf16x3_f32': 'implicit_gemm_bf16x3'}.get(name)
        if family is None:
            family = FAMILIES[lib.l2i_conv2d_family(p)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(entry(p, _lib.stream_ptr()), name)
        e1.record()
        PROFILE.append((e0, e1, 2.0 * B * L.cout * cin * L.kh * L.kw * OH * OW,
                        (B, cin, L.cout, L.kh, L.kw, L.stride, H, W, OH, OW, L.step, in_mask is not None, in_scale is not None,
                         ''.join(c for c, t in zip('dnbrmoas', (out_scale, noise, bias, residual, res_mask, out_mask, accumulate or None, res_sub)) if t is not None) + str(act)),
                        name, family))
        return
    _lib.check(entry(p, _lib.stream_ptr()), name)


def _bf16x3_eligible(L, x, in_mask, OW):
    """Mirror of l2i_bf16x3_pipe_eligible (csrc/l2i_conv16.hip): 1x1 (pad 0) / 3x3 (pad 1; stride 2 also pad 0) layers with a dense output
    window on maps >= 32 wide, whole 16-channel (1x1: 32-channel) groups, aligned 4-pixel row vectors.  <= 4 output channels stay on
    the direct VALU kernel, everything else that is not eligible on the fp32 matrix kernels."""
    k1 = L.kh == 1 and L.kw == 1 and L.pad_y == 0 and L.pad_x == 0
    k3 = L.kh == 3 and L.kw == 3 and L.pad_y == L.pad_x and (L.pad_x == 1 or (L.pad_x == 0 and L.stride == 2))
    if k1 and L.stride == 1 and (in_mask is not None or L.cin * L.cout < 65536):
        return False        # HBM-bound 1x1 layers: the DMA-fed fp32 GEMM (no staging registers, no split VALU) streams faster (measured per shape)
    return ((k1 or k3) and L.step == 1 and L.w_src is not None and L.cin % (32 if k1 else 16) == 0 and L.cout > 4 and OW >= 32
            and x.shape[3] % 4 == 0 and x.data_ptr() % 16 == 0 and (in_mask is None or in_mask.data_ptr() % 16 == 0))


def _wino_aligned(*tensors):
    return all(t is None or t.data_ptr() % 16 == 0 for t in tensors)


def run_plan(plan, x, y, accumulate=False, **kw):
    """Run every phase of a plan into ``y``.  Empty phases leave zeros (or the accumulated value) behind."""
    if any(L is None for L in plan) and not accumulate:
        y.zero_()
    for L in plan:
        if L is not None:
            run_launch(L, x, y, accumulate=accumulate, **kw)
    return y


class FrozenConv2d:
    """A convolution (or stride-2 transposed convolution) with constant weights: forward plan + input-gradient plan.

    ``weight`` is [Cout, Cin, K, K] in *correlation* form.  For ``transposed=True`` it means
    y[co, 2i+k-pad] += x[ci, i] * weight[co, ci, k]  (F.conv_transpose2d(x, weight.transpose(0,1), stride=2))."""

    def __init__(self, weight, stride=1, padding=0, transposed=False, device='cuda'):
        w = torch.as_tensor(np.asarray(weight) if not torch.is_tensor(weight) else weight, dtype=torch.float32)
        if not (torch.is_tensor(weight) and weight.is_cuda):        # frozen nets: numpy / CPU weights, packed once on the host
            w = w.cpu()                                             # (a CUDA weight — a net that is being trained — is packed where it lives)
        self.cout, self.cin, self.k, _ = w.shape
        self.stride, self.padding, self.transposed = stride, padding, transposed
        wt = w.transpose(0, 1).contiguous()                       # [Cin, Cout, K, K]: roles swapped for the gradient
        self.fwd_fused = self.bwd_fused = self.bwd_small = None
        fusable = (self.k, padding) in FUSED_TRANSPOSED_SHAPES
        if transposed:
            assert stride == 2
            self.fwd = transposed_plan(w, padding)
            self.bwd = correlation_plan(wt, 2, padding)           # dx[ci,i] = sum gy[co, 2i+k-pad] w[co,ci,k]
            if fusable and self.cin % 4 == 0:            # the fused kernel walks input channels four at a time
                self.fwd_fused = FusedTransposed(w, padding).to(device)
        else:
            self.fwd = correlation_plan(w, stride, padding)
            if stride == 1:
                self.bwd = correlation_plan(torch.flip(wt, [2, 3]), 1, self.k - 1 - padding)
            else:
                assert stride == 2
                self.bwd = transposed_plan(wt, padding)           # dx[ci, 2o+k-pad] += gy[co,o] w[co,ci,k]
                if fusable and self.cin > 4 and self.cout % 4 == 0:          # <= 4 output channels: the VALU kernels
                    self.bwd_fused = FusedTransposed(wt, padding).to(device)
                elif (self.k, padding) == (7, 3) and self.cin <= 3:          # ResNet-50 stem: all four parities in one launch
                    self.bwd_small = SmallTransposed(wt, padding).to(device)
        for L in self.fwd + self.bwd:
            if L is not None:
                L.to(device)

    def out_hw(self, h, w):
        if self.transposed:
            return (h - 1) * 2 - 2 * self.padding + self.k, (w - 1) * 2 - 2 * self.padding + self.k
        return (h + 2 * self.padding - self.k) // self.stride + 1, (w + 2 * self.padding - self.k) // self.stride + 1

    def forward(self, x, out=None, **kw):
        oh, ow = self.out_hw(x.shape[2], x.shape[3])
        if out is None:
            out = torch.empty(x.shape[0], self.cout, oh, ow, device=x.device, dtype=torch.float32)
        if self.fwd_fused is not None and USE_FUSED_TRANSPOSED and set(kw) <= _FUSED_KW:
            return run_fused_transposed(self.fwd_fused, x, out, **kw)
        return run_plan(self.fwd, x, out, **kw)

    def dgrad_h8in(self, gy, in_hw, in_mask=None, mask=(1.0, 0.0), out_gain=1.0):
        """[r5] Input-gradient of the 7x7 / stride 2 stem conv from a 16-bit h8 gradient (and h8 ReLU mask): fp32 NCHW image gradient out
        (l2i_conv_params::in_h8, csrc/l2i_convt_small.hip)."""
        assert self.bwd_small is not None and gy.dim() == 5 and gy.shape[1] * 8 == self.cout
        out = torch.empty(gy.shape[0], self.cin, in_hw[0], in_hw[1], device=gy.device, dtype=torch.float32)
        return run_small_transposed(self.bwd_small, gy, out, in_mask=in_mask, mask=mask, out_gain=out_gain)

    def dgrad(self, gy, in_hw, out=None, **kw):
        """Gradient w.r.t. the input of ``forward`` given the gradient ``gy`` w.r.t. its (pre-epilogue) output."""
        if out is None:
            out = torch.empty(gy.shape[0], self.cin, in_hw[0], in_hw[1], device=gy.device, dtype=torch.float32)
        if self.bwd_fused is not None and USE_FUSED_TRANSPOSED and set(kw) <= _FUSED_KW:
            return run_fused_transposed(self.bwd_fused, gy, out, **kw)
        if (self.bwd_small is not None and USE_FUSED_TRANSPOSED and set(kw) <= {'in_mask', 'mask', 'out_gain'} and gy.shape[3] % 4 == 0 and out.shape[3] % 4 == 0
                and gy.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and (kw.get('in_mask') is None or kw['in_mask'].data_ptr() % 16 == 0)):
            return run_small_transposed(self.bwd_small, gy, out, **kw)
        return run_plan(self.bwd, gy, out, **kw)


# ------------------------------------------------------------------------------------------------------------------------------------
# The 16-bit path (BASELINE config 5): bf16 tensors in the channel-blocked "h8" layout [B, C/8, H, W, 8] (include/l2i.h: l2i_conv2d_h8)
# ------------------------------------------------------------------------------------------------------------------------------------
def to_h8(x, pad_to=8, dtype=None):
    """fp32 / 16-bit NCHW -> 16-bit h8 [B, C/8, H, W, 8] (channels zero-padded to a multiple of ``pad_to``) in the path's element type (or ``dtype``).
    A torch reshuffle: used at the few fp32 boundaries of the 16-bit path and by the tests, not inside the conv stack (the kernels read and write
    h8 directly)."""
    B, C, H, W = x.shape
    cp = (C + pad_to - 1) // pad_to * pad_to
    if cp != C:
        x = torch.cat([x, x.new_zeros(B, cp - C, H, W)], 1)
    return x.reshape(B, cp // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(dtype or h8_dtype())


def from_h8(t, channels=None):
    """bf16 h8 [B, C/8, H, W, 8] -> fp32 NCHW (the first ``channels`` channels)."""
    B, G8, H, W, _ = t.shape
    x = t.float().permute(0, 1, 4, 2, 3).reshape(B, G8 * 8, H, W)
    return x if channels is None else x[:, :channels].contiguous()


def pack_weight_h8_f32(w):
    """[Cout, Cin, KH, KW] fp32 -> fp32 in the plane order [Cin/16][KH*KW][2][CoutP][8] (Cin zero-padded to a multiple of 32): the source of
    the per-sample modulated planes (l2i_modulate_planes_h8)."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    cinp, coutp = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    full = torch.zeros(coutp, cinp, kh, kw, dtype=torch.float32, device=w.device)
    full[:cout, :cin] = w
    return full.reshape(coutp, cinp // 16, 2, 8, kh * kw).permute(1, 4, 2, 0, 3).contiguous()


def pack_weight_h8(w, cin_pad=32, dtype=None):
    """[Cout, Cin, KH, KW] fp32 -> 16-bit plane [Cin/16][KH*KW][2][CoutP][8] (int16 view) in the path's element type (or ``dtype``): the LDS image
    order of csrc/l2i_conv_h8.hip.  Cin is zero-padded to a multiple of ``cin_pad`` (32: the kernel's K chunk; 16 is enough for 3x3 stride-1 layers)."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    cinp = (cin + cin_pad - 1) // cin_pad * cin_pad
    if cinp != cin:
        w = torch.cat([w, w.new_zeros(cout, cinp - cin, kh, kw)], 1)
    if (dtype or h8_dtype()) == torch.bfloat16:
        return pack_weight_bf16x3(w)[0]
    coutp = (cout + 31) // 32 * 32
    full = torch.zeros(coutp, cinp, kh, kw, dtype=torch.float32, device=w.device)
    full[:cout] = w
    return full.to(torch.float16).reshape(coutp, cinp // 16, 2, 8, kh * kw).permute(1, 4, 2, 0, 3).contiguous().view(torch.int16)


def pack_weight_img_h8(w, dtype=None):
    """[Cout, Cin <= 4, K, K] fp32 -> the 16-bit planes of l2i_conv_img_h8 (include/l2i.h): [ceil(Cin K / 2)][1][2][CoutP][8] (int16 view), element
    (s, half, co, e) = w[co, c, ky, e] for (c, ky) = divmod(2 s + half, K), zero for e >= K and rows past Cin K — a 1x1 weight over the kernel's
    own contraction index k' = 16 s + 8 half + e, packed like every other h8 weight."""
    w = torch.as_tensor(w, dtype=torch.float32)
    cout, cin, k, _ = w.shape
    assert cin <= 4 and k <= 8
    ns = (cin * k + 1) // 2
    rows = torch.zeros(cout, 2 * ns, 8, dtype=torch.float32)
    rows[:, :cin * k, :k] = w.reshape(cout, cin * k, k)
    return pack_weight_h8(rows.reshape(cout, 16 * ns, 1, 1), cin_pad=16, dtype=dtype)


class ImgConvH8:
    """[r5] An image-side convolution of the 16-bit path (csrc/l2i_img_h8.hip): fp32 NCHW image in, h8 map out; forward only (the input gradients
    land on the fp32 image through the generic kernels: H8Conv.dgrad(out_f32=True), FrozenConv2d.dgrad_h8in)."""

    def __init__(self, weight, stride=1, padding=0, device='cuda'):
        w = torch.as_tensor(np.asarray(weight) if not torch.is_tensor(weight) else weight, dtype=torch.float32).cpu()
        self.cout, self.cin, self.k, _ = w.shape
        assert (self.k, stride) in ((1, 1), (3, 1), (7, 2)) and self.cout % 8 == 0, (self.k, stride, self.cout)
        self.stride, self.padding = stride, padding
        self.dtype = h8_dtype()
        self.planes = pack_weight_img_h8(w).to(device)

    def out_hw(self, h, w):
        return (h + 2 * self.padding - self.k) // self.stride + 1, (w + 2 * self.padding - self.k) // self.stride + 1

    def forward(self, x, bias=None, act=ACT_NONE, slope=0.2, gain=1.0, out_gain=1.0, sq=None):
        lib = _lib.load()
        B, cin, H, W = x.shape
        assert cin == self.cin and x.dtype == torch.float32
        oh, ow = self.out_hw(H, W)
        y = torch.empty(B, self.cout // 8, oh, ow, 8, device=x.device, dtype=self.dtype)
        p = ConvParams()
        p.x, p.w_hi, p.y = _lib.fptr(x), _lib.ptr(self.planes), _lib.ptr(y)
        p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = B, cin, H, W, self.cout, self.planes.shape[-2]
        p.KH = p.KW = self.k
        p.stride, p.pad_y, p.pad_x = self.stride, self.padding, self.padding
        p.OH, p.OW, p.OHf, p.OWf = oh, ow, oh, ow
        p.oy_step = p.ox_step = 1
        p.bias = _lib.fptr(bias)
        p.act, p.act_slope, p.act_gain, p.out_gain = act, slope, gain, out_gain
        if sq is not None:                                            # (reference like y, [SQ_SLOTS] zeroed accumulator, [fused flag])
            assert sq[0].shape == y.shape and sq[0].dtype == y.dtype and sq[1].numel() == _lib.SQ_SLOTS
            p.sq_ref, p.sq_out = _lib.ptr(sq[0]), _lib.fptr(sq[1])
            sq[2][0] = True
        name = 'l2i_conv_img_h8' + ('_f16' if self.dtype == torch.float16 else '')
        entry = getattr(lib, name)
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(entry(p, _lib.stream_ptr()), name)
            e1.record()
            PROFILE.append((e0, e1, 2.0 * B * self.cout * cin * self.k * self.k * oh * ow,
                            (B, cin, self.cout, self.k, self.k, self.stride, H, W, oh, ow, 1, False, False, ('b' if bias is not None else '') + str(act)), name, 'conv_h8'))
            return y
        _lib.check(entry(p, _lib.stream_ptr()), name)
        return y


class H8Conv:
    """A frozen convolution of the 16-bit path: ``weight`` [Cout, Cin, K, K] in correlation form (K = 1 or 3), stride 1 or 2, or the stride-2
    TRANSPOSED conv (``transposed=True``: y[co, 2i+k-pad] += x[ci, i] w[co, ci, k], as FrozenConv2d).  Forward and input-gradient both run on
    l2i_conv2d_h8 / l2i_conv_transpose2d_h8; no weight gradients (the walk is the only trainable tensor)."""

    def __init__(self, weight, stride=1, padding=0, transposed=False, device='cuda', cin_pad=32):
        w = torch.as_tensor(np.asarray(weight) if not torch.is_tensor(weight) else weight, dtype=torch.float32).cpu()
        self.cout, self.cin, self.k, _ = w.shape
        self.stride, self.padding, self.transposed, self.device = stride, padding, transposed, device
        assert cin_pad == 32 or (cin_pad == 16 and self.k == 3 and stride == 1 and not transposed), 'a 16-channel chunk exists for 3x3 stride-1 layers only'
        self.cinp, self.coutp_in = (self.cin + cin_pad - 1) // cin_pad * cin_pad, (self.cout + 31) // 32 * 32
        self.dtype = h8_dtype()                                       # element type of the weight planes: the maps it is called with must have it
        wt = w.transpose(0, 1).contiguous()
        self.fwd_planes = pack_weight_h8(w, cin_pad).to(device)
        if transposed or stride == 2:
            self.bwd_planes = pack_weight_h8(wt).to(device)          # transposed fwd: dx = corr_s2(gy, wt); stride-2 fwd: dx = transposed(gy, wt)
        else:
            self.bwd_planes = pack_weight_h8(torch.flip(wt, [2, 3])).to(device)

    def out_hw(self, h, w):
        if self.transposed:
            return (h - 1) * 2 - 2 * self.padding + self.k, (w - 1) * 2 - 2 * self.padding + self.k
        return (h + 2 * self.padding - self.k) // self.stride + 1, (w + 2 * self.padding - self.k) // self.stride + 1

    def forward(self, x, out=None, planes=None, w_bstride=0, out_hw=None, **kw):
        """x h8 [B, Cin/8, H, W, 8] -> h8 [B, Cout/8, OH, OW, 8] (or fp32 NCHW with out_f32=True)."""
        B, _, H, W, _ = x.shape
        oh, ow = out_hw if out_hw is not None else self.out_hw(H, W)
        out_f32 = kw.get('out_f32', False)
        if out is None:
            out = (torch.empty(B, self.cout, oh, ow, device=x.device, dtype=torch.float32) if out_f32
                   else torch.empty(B, (self.cout + 7) // 8, oh, ow, 8, device=x.device, dtype=x.dtype))
        return run_h8(planes if planes is not None else self.fwd_planes, x, out, self.cinp, self.cout, self.k, 2 if self.transposed else self.stride, self.padding,
                      transposed=self.transposed, w_bstride=w_bstride, **kw)

    def dgrad(self, gy, in_hw, out=None, planes=None, w_bstride=0, **kw):
        """Gradient w.r.t. the input of ``forward`` given the gradient w.r.t. its (pre-epilogue) output, both h8."""
        B = gy.shape[0]
        out_f32 = kw.get('out_f32', False)
        if out is None:
            out = (torch.empty(B, self.cin, in_hw[0], in_hw[1], device=gy.device, dtype=torch.float32) if out_f32
                   else torch.empty(B, (self.cin + 7) // 8, in_hw[0], in_hw[1], 8, device=gy.device, dtype=gy.dtype))
        pl = planes if planes is not None else self.bwd_planes
        if self.transposed:                                   # dx[ci, i] = sum gy[co, 2i + k - pad] w[co, ci, k]: a stride-2 correlation
            return run_h8(pl, gy, out, self.coutp_in, self.cin, self.k, 2, self.padding, transposed=False, w_bstride=w_bstride, **kw)
        if self.stride == 2:                                  # dx[ci, 2o + k - pad] += gy[co, o] w[co, ci, k]: the transposed conv
            return run_h8(pl, gy, out, self.coutp_in, self.cin, self.k, 2, self.padding, transposed=True, w_bstride=w_bstride, **kw)
        return run_h8(pl, gy, out, self.coutp_in, self.cin, self.k, 1, self.k - 1 - self.padding, transposed=False, w_bstride=w_bstride, **kw)


def _h8_dgrad_compact(self, gy, planes=None, **kw):
    """Input-gradient of a STRIDED 1x1 conv without the zero insertion: the 1x1 stride-1 conv of gy with the transposed weights, on the
    compact (output-resolution) map; the caller scatters it into every second pixel (kernels16.add_zero_insert)."""
    assert self.k == 1 and self.stride == 2 and not self.transposed
    B, _, oh, ow, _ = gy.shape
    out = torch.empty(B, (self.cin + 7) // 8, oh, ow, 8, device=gy.device, dtype=gy.dtype)
    return run_h8(planes if planes is not None else self.bwd_planes, gy, out, self.coutp_in, self.cin, 1, 1, 0, **kw)


H8Conv.dgrad_compact = _h8_dgrad_compact


def run_h8(planes, x, y, cin, cout, k, stride, pad, transposed=False, w_bstride=0, out_f32=False, out_scale=None, noise=None, noise_w=0.0, bias=None,
           residual=None, res_mask=None, out_mask=None, mask=(1.0, 0.0), act=ACT_NONE, slope=0.2, gain=1.0, out_gain=1.0, accumulate=False, res_sub=None, res_coef=1.0,
           res_coef_dev=None, sq=None, relu_in=False, rgb=None, mask_out=None, mask_bits=False, _defer=None):
    """Enqueue l2i_conv2d_h8 / l2i_conv_transpose2d_h8 on the current stream.  x: h8 bf16 with ``cin`` (multiple of 32) channels.
    ``_defer``: a list — the filled parameter struct is appended to it instead of being launched (``launch_pair_h8`` launches two of them as one kernel)."""
    lib = _lib.load()
    B, cg, H, W, _ = x.shape
    assert x.dtype in (torch.bfloat16, torch.float16) and cg * 8 == cin and cin % 16 == 0, (x.dtype, x.shape, cin)
    f16 = x.dtype == torch.float16
    coutp = planes.shape[-2]
    assert planes.shape[-5] == cin // 16 and planes.shape[-4] == k * k and coutp >= cout, (planes.shape, cin, k, cout)
    if out_f32:
        assert y.dtype == torch.float32 and y.shape[1] == cout
        OHf, OWf = y.shape[2], y.shape[3]
    else:
        assert y.dtype == x.dtype and y.shape[1] * 8 >= cout and cout % 8 == 0, (y.dtype, y.shape, cout)
        OHf, OWf = y.shape[2], y.shape[3]
    p = ConvParams()
    p.x, p.w_hi, p.y = _lib.ptr(x), _lib.ptr(planes), _lib.ptr(y)
    p.B, p.Cin, p.H, p.W, p.Cout, p.CoutP = B, cin, H, W, cout, coutp
    p.KH = p.KW = k
    p.stride, p.pad_y, p.pad_x = stride, pad, pad
    p.OHf, p.OWf = OHf, OWf
    if transposed:
        p.OH, p.OW = (OHf + 1) // 2, (OWf + 1) // 2
        p.oy_step = p.ox_step = 2
    else:
        p.OH, p.OW = min(OHf, (H + 2 * pad - k) // stride + 1), min(OWf, (W + 2 * pad - k) // stride + 1)
        p.oy_step = p.ox_step = 1
    p.out_scale, p.noise, p.noise_w, p.bias = _lib.fptr(out_scale), _lib.fptr(noise), float(noise_w), _lib.fptr(bias)
    for t in (residual, res_sub) + (() if mask_bits else (res_mask, out_mask)):
        assert t is None or (t.shape == y.shape and t.dtype == y.dtype)
    if mask_bits or mask_out is not None:             # [r6] sign planes (l2i.h: mask_out / mask_bits): one byte per 16-byte pixel slot, [B, Cout/8, OHf, OWf] uint8
        assert not out_f32
        for t in (mask_out,) + ((res_mask, out_mask) if mask_bits else ()):
            assert t is None or (t.dtype == torch.uint8 and tuple(t.shape) == tuple(y.shape[:4]) and t.is_contiguous()), (None if t is None else (t.dtype, t.shape), y.shape)
        p.mask_out, p.mask_bits = _lib.ptr(mask_out), int(bool(mask_bits))
    p.residual, p.res_mask, p.out_mask = _lib.ptr(residual), _lib.ptr(res_mask), _lib.ptr(out_mask)
    p.mask_pos, p.mask_neg = mask                                 # of the OUTPUT mask here: * (out_mask > 0 ? mask[0] : mask[1])
    if relu_in:                                                   # ReLU-on-load (VGG-19 reads pre-ReLU taps): in_mask == x, mask (1, 0)
        assert out_mask is None
        p.in_mask, p.mask_pos, p.mask_neg = p.x, 1.0, 0.0
    if res_sub is not None:
        p.res_sub, p.res_coef, p.res_coef_dev = _lib.ptr(res_sub), float(res_coef), _lib.fptr(res_coef_dev)
    p.act, p.act_slope, p.act_gain, p.out_gain = act, slope, gain, out_gain
    p.accumulate = int(accumulate)
    p.w_bstride, p.out_f32 = int(w_bstride), int(out_f32)
    if sq is not None:                                            # (reference like y, [SQ_SLOTS] zeroed accumulator, [fused flag])
        assert sq[0].shape == y.shape and sq[0].dtype == y.dtype and sq[1].numel() == _lib.SQ_SLOTS
        p.sq_ref, p.sq_out = _lib.ptr(sq[0]), _lib.fptr(sq[1])
        sq[2][0] = True
    if rgb is not None:                                           # [r5] (wmod [B, 3, Cout], bias [3], out [B, 3, OHf, OWf]) fp32: ToRGB of the output in the epilogue (l2i.h: rgb_w)
        assert not transposed and not out_f32 and cout in (32, 64) and tuple(rgb[0].shape) == (B, 3, cout) and tuple(rgb[2].shape) == (B, 3, OHf, OWf)
        p.rgb_w, p.rgb_bias, p.rgb_out = _lib.fptr(rgb[0]), _lib.fptr(rgb[1]), _lib.fptr(rgb[2])
    name = ('l2i_conv_transpose2d_h8' if transposed else 'l2i_conv2d_h8') + ('_f16' if f16 else '')
    if _defer is not None:
        assert not transposed
        _defer.append((p, f16, (planes, x, y, residual, res_mask, out_mask, mask_out, bias), ''.join(c for c, t in zip('brmo', (bias, residual, res_mask, out_mask)) if t is not None) + str(act)))
        return y
    entry = getattr(lib, name)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(entry(p, _lib.stream_ptr()), name)
        e1.record()
        npx = H * W if transposed else int(p.OH) * int(p.OW)
        PROFILE.append((e0, e1, 2.0 * B * cout * cin * k * k * npx, (B, cin, cout, k, k, stride, H, W, int(p.OH), int(p.OW), 2 if transposed else 1, False, False,
                                                                     ''.join(c for c, t in zip('dnbrmoas', (out_scale, noise, bias, residual, res_mask, out_mask, accumulate or None, res_sub)) if t is not None) + str(act)),
                        name, 'transposed_h8' if transposed else 'conv_h8'))
        return y
    _lib.check(entry(p, _lib.stream_ptr()), name)
    return y


PAIR_VARIANT = int(_os.environ.get('L2I_H8_PAIR_TILE', '-1'))      # -1: by map size (launch_pair_h8); 0 / 1: l2i_conv1x1_pair_h8's variant argument


PAIR_MAX_COUT2 = int(_os.environ.get('L2I_H8_PAIR_MAXC', '256'))     # (A/B: 128 leaves the 256-channel shapes — one block per CU — to separate launches)
PAIR_SHAPES = ((64, 64), (128, 128), (256, 256), (64, 128), (128, 256))          # (input channels of the first conv, output channels of the second) l2i_conv1x1_pair_h8 is built for


def pair_h8_shapes_ok(cin1, cout1, cout2, npix):
    """Does l2i_conv1x1_pair_h8 take two chained 1x1 stride-1 convs cin1 -> cout1 -> cout2 on maps of ``npix`` pixels (include/l2i.h lists the conditions)?"""
    return (cin1, cout2) in PAIR_SHAPES and cout2 <= PAIR_MAX_COUT2 and cout1 % 32 == 0 and npix % 128 == 0


CHAIN3_SHAPES = ((64, 64), (128, 128), (64, 128))        # (channels of the 3x3 conv, output channels of the last conv) l2i_conv_chain3_h8 is built for


def chain3_h8_shapes_ok(c, cout1, cout2, h, w):
    """Does l2i_conv_chain3_h8 take 3x3 (c -> c) -> 1x1 (c -> cout1) -> 1x1 (cout1 -> cout2) on h x w maps?"""
    return (c, cout2) in CHAIN3_SHAPES and cout2 <= PAIR_MAX_COUT2 and cout1 % 32 == 0 and w % 32 == 0 and h % 4 == 0


def launch_pair_h8(deferred, variant=None):
    """``deferred``: the structs ``run_h8(..., _defer=deferred)`` left — two (a 1x1 conv, then the 1x1 conv that reads its output: l2i_conv1x1_pair_h8) or three
    (a 3x3 stride-1 conv in front of them: l2i_conv_chain3_h8): ONE launch on the current stream.  Raises L2IError when the library refuses the chain (the caller
    checks ``pair_h8_shapes_ok`` / ``chain3_h8_shapes_ok`` first)."""
    lib = _lib.load()
    head = deferred[0] if len(deferred) == 3 else None
    (p1, f16, keep1, fl1), (p2, f16b, keep2, fl2) = deferred[-2:]
    assert f16 == f16b and (head is None or head[1] == f16)
    name = ('l2i_conv_chain3_h8' if head is not None else 'l2i_conv1x1_pair_h8') + ('_f16' if f16 else '')
    entry = getattr(lib, name)
    npix = int(p1.H) * int(p1.W)
    if variant is None:
        variant = PAIR_VARIANT if PAIR_VARIANT >= 0 else (1 if int(p1.B) * npix < 256 * 1024 else 0)      # 128-pixel tiles below 1024 blocks of 256 pixels
    args = ((head[0],) if head is not None else ()) + (p1, p2, int(variant), _lib.stream_ptr())
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(entry(*args), name)
        e1.record()
        B, c1, c2, c3 = int(p1.B), int(p1.Cin), int(p1.Cout), int(p2.Cout)
        flop = 2.0 * B * npix * (c1 * c2 + c2 * c3 + (9 * c1 * c1 if head is not None else 0))
        # (priced in the conv_h8 family: its launches are what this one replaces; bench.call_bytes knows the entry names)
        PROFILE.append((e0, e1, flop, (B, c1, c2, 1, 1, 1, int(p1.H), int(p1.W), int(p1.H), int(p1.W), 1, False, False, fl1 + '+' + fl2, c3), name, 'conv_h8'))
        return
    _lib.check(entry(*args), name)


PAIR_F32_SHAPES = ((64, 64), (64, 128), (128, 128))       # (input channels of the first conv, output channels of the second) l2i_conv1x1_pair_f32 is built for


def pair_f32_shapes_ok(cin1, cout1, cout2, npix):
    return (cin1, cout2) in PAIR_F32_SHAPES and cout1 % 32 == 0 and npix % 256 == 0


def launch_pair_f32(deferred):
    """``deferred``: the two structs ``run_launch(..., _defer=deferred)`` left (a 1x1 conv with bias / residual / ReLU, then the 1x1 conv that reads its output):
    ONE launch of l2i_conv1x1_pair_f32 on the current stream."""
    lib = _lib.load()
    (p1, keep1), (p2, keep2) = deferred
    name = 'l2i_conv1x1_pair_f32'
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.l2i_conv1x1_pair_f32(p1, p2, _lib.stream_ptr()), name)
        e1.record()
        B, c1, c2, c3, npix = int(p1.B), int(p1.Cin), int(p1.Cout), int(p2.Cout), int(p1.H) * int(p1.W)
        PROFILE.append((e0, e1, 2.0 * B * npix * (c1 * c2 + c2 * c3), (B, c1, c2, 1, 1, 1, int(p1.H), int(p1.W), int(p1.H), int(p1.W), 1, False, False, 'br1+b1', c3), name, 'gemm1x1_f32'))
        return
    _lib.check(lib.l2i_conv1x1_pair_f32(p1, p2, _lib.stream_ptr()), name)
