"""The frozen networks of the walk-training step on the 16-bit path (BASELINE config 5: "fp16 storage / MFMA with fp32 accumulate"; bf16 here —
fp32's exponent range, so activations of 1e2 and gradients of 1e-9 need no scaling): every feature map and every saved tensor is a bf16 tensor
in the channel-blocked h8 layout [B, C/8, H, W, 8]; every contraction is one v_mfma_f32_32x32x16_bf16 product per MAC with fp32 accumulation
(csrc/l2i_conv_h8.hip); the streaming kernels read and write h8 (csrc/l2i_stream_h8.hip).  fp32 stays where the reference's numbers live:
images (3 channels), latents, styles / demodulation, noise, biases, every reduction and loss sum, the walk.

Same classes, call signatures and autograd contracts as the fp32 modules (generator.py, discriminator.py, regressor.py, perceptual.py) — images
in, images / predictions / losses out — so graph.py switches between the two by ``conv.PRECISION`` ('bf16' = this module).  Structural
differences from the fp32 path, all forced by "no prologue in the conv kernel" (its operands go global -> LDS by DMA):
  * style modulation: the style is folded into one weight-plane set per sample (the reference's own formulation, networks.py:234-243) by
    l2i_modulate_planes_h8; demodulation is the conv's out_scale;
  * activation-gradient masks ride on the PRODUCING epilogue (out_mask with leaky values), on the FIR's epilogue, or on one masked copy where
    the gradient also feeds an unmasked branch;
  * VGG-19 reads pre-ReLU taps: ReLU-on-load inside the conv (four packed integer max per fragment).
Reference call sites as in the fp32 modules."""
import math
import os

import numpy as np
import torch

from . import conv as C
from . import kernels as K
from . import kernels16 as K16
from . import specs
from .discriminator import Discriminator as _Discriminator32, _ConvLReLUFn, _eq_conv, _vec
from .generator import Generator as _Generator32, _ModPlan, _Mod, _ToRGB, _draw_noise, _noise_for, _t
from .perceptual import VGG_STD, VGG_MEAN
from .regressor import _CB, _fold_bn
from .specs import RESNET50_LAYERS

SQRT2 = math.sqrt(2.0)
LRELU_MASK = (SQRT2, 0.2 * SQRT2)

# [r5] IEEE fp16 elements (conv.PRECISION = 'f16'; BASELINE configs[4]: "fp16 MFMA") have 3 more mantissa bits than bf16 and a range of 6e-8 .. 65504:
# the forward maps of the four networks fit it as they are (demodulated generator activations, folded-BatchNorm ResNet, VGG taps of [-1,1] images),
# the GRADIENT maps do not (a ContentLoss gradient at 1024^2 is ~1e-11 per element).  Each loss branch therefore multiplies the gradient it receives by
# a power of two on the way in and its fp32 result (the 3-channel image gradient; the latent gradient of the generator) by the inverse on the way
# out: exact in every fp32 quantity, and inside a branch all maps are linear in the incoming gradient.  The exponents are STATIC per (branch,
# resolution, per-GPU batch): `loss_scale_for` below — measured with tools/bf16_study.py --probe (max / median magnitude of every gradient map), chosen
# to put the largest map of a branch near 2^5.  [r6] On top of them ONE dynamic power of two with torch GradScaler's semantics, kept on the device
# (optim.LossScaler / optim.GuardedAdam, csrc/l2i_optim.hip): a step whose walk gradient holds an inf / NaN is skipped and the factor halves, it doubles
# again after `constants.LOSS_SCALE_GROWTH_INTERVAL` clean steps.  bf16 elements: every scale is 1 (not applied).
# [r5] the three image-side convs (VGG conv1_1, the discriminator's from-RGB 1x1, ResNet-50's 7x7 stem) read the fp32 3-channel image and STORE h8
# (l2i_conv_img_h8, csrc/l2i_img_h8.hip), the stem's input gradient READS h8 (l2i_conv_params::in_h8): no padded 16 / 32-channel 16-bit copy of the
# image, no fp32 stem map, no cast passes (-6 GB of the step's HBM traffic at 1024^2 batch 8).  L2I_H8_IMG_CONVS=0: the round-4 form (A/B).
IMG_CONVS = os.environ.get('L2I_H8_IMG_CONVS', '1') != '0'
# [r5] ToRGB of the 512^2 / 1024^2 StyledConv outputs (64 / 32 channels: one block of the conv holds them all) in that conv's epilogue
# (l2i_conv_params::rgb_w) instead of a pass that reads the feature map again.  L2I_H8_RGB_FUSED=0: the separate l2i_torgb_fwd_h8 launch (A/B).
RGB_FUSED = os.environ.get('L2I_H8_RGB_FUSED', '1') != '0'
# [r6] ResNet-50's backward reads one-bit sign planes written by the forward convs instead of the activation maps themselves (l2i.h: mask_out / mask_bits).
# L2I_H8_SIGN_PLANES=0: the maps, as in round 5 (A/B).
SIGN_PLANES = os.environ.get('L2I_H8_SIGN_PLANES', '1') != '0'
CHAIN3 = os.environ.get('L2I_H8_CHAIN3', '1') != '0'       # [r6] ... and the 3x3 conv in front of such a pair in the same launch (l2i_conv_chain3_h8)
PAIR = os.environ.get('L2I_H8_PAIR', '1') != '0'          # [r6] ResNet-50's trunk: chained 1x1 convs as one launch (csrc/l2i_pair_h8.hip); 0: separate launches (A/B)
# [r5] the per-sample weight planes of all modulated convs of a pass in ONE launch (kernels16.ModulatePlan) instead of one 15 us launch per layer and
# pass (51 per step).  L2I_H8_MOD_MULTI=0: per layer (A/B).
MOD_MULTI = os.environ.get('L2I_H8_MOD_MULTI', '1') != '0'

PROBE = None            # tools/bf16_study.py --probe: a list that receives (tag, shape, max |g|, median |g| of the non-zero entries) per gradient map


def _probe(tag, t):
    if PROBE is not None:
        a = t.detach().float().abs().reshape(-1)
        nz = a[a > 0]
        PROBE.append((tag, tuple(t.shape), float(a.max()), float(nz.median()) if nz.numel() else 0.0, float((a == 0).float().mean())))


def loss_scale_for(resolution, batch):
    """log2 of the per-branch STATIC gradient scales of the fp16 path at `resolution`^2 and PER-GPU `batch` (each rank's losses are means over its own
    shard, dist.shard: its gradients scale with 1 / B_local, not with the global batch).  The magnitudes follow the loss normalisations: the
    ContentLoss is a mean over B * C * H * W elements (gradient ~ 1 / (B H W)), the regressor's BCE a mean over B * attrs through an average pool
    (1 / B), the discriminator's BCE a mean over B through learned-scale convs, and the generator receives their sum.  An override for experiments
    and for the overflow-guard tests: L2I_F16_SCALES="R,V,D,G" (log2 values)."""
    env = os.environ.get('L2I_F16_SCALES')
    if env:
        r, v, d, g = (int(x) for x in env.split(','))
        return dict(R=r, V=v, D=d, G=g)
    lb = int(round(math.log2(max(batch, 1))))
    lr = int(round(math.log2(resolution)))
    return dict(R=F16_SCALE_BASE['R'] + lb + F16_SCALE_RES['R'] * (lr - 8), V=F16_SCALE_BASE['V'] + lb + F16_SCALE_RES['V'] * (lr - 8),
                D=F16_SCALE_BASE['D'] + lb + F16_SCALE_RES['D'] * (lr - 8), G=F16_SCALE_BASE['G'] + lb + F16_SCALE_RES['G'] * (lr - 8))


# exponents at 256^2, batch 1, and their growth per doubling of the resolution (tools/bf16_study.py --probe, profiles/r05_fp16_gradient_ranges.txt)
# measured largest map of a branch (log2, one attribute; five attributes ~1 lower): R -(9 + lb + 2 (lr - 8)), VGG content -(22 + lb + 2 (lr - 8)) (up to
# 1.7 above that at 1024^2 batch 8), D -(11 .. 14.5) at its 8^2 end falling 1.75 octaves per block towards the image (14 octaves at 1024^2: the
# discriminator's backward therefore also doubles the gradient once per block, F16_D_BLOCK_GAIN), G -(8 .. 10.4) at its 4^2 end falling 10 octaves
# towards the image; lb = log2(batch), lr = log2(resolution).  The scales put the largest map of each branch near 2^5: eleven octaves below fp16's
# 2^16 (the first choice, 2^8 .. 2^11, went non-finite after ~100 bench steps: the magnitudes move by a few octaves with the alpha draw and the
# sample), with the MEDIAN of the smallest maps (G at the image end, 1024^2 batch 8) at 2^-10.5: normal numbers (>= 2^-14) throughout.
F16_SCALE_BASE = dict(R=14, V=26, D=16, G=12)
F16_SCALE_RES = dict(R=2, V=2, D=0, G=0)
F16_D_BLOCK_GAIN = 2.0


# [r6] The scales live on the NETWORK objects (`net.scaler`, an optim.LossScaler shared by the four networks of one graph; None = unscaled: bf16
# elements, or an fp16 network driven directly by a test) — round 5 kept them in a module-global dict that every load_networks() call overwrote, so two
# live graphs of different resolution or batch shared whichever was built last.  On top of the static exponents the scaler carries ONE dynamic power
# of two on the device (GradScaler semantics without a host read: optim.py, csrc/l2i_optim.hip): every branch multiplies its incoming gradient by
# static * dynamic, the generator's latent gradient leaves multiplied by 1 / (static_G * dynamic).
def attach_scaler(nets, scaler):
    for n in nets:
        if n is not None:
            n.scaler = scaler
    return scaler


def _gs(net, key):
    """The branch's STATIC gradient scale (a float power of two; 1.0 without a scaler: bf16 elements / stand-alone networks)."""
    sc = getattr(net, 'scaler', None)
    if sc is None or net.dtype != torch.float16:
        return 1.0
    return sc.static(key)


def _dyn(net, inverse=False):
    """The dynamic factor (or its inverse) as a one-element device tensor, or None."""
    sc = getattr(net, 'scaler', None)
    if sc is None or net.dtype != torch.float16:
        return None
    return sc.inv_dyn if inverse else sc.dyn


# =====================================================================================================================================
# VGG-19 prefix content loss (perceptual.py)
# =====================================================================================================================================
class VGG19Prefix:
    def __init__(self, state, device='cuda'):
        P = state
        self.device = device
        self.dtype = K16.h8_dtype()
        w0 = torch.as_tensor(np.asarray(P['0.weight']), dtype=torch.float32)
        w0 = w0 / torch.tensor(VGG_STD, dtype=torch.float32).reshape(1, 3, 1, 1)
        ws = [w0] + [torch.as_tensor(np.asarray(P['%d.weight' % i]), dtype=torch.float32) for i in (2, 5, 7)]
        self.convs = [C.H8Conv(w, 1, 1, device=device, cin_pad=16 if i == 0 else 32) for i, w in enumerate(ws)]      # conv1_1: three real channels of ONE 16-channel chunk
        self.biases = [torch.as_tensor(np.asarray(P['%d.bias' % i]), dtype=torch.float32).contiguous().to(device) for i in (0, 2, 5, 7)]
        self.conv0_img = C.ImgConvH8(w0, 1, 1, device=device) if IMG_CONVS else None      # conv1_1 forward on the fp32 image, h8 out (its gradient: convs[0].dgrad)
        self.neg_mean = (-torch.tensor(VGG_MEAN, dtype=torch.float32)).to(device)

    def taps(self, img, org=None):
        """[B,3,H,W] fp32 -> (c1, c2, p, c3, c4, pool_idx) h8: pre-ReLU conv outputs conv_1..conv_4, p = relu(maxpool(c2)).  ``org`` = the four
        taps of the original image: a seventh element, the four sums of (c_k - org_k)^2, formed in the conv epilogues."""
        xc = K.fused_bias_act(img.contiguous(), self.neg_mean, None, 1, 0, 0.0, 1.0)        # x - mean (fp32: the padding stays exactly zero)
        sq = [None] * 4
        if org is not None:
            acc = torch.zeros(4, C._lib.SQ_SLOTS, device=img.device, dtype=torch.float32)
            sq = [(org[k], acc[k], [False]) for k in range(4)]
        if self.conv0_img is not None:
            c1 = self.conv0_img.forward(xc, bias=self.biases[0], sq=sq[0])
        else:
            xh = K16.cast_to_h8(xc, 16, dtype=self.dtype)          # three real channels of a 16-channel chunk
            c1 = self.convs[0].forward(xh, bias=self.biases[0], sq=sq[0])
        c2 = self.convs[1].forward(c1, relu_in=True, bias=self.biases[1], sq=sq[1])
        p, idx = K16.maxpool2d_fwd(c2, 2, 2, 0, relu=True)          # relu(maxpool(.)) == maxpool(relu(.)); the stored map is already rectified
        c3 = self.convs[2].forward(p, bias=self.biases[2], sq=sq[2])
        c4 = self.convs[3].forward(c3, relu_in=True, bias=self.biases[3], sq=sq[3])
        if PROBE is not None:
            for tag, t in (('V.fwd.c1', c1), ('V.fwd.c2', c2), ('V.fwd.c3', c3), ('V.fwd.c4', c4)):
                _probe(tag, t)
        if org is None:
            return c1, c2, p, c3, c4, idx
        return c1, c2, p, c3, c4, idx, [acc[k].sum().reshape(1) for k in range(4)]

    def org_taps(self, org):
        with torch.no_grad():
            o1, o2, _, o3, o4, _ = self.taps(org.detach())
        return (o1, o2, o3, o4)

    def content_losses(self, org, shifted, org_taps=None):
        return _Content16Fn.apply(shifted, self, org_taps if org_taps is not None else self.org_taps(org))


class _Content16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, net, org_taps):
        c1, c2, p, c3, c4, idx, sums = net.taps(img.detach(), org=org_taps)
        mine = (c1, c2, c3, c4)
        losses = torch.cat([s / float(b.numel()) for s, b in zip(sums, mine)])
        if img.requires_grad:
            ctx.net, ctx.org, ctx.acts, ctx.in_hw = net, org_taps, (c1, c2, p, c3, c4, idx), (img.shape[2], img.shape[3])
        return losses

    @staticmethod
    def backward(ctx, g_losses):
        net, (o1, o2, o3, o4) = ctx.net, ctx.org
        c1, c2, p, c3, c4, idx = ctx.acts
        S, dyn = _gs(net, 'V'), _dyn(net)
        gs = g_losses if dyn is None else g_losses * (dyn * S)           # static * dynamic scale (fp16 elements; exact: powers of two)
        gl = [gs[k:k + 1].contiguous() for k in range(4)]
        hw = lambda t: (t.shape[2], t.shape[3])
        d4 = K16.sqdiff(o4, c4, coef=2.0 / c4.numel(), coef_dev=gl[3], want_grad=True, want_sum=False)[1]
        d4_probe = d4 if PROBE is not None else None
        g3 = net.convs[3].dgrad(d4, hw(c3), out_mask=c3, residual=c3, res_sub=o3, res_coef=2.0 / c3.numel(), res_coef_dev=gl[2])
        del d4
        _probe('V.g3', g3)
        gp = net.convs[2].dgrad(g3, hw(p), out_mask=p)
        del g3
        g2 = K16.maxpool2d_bwd(gp, idx, hw(c2), 2, 2, 0, a=o2, b=c2, coef=2.0 / c2.numel(), coef_dev=gl[1])
        del gp
        _probe('V.g2', g2)
        g1 = net.convs[1].dgrad(g2, hw(c1), out_mask=c1, residual=c1, res_sub=o1, res_coef=2.0 / c1.numel(), res_coef_dev=gl[0])
        del g2
        for tag, t in (('V.d4', d4_probe), ('V.g1', g1)) if PROBE is not None else ():
            _probe(tag, t)
        g_img = net.convs[0].dgrad(g1, ctx.in_hw, out_f32=True, out_gain=1.0 / S)
        ctx.acts = ctx.org = None
        return g_img, None, None


# =====================================================================================================================================
# ResNet-50 regressor (regressor.py)
# =====================================================================================================================================
class _CB16:
    def __init__(self, P, conv_name, bn_name, stride, padding, device):
        w, b = _fold_bn(P, conv_name, bn_name)
        self.conv = C.H8Conv(w, stride=stride, padding=padding, device=device)
        self.bias = b.contiguous().to(device)


class ResNet50:
    def __init__(self, state, device='cuda'):
        P = state
        self.device = device
        self.dtype = K16.h8_dtype()
        self.stem = _CB(P, 'conv1', 'bn1', 2, 3, device)            # 7x7 stride 2 on the 3-channel fp32 image: fp32 kernels (L2I_H8_IMG_CONVS=0) and the input gradient
        self.stem_img = C.ImgConvH8(_fold_bn(P, 'conv1', 'bn1')[0], 2, 3, device=device) if IMG_CONVS else None
        if IMG_CONVS:       # the input gradient contracts with the weights the forward multiplied by: the folded weights rounded to the element type
            self.stem_bwd = C.FrozenConv2d(_fold_bn(P, 'conv1', 'bn1')[0].to(self.dtype).float(), 2, 3, device=device)
        self.blocks = []
        for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
            for b in range(blocks):
                p = 'layer%d.%d' % (li + 1, b)
                s = stride if b == 0 else 1
                self.blocks.append(dict(c1=_CB16(P, p + '.conv1', p + '.bn1', 1, 0, device), c2=_CB16(P, p + '.conv2', p + '.bn2', s, 1, device),
                                        c3=_CB16(P, p + '.conv3', p + '.bn3', 1, 0, device),
                                        down=_CB16(P, p + '.downsample.0', p + '.downsample.1', s, 0, device) if b == 0 else None))
        self.fc_w = torch.as_tensor(np.asarray(P['fc.weight']), dtype=torch.float32).contiguous().to(device)
        self.fc_b = torch.as_tensor(np.asarray(P['fc.bias']), dtype=torch.float32).contiguous().to(device)

    def features(self, img):
        return _ResNet16Fn.apply(img, self)

    def __call__(self, img):
        return torch.addmm(self.fc_b, self.features(img), self.fc_w.t())


def _NO_MAP(like):
    """Stand-in for the output of a conv whose map is never written (the 3x3 head of l2i_conv_chain3_h8): carries shape and dtype for the parameter struct,
    owns no new memory (the deferred launch clears its pointer)."""
    return like


class _ResNet16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, net):
        keep = img.requires_grad
        x = img.detach().contiguous()
        if IMG_CONVS:
            a0 = net.stem_img.forward(x, bias=net.stem.bias, act=C.ACT_RELU)            # h8 [B,8,H/2,W/2,8] from the fp32 image (csrc/l2i_img_h8.hip)
            p0, idx0 = K16.maxpool2d_fwd(a0, 3, 2, 1)
        else:
            a0 = net.stem.conv.forward(x, bias=net.stem.bias, act=C.ACT_RELU)          # fp32 [B,64,H/2,W/2]
            p0, idx0 = K16.maxpool2d_fwd(K16.cast_to_h8(a0, dtype=net.dtype), 3, 2, 1)
        saved = dict(in_hw=(x.shape[2], x.shape[3]), a0=a0 if keep else None, idx0=idx0 if keep else None, blocks=[])
        cur = p0
        # [r6] SIGN_PLANES: the backward needs ONE BIT of every activation of this net (is the ReLU output positive); the three convs of a bottleneck write it
        # beside their map (l2i.h: mask_out, one byte per 16-byte pixel slot) and the gradient launches read the byte planes (mask_bits) — 1/16 of the
        # bytes of the maps they stand for, and y1 / y2 / out need not be kept for the backward at all.  L2I_H8_SIGN_PLANES=0: the round-5 form (A/B).
        bits = keep and SIGN_PLANES
        plane = lambda t: torch.empty(t.shape[:4], device=t.device, dtype=torch.uint8)
        n_blk = len(net.blocks)
        # [r6] PAIR: conv3 + identity + ReLU of a block and conv1 + ReLU of the NEXT block as one launch (l2i_conv1x1_pair_h8: the wide map goes from the first
        # conv's epilogue to the second conv's MFMAs in registers, written once, never read back) where the library has the shape; bit-identical
        ahead = None                                       # (y1, its sign plane) of this block when the previous block's pair launch already produced it
        for bi, blk in enumerate(net.blocks):
            hw_in = (cur.shape[2], cur.shape[3])
            s2 = blk['c2'].conv.stride
            hw_out = (hw_in[0] // s2, hw_in[1] // s2)
            if bits:
                m1 = ahead[1] if ahead is not None else torch.empty(cur.shape[0], blk['c1'].conv.cout // 8, hw_in[0], hw_in[1], device=cur.device, dtype=torch.uint8)
                m2 = torch.empty(cur.shape[0], blk['c2'].conv.cout // 8, hw_out[0], hw_out[1], device=cur.device, dtype=torch.uint8)
                mo = torch.empty(cur.shape[0], blk['c3'].conv.cout // 8, hw_out[0], hw_out[1], device=cur.device, dtype=torch.uint8)
            else:
                m1 = m2 = mo = None
            y1 = ahead[0] if ahead is not None else blk['c1'].conv.forward(cur, bias=blk['c1'].bias, act=C.ACT_RELU, mask_out=m1)
            idt = blk['down'].conv.forward(cur, bias=blk['down'].bias) if blk['down'] is not None else cur
            nxt = net.blocks[bi + 1] if bi + 1 < n_blk else None
            ahead = None
            c2, c3 = blk['c2'].conv, blk['c3'].conv
            pair = PAIR and nxt is not None and C.pair_h8_shapes_ok(c3.cinp, c3.cout, nxt['c1'].conv.cout, hw_out[0] * hw_out[1])
            # [r6] CHAIN3: conv2 joins the launch too where it is a stride-1 3x3 on a shape the library has (l2i_conv_chain3_h8): one launch per bottleneck, y2 never
            # written (its sign plane is, for the backward)
            chain = pair and CHAIN3 and c2.stride == 1 and (bits or not keep) and C.chain3_h8_shapes_ok(c2.cinp, c3.cout, nxt['c1'].conv.cout, hw_out[0], hw_out[1])
            d = []
            y2 = c2.forward(y1, bias=blk['c2'].bias, act=C.ACT_RELU, mask_out=m2, **(dict(_defer=d, out=_NO_MAP(y1)) if chain else {}))
            if pair:
                m1n = torch.empty(cur.shape[0], nxt['c1'].conv.cout // 8, hw_out[0], hw_out[1], device=cur.device, dtype=torch.uint8) if bits else None
                if not chain:
                    d = []
                out = c3.forward(y2, bias=blk['c3'].bias, residual=idt, act=C.ACT_RELU, mask_out=mo, _defer=d)
                y1n = nxt['c1'].conv.forward(out, bias=nxt['c1'].bias, act=C.ACT_RELU, mask_out=m1n, _defer=d)
                if chain:
                    d[0][0].y = None                       # (the 3x3 conv's map is not written: `y2` above only carries its shape)
                C.launch_pair_h8(d)
                ahead = (y1n, m1n)
            else:
                out = c3.forward(y2, bias=blk['c3'].bias, residual=idt, act=C.ACT_RELU, mask_out=mo)
            if bits:
                # (kept as maps: the input of a stride-2 block — its mask rides on the zero-insertion pass — and the last output, for the first mask)
                keep_cur = blk['down'] is not None and blk['down'].conv.stride == 2
                saved['blocks'].append((cur if keep_cur else cur.shape, m1, m2, mo, out if bi == n_blk - 1 else None))
            elif keep:
                saved['blocks'].append((cur, y1, y2, out))
            cur = out
            if PROBE is not None and blk['down'] is not None:
                _probe('R.fwd.out', out)
        b, g8, h, w, _ = cur.shape
        feat = K16.dot_reduce(cur) * (1.0 / (h * w))                                   # adaptive avg-pool (1,1), fp32 sums
        ctx.net, ctx.saved, ctx.last = net, saved if keep else None, cur if keep else None
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        net, saved, last = ctx.net, ctx.saved, ctx.last
        if saved is None:
            raise RuntimeError('regressor was run without a differentiable input')
        b, g8, h, w, _ = last.shape
        hw = lambda t: (t.shape[2], t.shape[3])
        # The residual-trunk gradient G stays bf16 (h8) from block to block.  [r4] Carrying it in fp32 through a stage (the round-3 review's
        # hypothesis: sixteen successive bf16 roundings cost the gradient its direction) was built — fp32 residual / output operands in the conv
        # epilogue — and MEASURED: input-gradient cosine against the exact oracle 0.9715 / 0.9526 at 64^2 / 256^2 with either trunk, identical to four
        # digits (profiles/r04_bf16_trunk_f32_vs_bf16.txt; the cause is the forward's storage rounding), while its two extra
        # epilogue branches cost the 16-bit conv kernel 3 - 17 % on launches with a residual (c5 209.5 -> 213.1 images/s without them): removed.
        S, dyn = _gs(net, 'R'), _dyn(net)
        if dyn is not None:
            g_feat = g_feat * dyn
        g = (g_feat * (S / (h * w))).reshape(b, g8, 1, 1, 8).expand(b, g8, h, w, 8).contiguous().to(net.dtype)
        G = K16.mask_mul(g, last)                                                        # gradient w.r.t. the pre-ReLU sum of the last block
        n = len(net.blocks)
        bits = len(saved['blocks'][0]) == 5                # sign planes (forward: SIGN_PLANES)
        ahead = None                                       # g_y2 of this block when the pair launch of the block above already produced it
        for bi in range(n - 1, -1, -1):
            blk = net.blocks[bi]
            if bits:
                cur, y1, y2, out, _ = saved['blocks'][bi]                      # y1 / y2 / out: byte planes; cur: the input map (stride-2 blocks) or its shape
                cur_hw = (cur[2], cur[3]) if isinstance(cur, torch.Size) else hw(cur)
                m = saved['blocks'][bi - 1][3] if bi > 0 else None
                mb = dict(mask_bits=True)
            else:
                cur, y1, y2, out = saved['blocks'][bi]
                cur_hw = hw(cur)
                m = cur if bi > 0 else None                # the block input is the previous block's ReLU output (the pooled stem map is not)
                mb = {}
            mm = dict(out_mask=m, res_mask=m, **mb) if m is not None else {}
            g_y2 = ahead if ahead is not None else blk['c3'].conv.dgrad(G, hw(y2), out_mask=y2, **mb)
            ahead = None
            prev = net.blocks[bi - 1] if bi > 0 else None
            # [r6] PAIR: conv1's input gradient + trunk gradient, masked, and conv3's input gradient of the block below in one launch; CHAIN3: conv2's input gradient in front
            pair = (PAIR and bits and m is not None and blk['down'] is None
                    and C.pair_h8_shapes_ok(blk['c1'].conv.coutp_in, blk['c1'].conv.cin, prev['c3'].conv.cin, cur_hw[0] * cur_hw[1]))
            chain = pair and CHAIN3 and blk['c2'].conv.stride == 1 and C.chain3_h8_shapes_ok(blk['c2'].conv.coutp_in, blk['c1'].conv.cin, prev['c3'].conv.cin, cur_hw[0], cur_hw[1])
            d = []
            g_y1 = blk['c2'].conv.dgrad(g_y2, hw(y1), out_mask=y1, **mb, **(dict(_defer=d, out=_NO_MAP(g_y2)) if chain else {}))
            if not chain:
                del g_y2
            if blk['down'] is None:
                if pair:
                    y2p = saved['blocks'][bi - 1][2]
                    if not chain:
                        d = []
                    Gp = blk['c1'].conv.dgrad(g_y1, cur_hw, residual=G, _defer=d, **mm)
                    ahead = prev['c3'].conv.dgrad(Gp, hw(y2p), out_mask=y2p, _defer=d, **mb)
                    if chain:
                        d[0][0].y = None
                    C.launch_pair_h8(d)
                else:
                    Gp = blk['c1'].conv.dgrad(g_y1, cur_hw, residual=G, **mm)
            elif blk['down'].conv.stride == 1:
                t = blk['c1'].conv.dgrad(g_y1, cur_hw)
                Gp = blk['down'].conv.dgrad(G, cur_hw, residual=t, **mm)
                del t
            else:
                Gp = blk['c1'].conv.dgrad(g_y1, cur_hw, **(dict(out_mask=m, **mb) if m is not None else {}))
                K16.add_zero_insert(Gp, blk['down'].conv.dgrad_compact(G), mask=cur if bi > 0 else None)       # strided 1x1: compact 1x1 conv + zero insertion
            del g_y1
            G = Gp
            if PROBE is not None and (bi in (0, n - 1) or net.blocks[bi]['down'] is not None):
                _probe('R.G%d' % bi, G)
        a0 = saved['a0']
        if a0.dim() == 5:                                  # [r5] h8 stem map: the 7x7 gradient kernel reads the 16-bit pool gradient and mask
            g_a0 = K16.maxpool2d_bwd(G, saved['idx0'], (a0.shape[2], a0.shape[3]), 3, 2, 1)
            g_img = net.stem_bwd.dgrad_h8in(g_a0, saved['in_hw'], in_mask=a0, mask=(1.0, 0.0), out_gain=1.0 / S)
        else:
            g_a0 = K16.cast_from_h8(K16.maxpool2d_bwd(G, saved['idx0'], (a0.shape[2], a0.shape[3]), 3, 2, 1), a0.shape[1])
            g_img = net.stem.conv.dgrad(g_a0, saved['in_hw'], in_mask=a0, mask=(1.0, 0.0), out_gain=1.0 / S)
        ctx.saved = ctx.last = None
        return g_img, None


# =====================================================================================================================================
# StyleGAN2 discriminator (discriminator.py)
# =====================================================================================================================================
def _eq16(P, name, stride, padding, device):
    w = torch.as_tensor(np.asarray(P[name]), dtype=torch.float32)
    w = w * (1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3]))
    return C.H8Conv(w, stride=stride, padding=padding, device=device)


class Discriminator(_Discriminator32):
    """Body on the 16-bit path; the 4x4 tail (minibatch stddev, final conv, two linears: [B,513,4,4] and smaller) is the fp32 code of the base class."""

    def __init__(self, state, size, device='cuda'):
        P = state
        self.size, self.device = size, device
        self.dtype = K16.h8_dtype()
        log_size = int(math.log2(size))
        self.conv0 = _eq16(P, 'convs.0.0.weight', 1, 0, device)
        w0 = torch.as_tensor(np.asarray(P['convs.0.0.weight']), dtype=torch.float32)
        self.conv0_img = C.ImgConvH8(w0 * (1.0 / math.sqrt(w0.shape[1] * w0.shape[2] * w0.shape[3])), 1, 0, device=device) if IMG_CONVS else None
        self.bias0 = _vec(P, 'convs.0.1.bias', device)
        self.blocks = []
        for n in range(1, log_size - 1):
            p = 'convs.%d' % n
            blk = dict(c1=_eq16(P, p + '.conv1.0.weight', 1, 1, device), b1=_vec(P, p + '.conv1.1.bias', device),
                       c2=_eq16(P, p + '.conv2.1.weight', 2, 0, device), b2=_vec(P, p + '.conv2.2.bias', device),
                       sk1=_eq16(P, p + '.skip.1.weight', 1, 0, device), k=_vec(P, p + '.conv2.0.kernel', device))
            blk['kf'] = torch.flip(blk['k'], [0, 1]).contiguous()
            blk['ksep'], blk['kfsep'] = K16.separable(np.asarray(P[p + '.conv2.0.kernel'])), K16.separable(np.asarray(P[p + '.conv2.0.kernel'])[::-1, ::-1])
            self.blocks.append(blk)
        self.final_conv = _eq_conv(P, 'final_conv.0.weight', 1, 1, device)
        self.final_bias = _vec(P, 'final_conv.1.bias', device)
        w = _vec(P, 'final_linear.0.weight', device)
        self.lin0_wt = (w * (1.0 / math.sqrt(w.shape[1]))).t().contiguous()
        self.lin0_b = _vec(P, 'final_linear.0.bias', device)
        w = _vec(P, 'final_linear.1.weight', device)
        self.lin1_wt = (w * (1.0 / math.sqrt(w.shape[1]))).t().contiguous()
        self.lin1_b = _vec(P, 'final_linear.1.bias', device)

    def body(self, img):
        return _DBody16Fn.apply(img, self)


class _DBody16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, net):
        keep = img.requires_grad
        x = img.detach().contiguous()
        lr = dict(act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
        if net.conv0_img is not None:
            y0 = net.conv0_img.forward(x, bias=net.bias0, **lr)                            # from-RGB 1x1 on the fp32 image, h8 out
        else:
            y0 = net.conv0.forward(K16.cast_to_h8(x, 32, dtype=net.dtype), bias=net.bias0, **lr)
        saved = [y0]
        cur = y0
        bits = keep and SIGN_PLANES                    # [r6] the backward needs the SIGNS of y1 / y2 only: the convs write their sign planes, the maps are not kept
        for blk in net.blocks:
            h = cur.shape[2]
            m1 = torch.empty(cur.shape[0], blk['c1'].cout // 8, h, cur.shape[3], device=cur.device, dtype=torch.uint8) if bits else None
            m2 = torch.empty(cur.shape[0], blk['c2'].cout // 8, h // 2, cur.shape[3] // 2, device=cur.device, dtype=torch.uint8) if bits else None
            y1 = blk['c1'].forward(cur, bias=blk['b1'], mask_out=m1, **lr)
            t = K16.upfirdn2d(y1, blk['k'], pad=(2, 2, 2, 2), sep=blk['ksep'])  # Blur before the stride-2 3x3 (networks.py:530-536): (h+1)^2
            y2 = blk['c2'].forward(t, bias=blk['b2'], mask_out=m2, **lr)
            del t
            ts = K16.upfirdn2d(cur, blk['k'], down=2, pad=(1, 1, 1, 1), sep=blk['ksep'])          # skip (networks.py:586-590): the blur only where the stride-2 1x1 samples it
            out = blk['sk1'].forward(ts, residual=y2, out_gain=1.0 / SQRT2)      # (conv2 + skip) / sqrt2
            del ts
            if keep:
                saved.append((m1, m2, (h, cur.shape[3])) if bits else (y1, y2, (h, cur.shape[3])))
            cur = out
            _probe('D.fwd.out@%d' % h, out)
        ctx.net, ctx.saved, ctx.in_hw = net, saved if keep else None, (x.shape[2], x.shape[3])
        return K16.cast_from_h8(cur)                                               # [B,512,4,4] fp32 for the tail

    @staticmethod
    def backward(ctx, g32):
        net, saved = ctx.net, ctx.saved
        if saved is None:
            raise RuntimeError('discriminator was run without a differentiable input')
        S, dyn = _gs(net, 'D'), _dyn(net)
        if dyn is not None:
            g32 = g32 * dyn
        bg = F16_D_BLOCK_GAIN if dyn is not None else 1.0        # fp16: the gradient is doubled once per block on both branches (undone with S at the image)
        g = K16.cast_to_h8(g32.contiguous() * S if S != 1.0 else g32.contiguous(), dtype=net.dtype)
        for blk, (y1, y2, in_hw) in zip(reversed(net.blocks), reversed(saved[1:])):
            _probe('D.g@%d' % in_hw[0], g)
            h = in_hw[0]
            gm = K16.mask_mul(g, y2, bg, 0.2 * bg)                                # 1/sqrt2 * lrelu' * sqrt2 on the conv2 branch (g also feeds the skip branch)
            g_t = blk['c2'].dgrad(gm, (h + 1, h + 1))
            del gm
            g_y1 = K16.upfirdn2d(g_t, blk['kf'], pad=(1, 1, 1, 1), mask=y1, mask_vals=LRELU_MASK, sep=blk['kfsep'], mask_bits=y1.dtype == torch.uint8)
            del g_t
            g_a = blk['c1'].dgrad(g_y1, in_hw)
            del g_y1
            g_ts = blk['sk1'].dgrad(g, (h // 2, h // 2), out_gain=bg / SQRT2)
            g = K16.upfirdn2d(g_ts, blk['kf'], up=2, pad=(2, 1, 2, 1), addend=g_a, sep=blk['kfsep'])      # adjoint of (blur, every second pixel): zero-insertion FIR
            del g_ts, g_a
        _probe('D.g_last', g)
        g_img = net.conv0.dgrad(K16.mask_mul(g, saved[0], *LRELU_MASK), ctx.in_hw, out_f32=True, out_gain=1.0 / (S * bg ** len(net.blocks)))
        ctx.saved = None
        return g_img, None


# =====================================================================================================================================
# StyleGAN2 generator (generator.py)
# =====================================================================================================================================
class _StyledLayer16:
    def __init__(self, P, prefix, cin, cout, upsample, device):
        w = torch.as_tensor(np.asarray(P[prefix + '.conv.weight']), dtype=torch.float32)[0]
        ws = w * (1.0 / math.sqrt(cin * 9))
        self.cin, self.cout, self.up = cin, cout, upsample
        self.conv = C.H8Conv(ws, stride=2 if upsample else 1, padding=0 if upsample else 1, transposed=upsample, device=device)
        wt = ws.transpose(0, 1).contiguous()
        # fp32 weights in plane order: the style (forward) / the demodulation factor (backward) is folded in per sample (l2i_modulate_planes_h8)
        self.w32_fwd = C.pack_weight_h8_f32(ws).to(device)
        self.w32_bwd = C.pack_weight_h8_f32(wt if upsample else torch.flip(wt, [2, 3])).to(device)
        self.T = (ws * ws).sum((2, 3)).contiguous().to(device)
        self.mod = _Mod(P, prefix + '.conv', device)
        self.noise_w = float(np.asarray(P[prefix + '.noise.weight']).reshape(-1)[0])
        self.bias = _t(P[prefix + '.activate.bias'], device)
        if upsample:
            self.blur_k = _t(P[prefix + '.conv.blur.kernel'], device)
            self.blur_k_flip = torch.flip(self.blur_k, [0, 1]).contiguous()
            kk = np.asarray(P[prefix + '.conv.blur.kernel'])
            self.blur_sep, self.blur_flip_sep = K16.separable(kk), K16.separable(kk[::-1, ::-1])


class Generator(_Generator32):
    def __init__(self, state, size, device='cuda', style_dim=512, n_mlp=8, lr_mlp=0.01):
        self.size, self.device, self.style_dim = size, device, style_dim
        self.dtype = K16.h8_dtype()
        self.log_size = int(math.log2(size))
        self.n_latent = self.log_size * 2 - 2
        self.num_layers = (self.log_size - 2) * 2 + 1
        P = state
        self.mlp = []
        for i in range(1, n_mlp + 1):
            w = _t(P['style.%d.weight' % i], device)
            self.mlp.append(((w * ((1.0 / math.sqrt(w.shape[1])) * lr_mlp)).t().contiguous(), _t(P['style.%d.bias' % i], device) * lr_mlp))
        self.const = _t(P['input.input'], device)
        self.const16 = C.to_h8(self.const, dtype=self.dtype)                    # [1, 64, 4, 4, 8]
        geo, _ = specs.generator_geometry(size)
        self.layers = [_StyledLayer16(P, name, cin, cout, up, device) for name, cin, cout, res, up in geo]
        self.rgbs = [_ToRGB(P, 'to_rgb1', geo[0][2], False, device)]
        for j in range(self.log_size - 2):
            self.rgbs.append(_ToRGB(P, 'to_rgbs.%d' % j, geo[2 + 2 * j][2], True, device))
        self.randomize_noise = True
        self.modplan = _ModPlan(self, device)
        self.mod_fwd = self.mod_bwd = None
        if MOD_MULTI:
            self.mod_fwd = K16.ModulatePlan([L.w32_fwd for L in self.layers], self.modplan.s_off[:len(self.layers)], device, widths=self.modplan.cin)
            self.mod_bwd = K16.ModulatePlan([L.w32_bwd for L in self.layers], self.modplan.d_off, device, widths=self.modplan.cout)

    def synthesis(self, latent, noise=None):
        return _Synthesis16Fn.apply(latent, self, noise)


class _Synthesis16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latent, gen, noise):
        B = latent.shape[0]
        dev = latent.device
        lat = latent.detach()
        keep = latent.requires_grad
        saved = []
        plan = gen.modplan
        s_all, d_all, w_all = plan.forward(lat.contiguous())
        x = gen.const16.expand(B, -1, -1, -1, -1).contiguous()
        skip = None
        lr = dict(act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
        drawn = _draw_noise(gen, noise, B, dev)
        planes_all = gen.mod_fwd.run(s_all, B, gen.dtype) if gen.mod_fwd is not None else None      # weight * style for every layer (networks.py:234-235)
        for li, L in enumerate(gen.layers):
            s, demod = plan.s(s_all, B, li), plan.demod(d_all, B, li)
            h = x.shape[2]
            res = h * 2 if L.up else h
            nz = _noise_for(gen, noise, li, B, res, dev, drawn)
            planes = planes_all[li] if planes_all is not None else K16.modulate_planes(L.w32_fwd, s, dtype=gen.dtype)        # weight * style, one plane set per sample
            bstride = planes[0].numel() * 2
            if L.up:
                t = L.conv.forward(x, planes=planes, w_bstride=bstride, out_scale=demod)          # (2H+1)^2
                y = K16.upfirdn2d(t, L.blur_k, pad=(1, 1, 1, 1), noise=nz, noise_w=L.noise_w, bias=L.bias, sep=L.blur_sep, **lr)
                del t
            else:
                has_rgb = li == 0 or li % 2 == 0
                rgb = None
                if has_rgb and RGB_FUSED and L.cout in (32, 64):                     # ToRGB in this conv's epilogue (the block holds every channel of its pixels)
                    rgb = torch.empty(B, 3, res, res, device=dev, dtype=torch.float32)
                y = L.conv.forward(x, planes=planes, w_bstride=bstride, out_scale=demod, noise=nz, noise_w=L.noise_w, bias=L.bias,
                                   rgb=None if rgb is None else (plan.wmod(w_all, B, li // 2), gen.rgbs[li // 2].bias, rgb), **lr)
            del planes
            if PROBE is not None and li % 2 == 0:
                _probe('G.fwd.y%d@%d' % (li, res), y)
            rec = dict(x=x if keep else None, y=y if keep else None, s=s, demod=demod, nz=nz)
            if li == 0 or (li % 2 == 0):
                R = gen.rgbs[li // 2]
                wmod = plan.wmod(w_all, B, li // 2)
                if L.up or not (RGB_FUSED and L.cout in (32, 64)):
                    rgb = K16.torgb_fwd(y, wmod, R.bias)                       # fp32 [B,3,H,W]: the skip image stays fp32
                skip = K.upfirdn2d(skip, R.up_k, up=(2, 2), pad=(2, 1, 2, 1), addend=rgb) if R.up else rgb
                rec['wmod'] = wmod
            saved.append(rec)
            x = y
        planes_all = None
        ctx.gen, ctx.saved, ctx.B = gen, saved if keep else None, B
        ctx.mod = (s_all, d_all) if keep else None
        return skip

    @staticmethod
    def backward(ctx, g_img):
        gen, saved, B = ctx.gen, ctx.saved, ctx.B
        if saved is None:
            raise RuntimeError('synthesis was run without a differentiable latent')
        dev = g_img.device
        plan = gen.modplan
        s_all, d_all = ctx.mod
        red_dz, q_all, red_rgb = plan.reductions(B, dev)
        n_rgb = len(gen.rgbs)
        g_rgb = [None] * n_rgb
        S = _gs(gen, 'G')
        g = g_img.contiguous() * S if S != 1.0 else g_img.contiguous()
        for j in range(n_rgb - 1, -1, -1):
            g_rgb[j] = g
            if j > 0:
                g = K.upfirdn2d(g, gen.rgbs[j].up_k_flip, up=(1, 1), down=(2, 2), pad=(1, 1, 1, 1))
        gin, gin_scale = None, None
        planes_all = gen.mod_bwd.run(d_all, B, gen.dtype) if gen.mod_bwd is not None else None     # the gradient convs' weights carry the demodulation factor
        for li in range(len(gen.layers) - 1, -1, -1):
            L, rec = gen.layers[li], saved[li]
            has_rgb = 'wmod' in rec
            grgb = g_rgb[li // 2] if has_rgb else None
            dz = K16.sg2_act_bwd(rec['y'], gin, gin_scale, grgb, rec.get('wmod'), L.bias, rec['nz'], L.noise_w, 0.2, SQRT2,
                                 plan.demod(red_dz, B, li), plan.red_rgb(red_rgb, B, li // 2) if has_rgb else None,
                                 red_q=plan.s(q_all, B, li + 1).view(-1) if gin is not None else None)      # layer li + 1's d s = sum_p gin * y (generator.py)
            demod, s = rec['demod'], rec['s']
            x = rec['x']
            hw = (x.shape[2], x.shape[3])
            planes = planes_all[li] if planes_all is not None else K16.modulate_planes(L.w32_bwd, demod, dtype=gen.dtype)
            bstride = planes[0].numel() * 2
            if L.up:
                dt = K16.upfirdn2d(dz, L.blur_k_flip, pad=(2, 2, 2, 2), sep=L.blur_flip_sep)        # gradient of the (2H+1)^2 map under the blur
                del dz
                dxmod = L.conv.dgrad(dt, hw, planes=planes, w_bstride=bstride)
                del dt
            else:
                dxmod = L.conv.dgrad(dz, hw, planes=planes, w_bstride=bstride)
                del dz
            del planes
            if li == 0:
                K16.dot_reduce(dxmod, x, out=plan.s(q_all, B, li).view(-1))      # d s via x * s (layers >= 1: inside the next sg2_act_bwd)
            if PROBE is not None and (li % 2 == 0 or li == len(gen.layers) - 1):
                _probe('G.dx%d@%d' % (li, hw[0]), dxmod)
            gin, gin_scale = dxmod, s
            rec['y'] = rec['x'] = None
        g_lat = plan.backward(B, s_all, d_all, red_dz, q_all, red_rgb)
        if S != 1.0:
            g_lat = g_lat * (1.0 / S)
        if _dyn(gen) is not None:
            g_lat = g_lat * _dyn(gen, inverse=True)
        return g_lat, None, None
