"""Frozen ResNet-50 attribute regressor (torchvision v0.5.0 layout, fc -> 40) on the l2i HIP kernels.

Reference call sites: transform_base.py:522-528 (construction / checkpoint ``ckpt['model']``), :396-403 and :416-424
(``regressor(img)[:, attrIdx]`` on the raw [-1,1] generator output, eval mode :267).  Eval-mode BatchNorm is an affine
map, so it is folded into the preceding conv at construction; ReLU and the residual add live in the conv epilogue;
the backward is input-gradient only, with every ReLU mask applied in the epilogue of the launch that produces the masked
gradient or in the prologue of the one that consumes it (no standalone elementwise passes but the first mask).
"""

import numpy as np
import torch

from . import conv as C
from . import kernels as K
from .specs import RESNET50_LAYERS


def _fold_bn(P, conv_name, bn_name, eps=1e-5):
    w = torch.as_tensor(np.asarray(P[conv_name + '.weight']), dtype=torch.float64)
    g = torch.as_tensor(np.asarray(P[bn_name + '.weight']), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(P[bn_name + '.bias']), dtype=torch.float64)
    m = torch.as_tensor(np.asarray(P[bn_name + '.running_mean']), dtype=torch.float64)
    v = torch.as_tensor(np.asarray(P[bn_name + '.running_var']), dtype=torch.float64)
    k = g / torch.sqrt(v + eps)
    return (w * k.reshape(-1, 1, 1, 1)).float(), (b - m * k).float()


class _CB:
    """conv + folded BN (+ReLU in the epilogue)."""

    def __init__(self, P, conv_name, bn_name, stride, padding, device):
        w, b = _fold_bn(P, conv_name, bn_name)
        self.conv = C.FrozenConv2d(w, stride=stride, padding=padding, device=device)
        self.bias = b.contiguous().to(device)


class ResNet50:
    def __init__(self, state, device='cuda'):
        P = state
        self.device = device
        self.stem = _CB(P, 'conv1', 'bn1', 2, 3, device)
        self.blocks = []
        for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
            for b in range(blocks):
                p = 'layer%d.%d' % (li + 1, b)
                s = stride if b == 0 else 1
                blk = dict(c1=_CB(P, p + '.conv1', p + '.bn1', 1, 0, device),
                           c2=_CB(P, p + '.conv2', p + '.bn2', s, 1, device),       # v1.5: stride on the 3x3
                           c3=_CB(P, p + '.conv3', p + '.bn3', 1, 0, device),
                           down=_CB(P, p + '.downsample.0', p + '.downsample.1', s, 0, device) if b == 0 else None)
                self.blocks.append(blk)
        self.fc_w = torch.as_tensor(np.asarray(P['fc.weight']), dtype=torch.float32).contiguous().to(device)
        self.fc_b = torch.as_tensor(np.asarray(P['fc.bias']), dtype=torch.float32).contiguous().to(device)

    def features(self, img):
        """[B,3,H,W] -> [B, 2048] pooled features (what ``fc`` reads); differentiable w.r.t. img."""
        return _ResNetFeatFn.apply(img, self)

    def __call__(self, img):
        """[B,3,H,W] -> [B, num_classes]; differentiable w.r.t. img."""
        return torch.addmm(self.fc_b, self.features(img), self.fc_w.t())


PAIR = __import__('os').environ.get('L2I_R_PAIR', '1') != '0'            # [r6] chained 1x1 convs of the trunk as one launch (csrc/l2i_pair_f32.hip); 0: separate launches (A/B)
PREMASK = __import__('os').environ.get('L2I_R_PREMASK', '1') != '0'      # 0: the round-5 mask plumbing of the backward (A/B)


def _backward_r5(net, saved, g):
    """The round-5 form (L2I_R_PREMASK=0): g = the gradient w.r.t. a block's ReLU output; its mask rides on BOTH consumers of g."""
    for blk, (y1, y2, out, in_hw) in zip(reversed(net.blocks), reversed(saved['blocks'])):
        pre = blk['c2'].conv.stride == 1
        g_y2 = blk['c3'].conv.dgrad(g, (y2.shape[2], y2.shape[3]), in_mask=out, mask=(1.0, 0.0), **(dict(out_mask=y2) if pre else {}))
        if pre:
            g_y1 = blk['c2'].conv.dgrad(g_y2, (y1.shape[2], y1.shape[3]), out_mask=y1)
            m1 = {}
        else:
            g_y1 = blk['c2'].conv.dgrad(g_y2, (y1.shape[2], y1.shape[3]), in_mask=y2, mask=(1.0, 0.0))
            m1 = dict(in_mask=y1, mask=(1.0, 0.0))
        if blk['down'] is None:
            g_in = blk['c1'].conv.dgrad(g_y1, in_hw, residual=g, res_mask=out, **m1)
        else:
            g_in = blk['c1'].conv.dgrad(g_y1, in_hw, **m1)
            blk['down'].conv.dgrad(g, in_hw, out=g_in, in_mask=out, mask=(1.0, 0.0), accumulate=True)
        g = g_in
    a0 = saved['a0']
    g_a0 = K.maxpool2d_bwd(g, saved['idx0'], (a0.shape[2], a0.shape[3]), 3, 2, 1)
    return net.stem.conv.dgrad(g_a0, saved['in_hw'], in_mask=a0, mask=(1.0, 0.0))


class _ResNetFeatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, net):
        keep = img.requires_grad
        x = img.detach().contiguous()
        a0 = net.stem.conv.forward(x, bias=net.stem.bias, act=C.ACT_RELU)
        p0, idx0 = K.maxpool2d_fwd(a0, 3, 2, 1)
        saved = dict(in_hw=(x.shape[2], x.shape[3]), a0=a0 if keep else None, idx0=idx0 if keep else None, blocks=[])
        cur = p0
        # [r6] PAIR: conv3 + identity + ReLU of a block and conv1 + ReLU of the NEXT block as one launch where the library has the shape (l2i_conv1x1_pair_f32:
        # the wide map goes from the first conv's accumulators to the second conv's MFMAs in registers — written once, not read back)
        ahead = None
        n_blk = len(net.blocks)
        for bi, blk in enumerate(net.blocks):
            y1 = ahead if ahead is not None else blk['c1'].conv.forward(cur, bias=blk['c1'].bias, act=C.ACT_RELU)
            y2 = blk['c2'].conv.forward(y1, bias=blk['c2'].bias, act=C.ACT_RELU)
            if blk['down'] is not None:
                idt = blk['down'].conv.forward(cur, bias=blk['down'].bias)
            else:
                idt = cur
            nxt = net.blocks[bi + 1] if bi + 1 < n_blk else None
            ahead = None
            c3 = blk['c3'].conv
            if (PAIR and C.PRECISION == 'f32' and nxt is not None and c3.k == 1 and c3.stride == 1 and nxt['c1'].conv.stride == 1
                    and C.pair_f32_shapes_ok(c3.cin, c3.cout, nxt['c1'].conv.cout, y2.shape[2] * y2.shape[3])):
                d = []
                out = c3.forward(y2, bias=blk['c3'].bias, residual=idt, act=C.ACT_RELU, _defer=d)
                ahead = nxt['c1'].conv.forward(out, bias=nxt['c1'].bias, act=C.ACT_RELU, _defer=d)
                C.launch_pair_f32(d)
            else:
                out = c3.forward(y2, bias=blk['c3'].bias, residual=idt, act=C.ACT_RELU)
            if keep:
                saved['blocks'].append((y1, y2, out, (cur.shape[2], cur.shape[3])))
            cur = out
        b, c, h, w = cur.shape
        feat = K.dot_reduce(cur) * (1.0 / (h * w))             # adaptive avg-pool (1,1)
        ctx.net, ctx.saved, ctx.last_shape = net, saved if keep else None, (b, c, h, w)
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        net, saved = ctx.net, ctx.saved
        if saved is None:
            raise RuntimeError('regressor was run without a differentiable input')
        b, c, h, w = ctx.last_shape
        g = (g_feat * (1.0 / (h * w))).reshape(b, c, 1, 1).expand(b, c, h, w).contiguous()
        blocks = saved['blocks']
        # [r6] G = the gradient w.r.t. the PRE-ReLU sum of a block (out = relu(c3(y2) + idt)): the ReLU mask of the block input (the previous block's
        # `out`, the 4x-wide map) is applied ONCE, by the launch that produces the block-input gradient (out_mask / res_mask = the same tensor: one
        # operand fetch in the 1x1 GEMM's epilogue), instead of twice by the two consumers of g (round 5: in_mask = out on c3's gradient GEMM — a
        # second DMA stream and 3 VALU per fragment in its K loop — and res_mask = out on c1's): one read of every wide map less per block (3.7 GB per
        # 1024^2 batch-8 step) and c3's gradient conv becomes the unmasked GEMM.  The 16-bit path has always done this (nets16._ResNet16Fn).
        if not PREMASK:
            return _backward_r5(net, saved, g), None
        G = K.relu_mask(g, blocks[-1][2])
        del g
        for bi in range(len(net.blocks) - 1, -1, -1):
            blk, (y1, y2, out, in_hw) = net.blocks[bi], blocks[bi]
            m = blocks[bi - 1][2] if bi > 0 else None          # the block input is the previous block's ReLU output (the pooled stem map is not)
            mk = dict(out_mask=m) if m is not None else {}
            # [r4] The masks of y2 / y1 are applied by the PRODUCING launch's epilogue (out_mask: a select on values it holds in registers) instead
            # of the consuming launch's prologue (in_mask): the same numbers, and the 3x3 gradient conv becomes an unmasked launch — the F(4x4,3x3)
            # Winograd kernel (csrc/l2i_wino4.hip) and the unmasked transposed instantiation take those
            # (the three stride-2 blocks keep the prologue masks: their c2 gradient is the one-launch transposed kernel, which fuses in_mask only)
            pre = blk['c2'].conv.stride == 1
            g_y2 = blk['c3'].conv.dgrad(G, (y2.shape[2], y2.shape[3]), **(dict(out_mask=y2) if pre else {}))
            if pre:
                g_y1 = blk['c2'].conv.dgrad(g_y2, (y1.shape[2], y1.shape[3]), out_mask=y1)
                m1 = {}
            else:
                g_y1 = blk['c2'].conv.dgrad(g_y2, (y1.shape[2], y1.shape[3]), in_mask=y2, mask=(1.0, 0.0))
                m1 = dict(in_mask=y1, mask=(1.0, 0.0))
            del g_y2
            if blk['down'] is None:
                Gp = blk['c1'].conv.dgrad(g_y1, in_hw, residual=G, **(dict(out_mask=m, res_mask=m) if m is not None else {}), **m1)
            elif blk['down'].conv.stride == 1:
                t = blk['c1'].conv.dgrad(g_y1, in_hw, **m1)
                Gp = blk['down'].conv.dgrad(G, in_hw, residual=t, **(dict(out_mask=m, res_mask=m) if m is not None else {}))
                del t
            else:
                Gp = blk['c1'].conv.dgrad(g_y1, in_hw, **mk, **m1)
                blk['down'].conv.dgrad(G, in_hw, out=Gp, accumulate=True, **mk)      # strided 1x1: lands on every second pixel, masked there
            del g_y1
            G = Gp
        g = G
        a0 = saved['a0']
        g_a0 = K.maxpool2d_bwd(g, saved['idx0'], (a0.shape[2], a0.shape[3]), 3, 2, 1)
        g_img = net.stem.conv.dgrad(g_a0, saved['in_hw'], in_mask=a0, mask=(1.0, 0.0))
        ctx.saved = None
        return g_img, None
