"""StyleGAN2 discriminator (frozen, random-init in the reference: transform_base.py:540-548 never loads it) on the l2i
HIP kernels.  Reference: networks.py:517-645 (ConvLayer, ResBlock, Discriminator).

Body (from_rgb conv + residual blocks) = one autograd function with hand-scheduled input-gradient; the 4x4 tail
(minibatch-stddev, final conv, two linears) runs on small torch ops + one FrozenConv2d call.
EqualConv2d's 1/sqrt(fan_in) is folded into the packed weights; FusedLeakyReLU bias+activation, the residual add and
the 1/sqrt(2) live in the conv epilogue; leaky-ReLU' is a prologue mask of the gradient convs.
"""
import math

import numpy as np
import torch

from . import conv as C
from . import kernels as K

SQRT2 = math.sqrt(2.0)
LRELU_MASK = (SQRT2, 0.2 * SQRT2)          # d/dx [lrelu(x, 0.2) * sqrt2] keyed on the sign of the saved output


def _eq_conv(P, name, stride, padding, device):
    w = torch.as_tensor(np.asarray(P[name]), dtype=torch.float32)
    w = w * (1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3]))
    return C.FrozenConv2d(w, stride=stride, padding=padding, device=device)


def _skip_compact(h):
    """Maps wide enough for the down-2 streaming FIR (csrc/l2i_stream.hip) take the compact skip path."""
    return h >= 192 and h % 8 == 0


def _vec(P, name, device):
    return torch.as_tensor(np.asarray(P[name]), dtype=torch.float32).contiguous().to(device)


class Discriminator:
    def __init__(self, state, size, device='cuda'):
        P = state
        self.size, self.device = size, device
        log_size = int(math.log2(size))
        self.conv0 = _eq_conv(P, 'convs.0.0.weight', 1, 0, device)
        self.bias0 = _vec(P, 'convs.0.1.bias', device)
        self.blocks = []
        for n in range(1, log_size - 1):
            p = 'convs.%d' % n
            self.blocks.append(dict(
                c1=_eq_conv(P, p + '.conv1.0.weight', 1, 1, device), b1=_vec(P, p + '.conv1.1.bias', device),
                c2=_eq_conv(P, p + '.conv2.1.weight', 2, 0, device), b2=_vec(P, p + '.conv2.2.bias', device),
                sk=_eq_conv(P, p + '.skip.1.weight', 2, 0, device),
                sk1=_eq_conv(P, p + '.skip.1.weight', 1, 0, device),       # the same 1x1 at stride 1, for blur maps evaluated at the sampled pixels only
                k=_vec(P, p + '.conv2.0.kernel', device)))
        for blk in self.blocks:
            blk['kf'] = torch.flip(blk['k'], [0, 1]).contiguous()
        self.final_conv = _eq_conv(P, 'final_conv.0.weight', 1, 1, device)
        self.final_bias = _vec(P, 'final_conv.1.bias', device)
        w = _vec(P, 'final_linear.0.weight', device)
        self.lin0_wt = (w * (1.0 / math.sqrt(w.shape[1]))).t().contiguous()
        self.lin0_b = _vec(P, 'final_linear.0.bias', device)
        w = _vec(P, 'final_linear.1.weight', device)
        self.lin1_wt = (w * (1.0 / math.sqrt(w.shape[1]))).t().contiguous()
        self.lin1_b = _vec(P, 'final_linear.1.bias', device)

    def __call__(self, img):
        """[B,3,size,size] -> [B,1] logits (networks.py:627-645)."""
        from .op import fused_leaky_relu
        out = self.body(img)                                                   # [B,512,4,4]
        batch, channel, height, width = out.shape
        group = min(batch, 4)
        if batch % group != 0:
            raise ValueError('minibatch-stddev needs a batch divisible by %d (networks.py:631-634)' % group)
        sd = out.view(group, -1, 1, channel, height, width)
        sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8)
        sd = sd.mean([2, 3, 4], keepdim=True).squeeze(2)
        sd = sd.repeat(group, 1, height, width)
        out = torch.cat([out, sd], 1)
        out = _ConvLReLUFn.apply(out, self.final_conv, self.final_bias)
        out = out.reshape(batch, -1)
        out = fused_leaky_relu(torch.mm(out, self.lin0_wt), self.lin0_b)
        return torch.addmm(self.lin1_b, out, self.lin1_wt)


    def body(self, img):
        """from_rgb conv + residual blocks: [B,3,size,size] -> [B,512,4,4] fp32 (the 16-bit path overrides this, nets16.Discriminator)."""
        return _DBodyFn.apply(img, self)


class _ConvLReLUFn(torch.autograd.Function):
    """conv + FusedLeakyReLU with frozen weights (ConvLayer with activate=True, networks.py:545-558)."""

    @staticmethod
    def forward(ctx, x, conv, bias):
        y = conv.forward(x.detach().contiguous(), bias=bias, act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
        # y is an OUTPUT of this node: saved through save_for_backward (a plain attribute would be a reference cycle node -> y -> node that keeps
        # the whole discriminator graph of the step alive until the cyclic collector runs — 55 MB per 1024^2 step, and the collection it
        # eventually triggers is a 70+ ms host stall)
        ctx.save_for_backward(y)
        ctx.conv, ctx.in_hw = conv, (x.shape[2], x.shape[3])
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        gx = ctx.conv.dgrad(g.contiguous(), ctx.in_hw, in_mask=y, mask=LRELU_MASK)
        return gx, None, None


class _DBodyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, net):
        keep = img.requires_grad
        x = img.detach().contiguous()
        y0 = net.conv0.forward(x, bias=net.bias0, act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
        saved = [y0]
        cur = y0
        for blk in net.blocks:
            y1 = blk['c1'].forward(cur, bias=blk['b1'], act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
            b_, _, h, _ = cur.shape
            # Blur before the stride-2 3x3 (networks.py:530-536: pad (2,2), an (h+1)^2 map).  The map is produced (h+4)^2 instead — three
            # more zero-padded rows / columns at the far edge that the stride-2 conv never reads — so that its rows are whole 16-byte
            # vectors: the conv kernels stage aligned 4-pixel vectors (an odd 1025-float pitch cannot be staged that way)
            t = K.upfirdn2d(y1, blk['k'], pad=(2, 5, 2, 5))
            y2 = torch.empty(b_, blk['c2'].cout, h // 2, h // 2, device=cur.device, dtype=torch.float32)
            blk['c2'].forward(t, out=y2, bias=blk['b2'], act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
            del t
            out = torch.empty(b_, blk['sk'].cout, h // 2, h // 2, device=cur.device, dtype=torch.float32)
            if _skip_compact(h):
                # Blur before the stride-2 1x1 skip (networks.py:586-590: pad (1,1), an (h-1)^2 map of which the conv reads every second pixel):
                # only those pixels are computed (FIR with down = 2: a quarter of the writes), and the 1x1 runs at stride 1 on the compact
                # map — the DMA-fed GEMM kernel instead of a strided gather.  Same taps, same sums: identical values.
                ts = K.upfirdn2d(cur, blk['k'], down=(2, 2), pad=(1, 1, 1, 1))
                blk['sk1'].forward(ts, out=out, residual=y2, out_gain=1.0 / SQRT2)
            else:
                ts = K.upfirdn2d(cur, blk['k'], pad=(1, 2, 1, 2))              # small maps: the full (h-1)^2 -> h^2 map (16-byte rows), stride-2 conv
                blk['sk'].forward(ts, out=out, residual=y2, out_gain=1.0 / SQRT2)  # (conv2 + skip) / sqrt2
            del ts
            if keep:
                saved.append((y1, y2, (cur.shape[2], cur.shape[3])))
            cur = out
        ctx.net, ctx.saved, ctx.in_hw = net, saved if keep else None, (x.shape[2], x.shape[3])
        return cur

    @staticmethod
    def backward(ctx, g):
        net, saved = ctx.net, ctx.saved
        if saved is None:
            raise RuntimeError('discriminator was run without a differentiable input')
        g = g.contiguous()
        for blk, (y1, y2, in_hw) in zip(reversed(net.blocks), reversed(saved[1:])):
            h = in_hw[0]
            # conv2 path: lrelu' * 1/sqrt2 folded into the prologue mask
            # (gradients of the padded blur maps of the forward: the extra rows / columns receive zeros and are cropped by the negative far pad)
            g_t = blk['c2'].dgrad(g, (h + 4, h + 4), in_mask=y2, mask=(1.0, 0.2))
            g_y1 = K.upfirdn2d(g_t, blk['kf'], pad=(1, -2, 1, -2), mask=y1, mask_vals=LRELU_MASK)     # [r5] the leaky-ReLU mask of y1 rides on the FIR: the 3x3
                                                                                                     # gradient conv below is unmasked = the F(4x4) kernel
            del g_t
            g_a = blk['c1'].dgrad(g_y1, in_hw)
            del g_y1
            # skip path
            if _skip_compact(h):
                # adjoint of (blur, keep every second pixel): zero-insertion FIR (up = 2) of the compact gradient (op/upfirdn2d.py:105-115: g_pad = (2, 1))
                g_ts = blk['sk1'].dgrad(g, (h // 2, h // 2), out_gain=1.0 / SQRT2)
                g = K.upfirdn2d(g_ts, blk['kf'], up=(2, 2), pad=(2, 1, 2, 1), addend=g_a)
            else:
                g_ts = blk['sk'].dgrad(g, (h, h), out_gain=1.0 / SQRT2)
                g = K.upfirdn2d(g_ts, blk['kf'], pad=(2, 1, 2, 1), addend=g_a)
            del g_ts, g_a
        g_img = net.conv0.dgrad(g, ctx.in_hw, in_mask=saved[0], mask=LRELU_MASK)
        ctx.saved = None
        return g_img, None
