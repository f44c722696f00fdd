"""VGG-19 ``features`` prefix content loss on the l2i HIP kernels.

Reference: transform_base.py:426-454 (get_content_loss), :44-63 (Normalization, ContentLoss), :465-470 (mean of the four
taps).  The reference rebuilds and re-runs the growing VGG prefix once per tap (conv_1 four times, conv_2 three times,
…); the taps are the same tensors each time, so here the prefix is evaluated once per image.  The four taps are the
PRE-ReLU conv outputs; the ReLU (and the ReLU after the max-pool, which commutes with it) is applied as an input mask in
the prologue of the next conv, and as an output mask in the epilogue of the gradient convs.
Normalisation (x - mean)/std: the per-channel 1/std is folded into the first conv's weights, x - mean is one
``fused_bias_act`` call (zero padding of the normalised image is preserved exactly).
"""
import os as _os

import numpy as np
import torch

from . import conv as C
from . import kernels as K

VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)

POOL_FUSED = _os.environ.get('L2I_POOL_FUSED', '1') != '0'     # [r5] VGG pool1 inside conv1_2's F(4x4) epilogue (0: its own launch; A/B)


class VGG19Prefix:
    def __init__(self, state, device='cuda'):
        P = state
        self.device = device
        w0 = torch.as_tensor(np.asarray(P['0.weight']), dtype=torch.float32)
        w0 = w0 / torch.tensor(VGG_STD, dtype=torch.float32).reshape(1, 3, 1, 1)
        ws = [w0] + [torch.as_tensor(np.asarray(P['%d.weight' % i]), dtype=torch.float32) for i in (2, 5, 7)]
        self.convs = [C.FrozenConv2d(w, 1, 1, device=device) for w in ws]
        self.biases = [torch.as_tensor(np.asarray(P['%d.bias' % i]), dtype=torch.float32).contiguous().to(device)
                       for i in (0, 2, 5, 7)]
        self.neg_mean = (-torch.tensor(VGG_MEAN, dtype=torch.float32)).to(device)

    def taps(self, img, org=None):
        """[B,3,H,W] -> (c1, c2, p, c3, c4, pool_idx): pre-ReLU conv outputs conv_1..conv_4, p = maxpool(c2).
        ``org`` = the four taps of the original image: returns a seventh element, the four sums of (c_k - org_k)^2 as 1-element tensors.
        Where a tap's conv runs on the Winograd kernel the sum is formed in its epilogue (the map is still in registers: no second pass
        over it); otherwise by ``sqdiff``."""
        xc = K.fused_bias_act(img.contiguous(), self.neg_mean, None, 1, 0, 0.0, 1.0)        # x - mean
        sq = [None] * 4
        if org is not None:
            acc = torch.zeros(4, C._lib.SQ_SLOTS, device=img.device, dtype=torch.float32)
            sq = [(org[k], acc[k], [False]) for k in range(4)]
        c1 = self.convs[0].forward(xc, bias=self.biases[0], sq=sq[0])
        # [r5] the 2x2 pool rides on conv1_2's F(4x4) epilogue (a lane holds a 4x4 output tile: lane-local) where that kernel runs: no pass over the 2 GB map
        pool = None
        if POOL_FUSED and c1.shape[2] % 2 == 0 and c1.shape[3] % 4 == 0:
            oc = self.convs[1].cout
            pool = (torch.empty(c1.shape[0], oc, c1.shape[2] // 2, c1.shape[3] // 2, device=c1.device, dtype=torch.float32),
                    torch.empty(c1.shape[0], oc, c1.shape[2] // 2, c1.shape[3] // 2, device=c1.device, dtype=torch.uint8), [False])
        c2 = self.convs[1].forward(c1, in_mask=c1, mask=(1.0, 0.0), bias=self.biases[1], sq=sq[1], pool=pool)
        if pool is not None and pool[2][0]:
            p, idx = pool[0], pool[1]
        else:
            p, idx = K.maxpool2d_fwd(c2, 2, 2, 0)                 # relu(maxpool(.)) == maxpool(relu(.))
        c3 = self.convs[2].forward(p, in_mask=p, mask=(1.0, 0.0), bias=self.biases[2], sq=sq[2])
        c4 = self.convs[3].forward(c3, in_mask=c3, mask=(1.0, 0.0), bias=self.biases[3], sq=sq[3])
        if org is None:
            return c1, c2, p, c3, c4, idx
        mine = (c1, c2, c3, c4)
        sums = [acc[k].sum().reshape(1) if sq[k][2][0] else K.sqdiff(org[k], mine[k])[0] for k in range(4)]
        return c1, c2, p, c3, c4, idx, sums

    def org_taps(self, org):
        """The four taps of the original image (no gradient): what ``content_losses`` compares against."""
        with torch.no_grad():
            o1, o2, _, o3, o4, _ = self.taps(org.detach())
        return (o1, o2, o3, o4)

    def content_losses(self, org, shifted, org_taps=None):
        """Four mse(feat_k(org).detach(), feat_k(shifted)) scalars as one [4] tensor, differentiable w.r.t. shifted.  ``org_taps``: the result of
        ``org_taps(org)`` when the caller has already started it (graph.TransformGraph.prefetch_content_taps)."""
        return _ContentFn.apply(shifted, self, org_taps if org_taps is not None else self.org_taps(org))


class _ContentFn(torch.autograd.Function):
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
        gl = [g_losses[k:k + 1].contiguous() for k in range(4)]   # device scalars: no host sync in the backward
        # direct term of every tap: d/d c_k [ mean (c_k - o_k)^2 ] = 2 (c_k - o_k) / N_k
        d4 = K.sqdiff(o4, c4, coef=2.0 / c4.numel(), coef_dev=gl[3], want_grad=True, want_sum=False)[1]
        # the direct term of tap 3 is formed inside the gradient conv's epilogue: + 2/N g (c3 - o3)
        g3 = net.convs[3].dgrad(d4, (c3.shape[2], c3.shape[3]), out_mask=c3, residual=c3, res_sub=o3, res_coef=2.0 / c3.numel(), res_coef_dev=gl[2])
        del d4
        gp = net.convs[2].dgrad(g3, (p.shape[2], p.shape[3]), out_mask=p)
        del g3
        if c2.shape[3] % 4 == 0 and c2.shape[2] % 2 == 0:      # pool backward + direct term of tap 2 in one pass
            g2 = K.maxpool2x2_bwd_add_diff(gp, idx, o2, c2, 2.0 / c2.numel(), gl[1])
        else:
            g2 = K.maxpool2d_bwd(gp, idx, (c2.shape[2], c2.shape[3]), 2, 2, 0)
            d2 = K.sqdiff(o2, c2, coef=2.0 / c2.numel(), coef_dev=gl[1], want_grad=True, want_sum=False)[1]
            K.axpby(g2, d2, 1.0, 1.0, out=g2)
            del d2
        del gp
        g1 = net.convs[1].dgrad(g2, (c1.shape[2], c1.shape[3]), out_mask=c1, residual=c1, res_sub=o1, res_coef=2.0 / c1.numel(), res_coef_dev=gl[0])
        del g2
        g_img = net.convs[0].dgrad(g1, ctx.in_hw)
        ctx.acts = ctx.org = None
        return g_img, None, None
