"""Oracle: walk, losses, optimiser and the whole per-batch training step (plain torch autograd on CPU).
TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nets, sg2


def to_torch(state, dtype=torch.float32):
    return {k: (torch.from_numpy(np.asarray(v)).to(dtype) if np.asarray(v).dtype.kind == 'f'
                else torch.from_numpy(np.asarray(v))) for k, v in state.items()}


def get_w(PG, z, n_latent):
    """TransformGraph.get_w (transform_base.py:372-378): [style(z)] * n_latent (same tensor object)."""
    w = sg2.style_mlp(PG, z)
    return [w] * n_latent


def walk_linear_multi_w(ws, alpha, walk_w, layers=None):
    """WalkLinearMultiW.forward (transform_base.py:151-165): w_i + alpha @ W[:, i, :] for i in layers (or all)."""
    out = []
    for i in range(len(ws)):
        if layers is None or i in layers:
            out.append(ws[i] + torch.mm(alpha, walk_w[:, i, :]))
        else:
            out.append(ws[i])
    return out


def get_logits(PG, ws, noise=None):
    """TransformGraph.get_logits, 'w' branch (transform_base.py:348-355): stack -> [B,n_latent,512] -> netG."""
    latent = torch.stack(ws).transpose(0, 1)
    return sg2.generator_synthesis(PG, latent, noise)


def get_reg_preds(PR, img, attr_idx):
    """TransformGraph.get_reg_preds (transform_base.py:396-403): integer column select, bit-exact."""
    preds = nets.resnet50_forward(PR, img)[:, attr_idx]
    if preds.dim() == 1:
        preds = preds.unsqueeze(1)
    return preds


def get_alphas(alpha_org, alpha_target):
    """sg2 TransformGraph.get_alphas (transform_base.py:405-408): epsilon = target - org."""
    return alpha_target - alpha_org


def get_alphas_clamp(alpha_org, alpha_delta):
    """pggan variant used by train_multi_attr.py:113 (graphs/pggan/transform_base.py:358-364)."""
    alpha_target = torch.clamp(alpha_org + alpha_delta, min=0, max=1)
    return alpha_target, alpha_target - alpha_org


def bce_loss(pred, y, eps=1e-12):
    """get_bce_loss (transform_base.py:412-414) on RAW regressor outputs."""
    return -(y * pred.clamp(min=eps).log() + (1 - y) * (1 - pred).clamp(min=eps).log()).mean()


def reg_loss(PR, logit, alpha, attr_idx):
    """get_reg_loss (transform_base.py:416-424): alpha -> float64, so the loss is float64."""
    preds = nets.resnet50_forward(PR, logit)[:, attr_idx]
    return bce_loss(preds, alpha.to(torch.double)).mean()


def content_loss(PV, org, shifted):
    """get_content_loss + averaging in optimizeParametersAll (transform_base.py:426-454,465-470):
    mean over 4 taps of mse(feat(org).detach(), feat(shifted))."""
    with torch.no_grad():
        fo = nets.vgg19_taps(PV, org)
    fs = nets.vgg19_taps(PV, shifted)
    losses = [F.mse_loss(a.detach(), b) for a, b in zip(fo, fs)]
    return sum(losses) / len(losses), losses


def gan_loss(PD, logit):
    """BCEWithLogits(netD(logit), ones) (transform_base.py:460-463)."""
    d = sg2.discriminator_forward(PD, logit)
    return F.binary_cross_entropy_with_logits(d, torch.ones_like(d))


def total_loss(reg, cont, gan, no_content_loss=False, no_gan_loss=False):
    """Weighting rule transform_base.py:475-486."""
    loss = reg if (no_content_loss and no_gan_loss) else 10 * reg
    if not no_content_loss:
        loss = loss + 0.05 * cont
    if not no_gan_loss:
        loss = loss + 0.05 * gan
    return loss


class Adam:
    """torch.optim.Adam(lr, betas=(0.5, 0.99), eps=1e-8) restated (transform_base.py:329-331)."""

    def __init__(self, param, lr, betas=(0.5, 0.99), eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = param, lr, betas[0], betas[1], eps
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.t = 0

    def step(self, grad):
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * grad
        self.v = self.b2 * self.v + (1 - self.b2) * grad * grad
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = self.v.sqrt() / math.sqrt(bc2) + self.eps
        self.p = self.p - (self.lr / bc1) * self.m / denom
        return self.p


def train_step(nets_state, walk_w, z, alpha_for_graph, attr_idx, layers=None, noise=None, noise2=None,
               no_content_loss=False, no_gan_loss=False, clamp_variant=False):
    """One iteration of the hot loop (train.py:48-110 -> transform_base.py:456-490).

    nets_state: dict(G=..., R=..., V=..., D=...) of torch state dicts.  walk_w: [C, n_latent, 512] tensor.
    alpha_for_graph: [B, C] float tensor (``ag`` of train.py:84-85).  Returns a dict with images, predictions,
    the individual loss terms, the total loss (float64) and d loss / d walk_w.
    """
    PG, PR, PV, PD = nets_state['G'], nets_state['R'], nets_state.get('V'), nets_state.get('D')
    n_latent = walk_w.shape[1]
    walk_w = walk_w.detach().clone().requires_grad_(True)
    ws = get_w(PG, z, n_latent)                                           # train.py:62
    x0 = get_logits(PG, ws, noise)                                        # train.py:66
    a0 = get_reg_preds(PR, x0, attr_idx)                                  # train.py:69
    if clamp_variant:                                                     # train_multi_attr.py:113,133
        target, eps = get_alphas_clamp(a0, alpha_for_graph)
    else:                                                                 # train.py:86
        target, eps = alpha_for_graph, get_alphas(a0, alpha_for_graph)
    w1 = walk_linear_multi_w(ws, eps, walk_w, layers)                     # train.py:89
    x1 = get_logits(PG, w1, noise if noise2 is None else noise2)          # train.py:94
    reg = reg_loss(PR, x1, target, attr_idx)
    cont = torch.zeros((), dtype=x1.dtype)
    cont_terms = []
    gan = torch.zeros((), dtype=x1.dtype)
    if not no_content_loss:
        cont, cont_terms = content_loss(PV, x0, x1)
    if not no_gan_loss:
        gan = gan_loss(PD, x1)
    loss = total_loss(reg, cont, gan, no_content_loss, no_gan_loss)
    grad, = torch.autograd.grad(loss, walk_w)
    return dict(x0=x0.detach(), x1=x1.detach(), alpha_org=a0.detach(), eps=eps.detach(), target=target.detach(),
                reg=reg.detach(), cont=cont.detach(), cont_terms=[c.detach() for c in cont_terms],
                gan=gan.detach(), loss=loss.detach(), grad=grad.detach())


def stddev_subgroups(batch, group=4):
    """Index sets that the discriminator's minibatch-stddev layer (networks.py:630-638) reduces over: ``view(group, B/group, ...)``
    puts sample b into subgroup b mod (B/group), group = min(B, 4).  Samples of different subgroups never interact anywhere in
    the step, so a batch can be evaluated one subgroup at a time."""
    g = min(batch, group)
    assert batch % g == 0, 'the reference view() needs batch %% min(batch, 4) == 0'
    n = batch // g
    return [list(range(j, batch, n)) for j in range(n)]


def _bounded_group(args):
    """One minibatch-stddev subgroup of ``train_step_bounded``."""
    nets_state, walk_w, zc, ac, attr_idx, layers, no_content_loss, no_gan_loss, clamp_variant = args
    PG, PR, PV, PD = nets_state['G'], nets_state['R'], nets_state.get('V'), nets_state.get('D')
    n_latent = walk_w.shape[1]
    ww = walk_w.detach().clone().requires_grad_(True)
    with torch.no_grad():
        ws = get_w(PG, zc, n_latent)
        x0 = get_logits(PG, ws, None)
        a0 = get_reg_preds(PR, x0, attr_idx)
    if clamp_variant:
        target, eps = get_alphas_clamp(a0, ac)
    else:
        target, eps = ac, get_alphas(a0, ac)
    w1 = walk_linear_multi_w(ws, eps, ww, layers)
    x1 = get_logits(PG, w1, None)
    gx = torch.zeros_like(x1)
    terms, cont_terms = {}, None
    for name in ('reg', 'cont', 'gan'):
        if (name == 'cont' and no_content_loss) or (name == 'gan' and no_gan_loss):
            terms[name] = torch.zeros((), dtype=x1.dtype)
            continue
        xl = x1.detach().requires_grad_(True)
        if name == 'reg':
            t = reg_loss(PR, xl, target, attr_idx)
            wgt = 1.0 if (no_content_loss and no_gan_loss) else 10.0
        elif name == 'cont':
            t, cts = content_loss(PV, x0, xl)
            cont_terms = [c.detach() for c in cts]
            wgt = 0.05
        else:
            t = gan_loss(PD, xl)
            wgt = 0.05
        g, = torch.autograd.grad(t * wgt, xl)
        gx += g.to(gx.dtype)
        terms[name] = t.detach()
        del t, g, xl
    grad, = torch.autograd.grad(x1, ww, gx)
    return dict(x0=x0, x1=x1.detach(), a0=a0, eps=eps.detach(), target=target.detach(), terms=terms, cont_terms=cont_terms, grad=grad.detach())


def train_step_bounded(nets_state, walk_w, z, alpha_for_graph, attr_idx, layers=None, no_content_loss=False, no_gan_loss=False,
                       clamp_variant=False):
    """Same function as ``train_step`` evaluated with bounded memory, for the BASELINE-size checks (1024^2, batch 8: the
    autograd graph of the whole batch is > 100 GB in float32): one minibatch-stddev subgroup at a time (every loss term is a
    mean over samples and the subgroups are equal-sized, so the batch value is the mean of the subgroup values), and inside a
    subgroup one loss branch at a time — each branch is back-propagated to the edited image, its graph is freed, and the summed
    image gradient goes through the generator once.  Same arithmetic per sample; only summation order of the final means
    differs from ``train_step`` (checked against it in tests/test_oracle_golden.py).
    (Round 4 tried python threads and worker processes over the subgroups: the GPU box gives a pod 16 CPUs of quota, nothing scales there.)"""
    B = z.shape[0]
    groups = stddev_subgroups(B)
    out = dict(x0=[None] * B, x1=[None] * B, alpha_org=[None] * B, eps=[None] * B, target=[None] * B)
    per_group = [_bounded_group((nets_state, walk_w.detach(), z[idx], alpha_for_graph[idx], attr_idx, layers, no_content_loss, no_gan_loss, clamp_variant))
                 for idx in groups]
    acc = dict(reg=0.0, cont=0.0, gan=0.0, grad=0.0, cont_terms=None)
    for idx, r in zip(groups, per_group):
        for k, i in enumerate(idx):
            out['x0'][i], out['x1'][i] = r['x0'][k], r['x1'][k]
            out['alpha_org'][i], out['eps'][i], out['target'][i] = r['a0'][k], r['eps'][k], r['target'][k]
        if r['cont_terms'] is not None:
            acc['cont_terms'] = [c / len(groups) + (a if acc['cont_terms'] else 0.0) for c, a in zip(r['cont_terms'], acc['cont_terms'] or [0.0] * len(r['cont_terms']))]
        for name in ('reg', 'cont', 'gan'):
            acc[name] = acc[name] + r['terms'][name] / len(groups)
        acc['grad'] = acc['grad'] + r['grad'] / len(groups)
    res = {k: torch.stack(v) for k, v in out.items()}
    res.update(reg=acc['reg'], cont=acc['cont'], gan=acc['gan'], cont_terms=acc['cont_terms'] or [],
               loss=total_loss(acc['reg'], acc['cont'], acc['gan'], no_content_loss, no_gan_loss), grad=acc['grad'])
    return res
