"""BASELINE config 1 on the product side: the in-repo PGGAN-256 generator and the PGGAN graph's z-space walk training step on the l2i HIP
kernels.  Same class / method names, argument meaning and error behaviour as the reference:

    graphs/pggan/model_256.py:53-150,188-259     EqualConv2d / PixelNorm / ConvBlock / Generator(code_dim=511, n_label=1)
    graphs/pggan/pggan_256.py:11-51              PGGAN holder (netG / netD)
    graphs/pggan/transform_base.py:86-102        WalkLinearZ_free
    graphs/pggan/transform_base.py:211-510       TransformGraph (get_logits with the bilinear halving, get_z_new_tensor, get_reg_preds,
                                                 the clamp pair get_alphas, the [B,1,C] x [B,C] broadcast of get_reg_loss, the `or`
                                                 loss-weight rule of optimizeParametersAll), :596-640 apply_alpha
    graphs/transform_graph_scene.py:5-125        SceneGraph / faceGraph built over this base

What the shipped reference does differently, and what is kept: its constructor loads the generator from torch.hub
(``facebookresearch/pytorch_GAN_zoo``, :554-566 — un-vendored, needs the network) and leaves the in-repo ``pggan_256.PGGAN`` commented out
(:219-220); config 1 names the in-repo generator, so that is what is built here.  ``model_256.Generator`` takes a 511-d code plus a 1-d label
embedding while ``constants.DIM_Z`` is 512: like the oracle and the fixture generator (tests/golden/make_golden.py::gen_pggan), ``netG`` uses
the first 511 columns of z.  ``model_256.Discriminator.forward`` returns a tuple, on which the reference's ``BCE_loss_logits`` raises: the GAN
term therefore raises here too and config 1 runs with ``--no_gan_loss``.

Arithmetic restructured for the hardware, same function: the 4x4 "conv" of a 1x1 map is one GEMM; every 3x3 conv is a frozen-weight
FrozenConv2d (Winograd / implicit-GEMM MFMA kernels) with the bias in its epilogue; PixelNorm + LeakyReLU is one fused pass
(l2i_pixelnorm_act_f32); with the graph's ``alpha = 0`` the last progression block is multiplied by zero (model_256.py:249-251) and is not
evaluated; ``to_rgb`` (1x1) is applied BEFORE the nearest upsample it commutes with (bit-identical values, a quarter of the work).  The
backward produces d(image)/dz only — every weight is frozen.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import constants, dist, synth
from . import conv as C
from . import kernels as K
from .graph import ContentLoss, FaceTransform, SceneTransform, _checkpoint_or_synthetic, _to_numpy_state
from .perceptual import VGG19Prefix
from .regressor import ResNet50

PG_CHANNELS = ((512, 512), (512, 512), (512, 512), (512, 512), (512, 256), (256, 128), (128, 64), (64, 32), (32, 16))   # model_256.py:208-216


def _t(a, device):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous().to(device)


class _EqualConv:
    """EqualConv2d (model_256.py:53-68,96-104): weight_orig * sqrt(2 / fan_in), plain bias; frozen."""

    def __init__(self, P, name, padding, device):
        w = torch.as_tensor(np.asarray(P[name + '.conv.weight_orig']), dtype=torch.float32)
        fan_in = w.shape[1] * w.shape[2] * w.shape[3]
        self.conv = C.FrozenConv2d(w * math.sqrt(2.0 / fan_in), stride=1, padding=padding, device=device)
        self.bias = _t(P[name + '.conv.bias'], device)


class Generator:
    """Frozen ``model_256.Generator(511, 1)``.  ``netG(z511, step=6, alpha=0)`` -> [B, 3, 4 * 2**step, 4 * 2**step]."""

    def __init__(self, state, device='cuda'):
        P = state
        self.device = device
        self.label = _t(P['label_embed.weight'], device)[0:1]                     # label 0 for every sample (model_256.py:232-234)
        self.code_dim = 512 - self.label.shape[1]
        w0 = torch.as_tensor(np.asarray(P['progression.0.conv.0.conv.weight_orig']), dtype=torch.float32)       # [512, 512, 4, 4]
        w0 = w0 * math.sqrt(2.0 / (w0.shape[1] * 16))
        # 4x4 kernel, padding 3, on a 1x1 map: out[b, co, y, x] = sum_ci in[b, ci] * w[co, ci, 3 - y, 3 - x] -> one [512] x [512, 512*16] GEMM
        self.w0 = torch.flip(w0, [2, 3]).permute(1, 0, 2, 3).reshape(w0.shape[1], -1).contiguous().to(device)
        self.b0 = _t(P['progression.0.conv.0.conv.bias'], device)
        self.n_blocks = len(PG_CHANNELS)
        self.convs = []
        for i in range(self.n_blocks):
            first = None if i == 0 else _EqualConv(P, 'progression.%d.conv.0' % i, 1, device)
            self.convs.append((first, _EqualConv(P, 'progression.%d.conv.3' % i, 1, device)))
        self.rgb = [(C.FrozenConv2d(np.asarray(P['to_rgb.%d.weight' % i]), 1, 0, device=device), _t(P['to_rgb.%d.bias' % i], device))
                    for i in range(self.n_blocks)]

    def __call__(self, input, label=None, step=6, alpha=0):
        if not 0 <= step < self.n_blocks:
            raise IndexError('step %r outside the %d progression blocks' % (step, self.n_blocks))
        if input.shape[1] != self.code_dim:
            raise RuntimeError('model_256.Generator(%d, %d) takes a %d-d code, got %d' % (self.code_dim, self.label.shape[1], self.code_dim, input.shape[1]))
        return _PGFn.apply(input, self, int(step), float(alpha))

    def eval(self):
        return self


class _PGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gen, step, alpha):
        keep = z.requires_grad
        zc = z.detach().contiguous()
        B = zc.shape[0]
        code = K.pixelnorm_act(zc, slope=1.0)                                       # code_norm (model_256.py:230)
        x = torch.cat([code, gen.label.expand(B, -1)], 1)
        saved = []
        blend = step > 0 and 0 <= alpha < 1                                        # (1 - alpha) * to_rgb[step-1](upsampled) + alpha * to_rgb[step](block)
        last = step - 1 if (blend and alpha == 0) else step                       # alpha = 0: the last block is multiplied by zero
        out = None
        for i in range(last + 1):
            if i == 0:
                a1 = torch.addmm(gen.b0.repeat_interleave(16), x, gen.w0).reshape(B, -1, 4, 4)
            else:
                a1 = gen.convs[i][0].conv.forward(K.upsample2x_nearest(out), bias=gen.convs[i][0].bias)
            h1 = K.pixelnorm_act(a1, slope=0.2)
            a2 = gen.convs[i][1].conv.forward(h1, bias=gen.convs[i][1].bias)
            prev, out = out, K.pixelnorm_act(a2, slope=0.2)
            saved.append((a1 if keep else None, a2 if keep else None))
        if not blend:
            img = gen.rgb[step][0].forward(out, bias=gen.rgb[step][1])
        elif alpha == 0:
            img = K.upsample2x_nearest(gen.rgb[step - 1][0].forward(out, bias=gen.rgb[step - 1][1]))     # to_rgb commutes with the nearest upsample
        else:
            skip = K.upsample2x_nearest(gen.rgb[step - 1][0].forward(prev, bias=gen.rgb[step - 1][1]))
            img = gen.rgb[step][0].forward(out, bias=gen.rgb[step][1])
            img = K.axpby(skip, img, 1.0 - alpha, alpha)
        ctx.gen, ctx.step, ctx.alpha, ctx.blend, ctx.last = gen, step, alpha, blend, last
        ctx.saved = saved if keep else None
        ctx.zc = zc if keep else None
        return img

    @staticmethod
    def backward(ctx, g_img):
        gen, step, alpha, saved = ctx.gen, ctx.step, ctx.alpha, ctx.saved
        if saved is None:
            raise RuntimeError('the generator was run without a differentiable latent')
        g_img = g_img.contiguous()
        B = g_img.shape[0]
        g_prev = None                                                               # gradient reaching block `last - 1`'s output through the skip branch
        if not ctx.blend:
            g = gen.rgb[step][0].dgrad(g_img, saved[-1][1].shape[2:])
        elif alpha == 0:
            g = gen.rgb[step - 1][0].dgrad(K.pool2x2(g_img, 1.0), saved[-1][1].shape[2:])
        else:
            g = gen.rgb[step][0].dgrad(K.axpby(g_img, None, alpha, 0.0), saved[-1][1].shape[2:])
            g_prev = gen.rgb[step - 1][0].dgrad(K.pool2x2(g_img, 1.0 - alpha), saved[-2][1].shape[2:])
        for i in range(ctx.last, -1, -1):
            a1, a2 = saved[i]
            g = K.pixelnorm_act_bwd(g, a2, slope=0.2)
            g = gen.convs[i][1].conv.dgrad(g, a1.shape[2:])
            g = K.pixelnorm_act_bwd(g, a1, slope=0.2)
            if i == 0:
                g_x = torch.mm(g.reshape(B, -1), gen.w0.t())                        # [B, 512]: code (511) | label embedding (frozen)
                break
            g_up = gen.convs[i][0].conv.dgrad(g, (a1.shape[2], a1.shape[3]))
            g = K.pool2x2(g_up, 1.0)                                                # adjoint of the nearest upsample
            if g_prev is not None and i == ctx.last:
                g = K.axpby(g, g_prev, 1.0, 1.0)
        g_z = K.pixelnorm_act_bwd(g_x[:, :gen.code_dim].contiguous(), ctx.zc, slope=1.0)
        ctx.saved = ctx.zc = None
        return g_z, None, None, None


class PGGAN:
    """Attribute contract of pggan_256.PGGAN (pggan_256.py:11-30): ``netG`` / ``netD``.  ``netD`` is None: the graph's GAN term cannot be
    evaluated on the in-repo discriminator (see the module docstring)."""

    def __init__(self, netG, netD=None):
        self.netG, self.netD = netG, netD


class WalkLinearZ_free(nn.Module):
    """Input-dependent linear walk in z (transform_base.py:86-102): z + alpha * z * w, w [n_attr, dim_z] ~ N(0, 0.02) from the global numpy RNG."""

    def __init__(self, dim_z, step, Nsliders, attrList):
        super().__init__()
        self.dim_z = dim_z
        self.step = step
        self.Nsliders = Nsliders
        self.w = nn.Parameter(torch.Tensor(np.random.normal(0.0, 0.02, [len(attrList), self.dim_z])))

    def forward(self, input, alpha, layers=None, name=None, index_=None):
        al = alpha.to(self.w.device)
        direction = al * input * self.w
        return input + direction


WalkLinearZ_free.__module__ = 'graphs.pggan.transform_base'            # pickles resolve in the reference's vis_w.py


def load_networks(device, need_vgg=True):
    """Frozen nets of the PGGAN graph.  The reference hard-codes private checkpoint paths (transform_base.py:536-538, 578-582): a configured
    path that does not exist is an error unless synthetic weights were requested explicitly (graph._checkpoint_or_synthetic)."""
    src = {}
    if _checkpoint_or_synthetic('PGGAN generator (pg_path)', constants.pg_path):
        ckpt = torch.load(constants.pg_path, map_location='cpu')['G']
        g_state = _to_numpy_state({k[7:] if k.startswith('module.') else k: v for k, v in ckpt.items()})           # transform_base.py:586-590
        src['G'] = constants.pg_path
    else:
        g_state = synth.pggan_generator_state(seed=constants.SYNTH_SEED_G)
        src['G'] = 'synthetic(seed=%d)' % constants.SYNTH_SEED_G
    if _checkpoint_or_synthetic('regressor (reg_path)', constants.reg_path):
        r_state = _to_numpy_state(torch.load(constants.reg_path, map_location='cpu')['model'])
        src['R'] = constants.reg_path
    else:
        r_state = synth.resnet50_state(seed=constants.SYNTH_SEED_R)
        src['R'] = 'synthetic(seed=%d)' % constants.SYNTH_SEED_R
    vgg = None
    if need_vgg:
        if _checkpoint_or_synthetic('VGG-19 (vgg_path; the reference downloads torchvision weights)', constants.vgg_path):
            v_state = {k.replace('features.', ''): v for k, v in _to_numpy_state(torch.load(constants.vgg_path, map_location='cpu')).items()}
            src['V'] = constants.vgg_path
        else:
            v_state = synth.vgg19_prefix_state(seed=constants.SYNTH_SEED_V)
            src['V'] = 'synthetic(seed=%d)' % constants.SYNTH_SEED_V
        vgg = VGG19Prefix(v_state, device=device)
    return Generator(g_state, device=device), ResNet50(r_state, device=device), vgg, src


class TransformGraph:
    def __init__(self, lr, walk_type, nsliders, loss_type, eps, N_f, trainEmbed, attrList, attrTable, layers, pgan_opts=None, nets=None):
        assert (loss_type in ['l2', 'lpips']), 'unimplemented loss'
        if not torch.cuda.is_available():
            raise RuntimeError('latent2im_amd runs on an MI355X (ROCm) device only: no GPU is visible and there is no CPU path')
        self.lr = lr
        self.useGPU = constants.useGPU
        self.device = torch.device('cuda', torch.cuda.current_device())
        if nets is None:
            nets = load_networks(self.device)
        netG, self.regressor, self.vgg19, self.weight_sources = nets
        self.module = PGGAN(netG, None)
        self.reg_optmizer = None
        self.dim_z = constants.DIM_Z
        self.Nsliders = nsliders
        self.img_size = constants.PG_RESOLUTION
        self.num_channels = constants.NUM_CHANNELS
        self.BATCH_SIZE = constants.BATCH_SIZE
        self.BCE_loss_logits = nn.BCEWithLogitsLoss()
        self.ContentLoss = ContentLoss()
        self.trainEmbed = trainEmbed
        self.step = 6                                            # PGAN 256 (transform_base.py:244-246)
        self.alpha = 0
        if not attrTable:                                        # transform_base.py:248-256
            from .hostutil import SCENE_DEFAULT_TABLE
            from collections import OrderedDict
            self.attrTable = OrderedDict(SCENE_DEFAULT_TABLE)
        else:
            self.attrTable = attrTable
        self.attrList = attrList
        self.attrIdx = self.get_attr_idx()
        if walk_type == 'linear':
            self.walk = WalkLinearZ_free(self.dim_z, self.step, nsliders, self.attrList).to(self.device)
        else:
            raise NotImplementedError('WalkMlpZ3 ("MLP", transform_base.py:279-283) is not on the config-1 path')
        self.optimizer = torch.optim.Adam(self.walk.parameters(), lr=self.lr, betas=(0.5, 0.99))
        self.walk_type = walk_type
        self.N_f = N_f
        self.eps = eps
        self.last_terms = None

    def get_attr_idx(self):
        return [self.attrTable[i] for i in self.attrList]

    def _attr_columns(self):
        t = getattr(self, '_attr_index', None)
        if t is None or t.numel() != len(self.attrIdx):
            t = self._attr_index = torch.tensor(list(self.attrIdx), dtype=torch.long, device=self.device)
        return t

    def get_logits(self, inputs_dict, reshape=True):
        """netG(z) then F.upsample(size=half, mode='bilinear') (:308-321) = the mean of every 2x2 window."""
        z = inputs_dict['z']
        outputs_orig = self.module.netG(z[:, :self.module.netG.code_dim], step=self.step, alpha=self.alpha)
        return _Half.apply(outputs_orig)

    def get_z_new_tensor(self, z, alpha, name=None, trainEmbed=False, index_=None, layers=None):
        return self.walk(z.squeeze(), alpha, name=name, index_=index_)

    def get_reg_preds(self, logit):
        preds = self.regressor(logit).index_select(1, self._attr_columns())
        if len(preds.size()) == 1:
            preds = preds.unsqueeze(1)
        return preds

    def get_alphas(self, alpha_org, alpha_delta):
        """(clamped target, new delta) (:358-364) — the pair train_multi_attr.py:113 unpacks."""
        alpha_target = torch.clamp(alpha_org + alpha_delta, min=0, max=1)
        return alpha_target, alpha_target - alpha_org

    def get_bce_loss(self, pred, y, eps=1e-12):
        return -(y * pred.clamp(min=eps).log() + (1 - y) * (1 - pred).clamp(min=eps).log()).mean()

    def get_reg_loss(self, feed_dict):
        """:366-380 — preds[:, attrIdx].unsqueeze(1) is [B,1,C] against alpha [B,C]: the BCE broadcasts to [B,B,C] before the mean (kept)."""
        logit = feed_dict['logit']
        alpha_gt = feed_dict['alpha'].to(torch.double)
        preds = self.regressor(logit).index_select(1, self._attr_columns())
        preds = preds.unsqueeze(1).to(torch.double)
        return self.get_bce_loss(preds, alpha_gt).mean()

    def get_content_loss(self, org_img, shifted_img):
        losses = self.vgg19.content_losses(org_img, shifted_img)
        return [losses[i] for i in range(4)]

    def get_w_loss(self, feed_dict, no_content_loss=False, no_gan_loss=False):
        """Total loss of optimizeParametersAll (:473-504) without the optimiser step; note the `or` in the weight rule (:496-499)."""
        if not no_gan_loss:
            raise TypeError("the PGGAN graph's GAN term needs the torch.hub discriminator (transform_base.py:476): model_256.Discriminator "
                            'returns a tuple, on which BCEWithLogitsLoss raises in the reference too — run config 1 with --no_gan_loss')
        content_losses = None
        if not no_content_loss:
            content_loss_list = self.get_content_loss(feed_dict['org'], feed_dict['logit'])
            content_losses = sum(content_loss_list) / len(content_loss_list)
        reg_loss = self.get_reg_loss(feed_dict)
        loss = reg_loss if (no_content_loss or no_gan_loss) else 10 * reg_loss
        if not no_content_loss:
            loss = loss + 0.05 * content_losses
        self.last_terms = dict(reg=reg_loss.detach(), cont=None if content_losses is None else content_losses.detach(), gan=None)
        return loss

    def optimizeParametersAll(self, feed_dict, trainEmbed, updateGAN, no_content_loss=False, no_gan_loss=False):
        self.optimizer.zero_grad()
        loss = self.get_w_loss(feed_dict, no_content_loss, no_gan_loss)
        loss.backward()
        dist.average_gradients(self.walk.parameters())
        self.optimizer.step()
        return loss

    optimize_parameters = optimizeParametersAll

    def save_multi_models(self, save_path_w, save_path_gan, trainEmbed=False, updateGAN=False, single_transform_name=None):
        print('Save W and GAN in %s and %s' % (save_path_w, save_path_gan))
        if updateGAN:
            raise NotImplementedError('jointly training the GAN is commented out in the reference (transform_base.py:414-468)')
        if dist.rank() == 0:
            torch.save(self.walk, save_path_w + '_walk_module.ckpt')

    def load_multi_models(self, save_path_w, save_path_gan, trainEmbed=False, updateGAN=False, single_transform_name=None):
        print('Load w in %s' % save_path_w)
        self.walk = torch.load(save_path_w, map_location=self.device, weights_only=False)

    def clip_ims(self, ims):
        return np.uint8(np.clip(((ims + 1) / 2.0) * 256, 0, 255))              # 256, not 255: transform_base.py:595

    def apply_alpha(self, graph_inputs, alpha_to_graph, layers=None, name=None, trainEmbed=False, index_=None, given_w=None):
        """:600-640: (edited image, alpha_org); alpha_delta = alpha_to_graph - alpha_org."""
        with torch.no_grad():
            zs_batch = graph_inputs['z']
            if not torch.is_tensor(zs_batch):
                zs_batch = torch.Tensor(zs_batch).to(self.device)
            out_zs = self.get_logits({'z': zs_batch})
            alpha_to_graph = torch.tensor(np.asarray(alpha_to_graph)).float().to(self.device)
            alpha_org = self.get_reg_preds(out_zs)
            alpha_delta = alpha_to_graph - alpha_org
            z_new = self.get_z_new_tensor(zs_batch, alpha_delta, name, trainEmbed=trainEmbed, index_=index_)
            best_im_out = self.get_logits({'z': z_new})
        return best_im_out, alpha_org

    def vis_multi_image_batch_alphas(self, graph_inputs, filename, alphas_to_graph, alphas_to_target, batch_start, layers=None, name=None,
                                     wgt=False, wmask=False, trainEmbed=False, computeL2=False, given_w=None, index_=None):
        """:714-760: one PNG strip per sample, one panel per requested alpha (``np.uint8(clip((x + 1) / 2 * 255))`` like the reference's
        loop, which does not use clip_ims here); file name ``<filename>_sample<i>[_wgt][_wmask].png``.  Returns the written paths."""
        from PIL import Image
        zs_batch = np.asarray(graph_inputs['z'])
        ims_transformed = []
        for ag in alphas_to_graph:
            best_im_out, alpha_org = self.apply_alpha({'z': torch.Tensor(zs_batch).to(self.device)}, ag, name=name, layers=layers,
                                                      trainEmbed=trainEmbed, given_w=given_w, index_=0)
            x = best_im_out.detach().cpu().numpy()
            ims_transformed.append(np.uint8(np.clip(((x + 1) / 2.0) * 255, 0, 255)))
        written = []
        for ii in range(zs_batch.shape[0]):
            strip = np.concatenate([x[ii].transpose(1, 2, 0) for x in ims_transformed], axis=1)
            path = filename + '_sample{}'.format(ii + batch_start) + ('_wgt' if wgt else '') + ('_wmask' if wmask else '') + '.png'
            Image.fromarray(strip).save(path)
            written.append(path)
        return written

    def vis_image_batch(self, graph_inputs, filename, batch_start, wgt=False, wmask=False, num_panels=7):
        raise NotImplementedError('Subclass should implement vis_image_batch')


def walk_training_step(graph, zs_batch, alpha_delta, no_content_loss=False, no_gan_loss=True):
    """One z-walk training step in the order of train_multi_attr.py:93-140 on the PGGAN graph's own methods.  (The shipped drivers call
    ``get_w`` / ``get_w_new_tensor``, which only the StyleGAN2 graph has; the PGGAN graph is driven method by method — this is that
    sequence, used by the tests and by anyone running config 1.)  Returns (loss, x0, x1, alpha_org, alpha_target)."""
    z = zs_batch if torch.is_tensor(zs_batch) else torch.Tensor(zs_batch).to(graph.device)
    with torch.no_grad():
        out_zs = graph.get_logits({'z': z})
        alpha_org = graph.get_reg_preds(out_zs)
    ad = alpha_delta if torch.is_tensor(alpha_delta) else torch.tensor(np.asarray(alpha_delta)).float().to(graph.device)
    alpha_target, alpha_delta_new = graph.get_alphas(alpha_org, ad)
    z_new = graph.get_z_new_tensor(z, alpha_delta_new)
    transformed_output = graph.get_logits({'z': z_new})
    feed_dict = {'z': z_new, 'org': out_zs, 'logit': transformed_output, 'alpha': alpha_target}
    loss = graph.optimizeParametersAll(feed_dict, trainEmbed=False, updateGAN=False, no_content_loss=no_content_loss, no_gan_loss=no_gan_loss)
    return loss, out_zs, transformed_output, alpha_org, alpha_target


class _Half(torch.autograd.Function):
    """F.upsample(x, size=(H // 2, W // 2), mode='bilinear') for even H, W (align_corners=False): the 2x2 mean; backward = 0.25 * nearest."""

    @staticmethod
    def forward(ctx, x):
        return K.pool2x2(x.detach(), 0.25)

    @staticmethod
    def backward(ctx, g):
        return K.upsample2x_nearest(g.contiguous(), 0.25)


class PixelTransform(TransformGraph):
    def __init__(self, *args, **kwargs):
        TransformGraph.__init__(self, *args, **kwargs)


def _make_graph(name, op_cls):
    def __init__(self, lr=0.001, walk_type='NNz', loss='l2', eps=1.41, N_f=4, **kwargs):
        nsliders = 1
        self.walk_type = walk_type
        self.num_channels = constants.NUM_CHANNELS
        self.Nsliders = nsliders
        self.img_size = constants.PG_RESOLUTION
        PixelTransform.__init__(self, lr, walk_type, nsliders, loss, eps, N_f, **kwargs)
        op_cls.__init__(self)

    def vis_image_batch(self, graph_inputs, filename, batch_start, wgt=False, wmask=False, num_panels=7, max_alpha=None, min_alpha=None,
                        N_attr=40):
        zs_batch = graph_inputs['z']
        alphas = np.linspace(min_alpha, max_alpha, num_panels) if (max_alpha is not None and min_alpha is not None) else np.linspace(0, 1, num_panels)
        return [self.scale_test_alpha_for_graph(a, zs_batch) for a in alphas], list(alphas)

    return type(name, (PixelTransform, op_cls), {'__init__': __init__, 'vis_image_batch': vis_image_batch})


SceneGraph = _make_graph('SceneGraph', SceneTransform)
faceGraph = _make_graph('faceGraph', FaceTransform)
