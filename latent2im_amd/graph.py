"""Host-side mirror of the reference Graph API for the StyleGAN2 walk-training path.

Same class / method names, argument meaning and error behaviour as the reference so that ``train.py`` reads the same:

    graphs/__init__.py:3-22                      find_model_using_name
    graphs/transform_graph_scene.py:5-125        get_transform_graphs -> [SceneGraph, faceGraph]
    graphs/stylegan_v2_real/transform_base.py    TransformGraph (:246-549), PixelTransform (:901-903),
                                                 WalkLinearMultiW (:140-165), ContentLoss/Normalization (:44-63)
    utils/transforms.py:634-735                  FaceTransform / SceneTransform alpha samplers

Underneath, every network is the frozen HIP implementation of this package (generator.py, regressor.py, perceptual.py,
discriminator.py); nothing here runs on the CPU and nothing imports the oracle.  Differences from the reference,
all output-equivalent: frozen networks carry no autograd state (the reference keeps requires_grad=True everywhere and
back-propagates weight gradients it never uses); the first generator/regressor pass therefore builds
no graph (A.4); the VGG prefix is evaluated once per image instead of once per tap.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import constants, dist, synth
from .discriminator import Discriminator
from .generator import Generator
from .perceptual import VGG19Prefix
from .regressor import ResNet50


# ----------------------------------------------------------------------------------------------------------------
# walk modules
# ----------------------------------------------------------------------------------------------------------------
class WalkLinearMultiW(nn.Module):
    """Input-independent linear walk in W+ (transform_base.py:140-165): w_i + alpha @ W[:, i, :].
    ``self.w`` [n_attr, 2*(step+1), dim_z] ~ N(0, 0.02) from the global numpy RNG, exactly like the reference."""

    def __init__(self, dim_z, step, Nsliders, attrList):
        super().__init__()
        self.dim_z = dim_z
        self.step = step
        self.w = nn.Parameter(torch.Tensor(np.random.normal(0.0, 0.02, [len(attrList), (self.step + 1) * 2, self.dim_z])))

    def forward(self, input, alpha, layers=None, name=None, index_=None):
        al = alpha.to(self.w.device)
        dirs = torch.matmul(al, self.w.permute(1, 0, 2))                 # [n_latent, B, dim_z] in one batched GEMM
        if layers is None and len(input) == dirs.shape[0] and all(t is input[0] for t in input):
            # get_w hands the SAME tensor n_latent times (transform_base.py:372-378): one broadcast add instead of n_latent, the list entries are
            # views of its result (same values, same list-of-[B,512] contract)
            return list((input[0].unsqueeze(0) + dirs).unbind(0))
        w_transformed = []
        for i in range(len(input)):
            if layers is None or i in layers:
                w_transformed.append(input[i] + dirs[i])
            else:
                w_transformed.append(input[i])
        return w_transformed


# pickles written by save_multi_models must resolve in the reference's vis_w.py (transform_base.py:499-509)
WalkLinearMultiW.__module__ = 'graphs.stylegan_v2_real.transform_base'


class WalkMlpMultiW(nn.Module):
    """Input-dependent MLP walk in W+ (transform_base.py:168-204; "an unused setting in the paper", built only when
    ``is_mlp`` is switched on by hand): w_i + alpha[:, :1] * MLP(w_i), MLP = 512 -> 1024 -> 1024 -> 512 with LeakyReLU(0.2).
    With ``layers`` given the reference calls ``self.linear(input[i], 1)``, which raises TypeError: kept."""

    def __init__(self, dim_z, step, Nsliders, attrList):
        super().__init__()
        self.dim_z = dim_z
        self.step = step
        self.Nsliders = Nsliders
        self.linear = nn.Sequential(nn.Linear(dim_z, 2 * dim_z), nn.LeakyReLU(0.2, True),
                                    nn.Linear(2 * dim_z, 2 * dim_z), nn.LeakyReLU(0.2, True),
                                    nn.Linear(2 * dim_z, dim_z))

    def forward(self, input, alpha, layers=None, name=None, index_=None):
        al = torch.unsqueeze(alpha[:, 0], 1).to(self.linear[0].weight.device)
        if layers is None:
            same = all(t is input[0] for t in input)               # get_w hands the SAME tensor n_latent times: one MLP pass
            step0 = self.linear(input[0]) if same else None
            return [input[i] + al * (step0 if same else self.linear(input[i])) for i in range(len(input))]
        out = []
        for i in range(len(input)):
            out.append(input[i] + al * self.linear(input[i], 1) if i in layers else input[i])
        return out


class WalkNonLinearW(nn.Module):
    """MLP walk conditioned on an embedding of alpha (transform_base.py:207-243), chosen by ``--walk_type NN*``:
    e = Linear(10, 256)(alpha[:, :1] repeated 10x); d = MLP([e, w_i]); w_i + d / ||d|| (no normalisation when ``layers``
    is given).  NOTE the argument order (input, name, alpha, index_, layers): the reference's get_w_new_tensor calls every
    walk as ``walk(multi_ws, alpha=..., layers=...)`` (transform_base.py:380-386), so with this walk it raises TypeError
    (missing ``name`` / ``index_``) — the same happens here; call the module directly to use it."""

    def __init__(self, dim_z, step, Nsliders, attrList):
        super().__init__()
        self.dim_z = dim_z
        self.step = step
        self.Nsliders = Nsliders
        self.embed = nn.Linear(10, dim_z // 2)
        self.linear = nn.Sequential(nn.Linear(dim_z // 2 + dim_z, 2 * dim_z), nn.LeakyReLU(0.2, True),
                                    nn.Linear(2 * dim_z, dim_z))

    def forward(self, input, name, alpha, index_, layers=None):
        al = torch.unsqueeze(alpha[:, 0], 1).to(self.embed.weight.device)
        out = self.embed(al.repeat(1, 10))
        w_transformed = []
        for i in range(len(input)):
            if layers is None:
                out2 = self.linear(torch.cat([out, input[i]], 1))
                w_transformed.append(input[i] + out2 / torch.norm(out2, dim=1, keepdim=True))
            elif i in layers:
                w_transformed.append(input[i] + self.linear(torch.cat([out, input[i]], 1)))
            else:
                w_transformed.append(input[i])
        return w_transformed


WalkMlpMultiW.__module__ = WalkNonLinearW.__module__ = 'graphs.stylegan_v2_real.transform_base'


class ContentLoss(nn.Module):
    """transform_base.py:57-63."""

    def forward(self, org, shifted):
        self.loss = torch.nn.functional.mse_loss(org.detach(), shifted)
        return self.loss


# ----------------------------------------------------------------------------------------------------------------
# network holders
# ----------------------------------------------------------------------------------------------------------------
class StyleGAN:
    """Attribute contract of stylegan2.StyleGAN (stylegan2.py:19-31): ``netG`` / ``netD``.  The reference also builds
    ``g_running`` and two Adam optimisers that the walk path never touches (updateGAN raises, train.py:40-41)."""

    def __init__(self, netG, netD):
        self.netG, self.netD = netG, netD


def _to_numpy_state(sd):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}


def _checkpoint_or_synthetic(what, path):
    """True: load ``path``.  False: synthetic weights were requested explicitly.  Otherwise raise: a mistyped checkpoint path must
    not train a walk against random frozen networks without anyone noticing."""
    if path and os.path.isfile(path):
        return True
    if constants.ALLOW_SYNTHETIC_WEIGHTS:
        return False
    raise FileNotFoundError('%s checkpoint %r does not exist (graphs.stylegan_v2_real.constants); pass --synthetic_weights / set '
                            'L2I_SYNTHETIC_WEIGHTS=1 to run on seeded random-init weights of the same architecture instead' % (what, path))


def load_networks(resolution, device, need_vgg=True, need_d=True):
    """Frozen nets from the checkpoint paths of ``constants`` (reference transform_base.py:522-549).  Deterministic synthetic
    weights of the same architecture (latent2im_amd.synth) are used only on explicit request (constants.ALLOW_SYNTHETIC_WEIGHTS);
    the choice per network is returned as the last element and logged by the drivers."""
    src = {}
    if _checkpoint_or_synthetic('generator (g_path)', constants.g_path):
        g_state = _to_numpy_state(torch.load(constants.g_path, map_location='cpu')['g_ema'])
        src['G'] = constants.g_path
    else:
        g_state = synth.generator_state(resolution, seed=constants.SYNTH_SEED_G, noise_strength=constants.SYNTH_NOISE_STRENGTH)
        src['G'] = 'synthetic(seed=%d%s)' % (constants.SYNTH_SEED_G, ', noise_strength=%g' % constants.SYNTH_NOISE_STRENGTH if constants.SYNTH_NOISE_STRENGTH else '')
    if _checkpoint_or_synthetic('regressor (reg_path)', constants.reg_path):
        r_state = _to_numpy_state(torch.load(constants.reg_path, map_location='cpu')['model'])
        src['R'] = constants.reg_path
    else:
        r_state = synth.resnet50_state(seed=constants.SYNTH_SEED_R)
        src['R'] = 'synthetic(seed=%d)' % constants.SYNTH_SEED_R
    from . import conv
    if conv.PRECISION in conv.H8_PRECISIONS:                           # the 16-bit path (BASELINE config 5): bf16 h8 feature maps, one bf16 MFMA per MAC (nets16.py)
        from . import nets16
        GenCls, RegCls, VggCls, DCls = nets16.Generator, nets16.ResNet50, nets16.VGG19Prefix, nets16.Discriminator
    else:
        GenCls, RegCls, VggCls, DCls = Generator, ResNet50, VGG19Prefix, Discriminator
    src['precision'] = conv.PRECISION
    netG = GenCls(g_state, resolution, device=device)
    reg = RegCls(r_state, device=device)
    vgg = netD = None
    if need_vgg:
        if _checkpoint_or_synthetic('VGG-19 (vgg_path; the reference downloads torchvision weights)', constants.vgg_path):
            v_state = _to_numpy_state(torch.load(constants.vgg_path, map_location='cpu'))
            v_state = {k.replace('features.', ''): v for k, v in v_state.items()}
            src['V'] = constants.vgg_path
        else:
            v_state = synth.vgg19_prefix_state(seed=constants.SYNTH_SEED_V)
            src['V'] = 'synthetic(seed=%d)' % constants.SYNTH_SEED_V
        vgg = VggCls(v_state, device=device)
    if need_d:
        # the reference's netD is ALWAYS freshly initialised (never loaded): seeded here for reproducibility
        netD = DCls(synth.discriminator_state(resolution, seed=constants.SYNTH_SEED_D), resolution, device=device)
        src['D'] = 'random-init(seed=%d)' % constants.SYNTH_SEED_D
    if conv.PRECISION == 'f16':
        # gradient scales of the fp16 path (nets16.loss_scale_for): static exponents for this resolution and the PER-RANK batch — every rank's losses
        # are means over its own shard — times one dynamic factor on the device; one scaler per graph, carried by its four networks
        from . import optim
        per_rank = max(constants.BATCH_SIZE // dist.world_size(), 1)
        scaler = optim.LossScaler(nets16.loss_scale_for(resolution, per_rank), device, growth_interval=constants.LOSS_SCALE_GROWTH_INTERVAL)
        nets16.attach_scaler((netG, netD, reg, vgg), scaler)
        src['loss_scale_log2'] = dict(scaler.log2)
    return netG, netD, reg, vgg, src


def attribute_change_bucket(pred, org):
    """Bucket index per sample (transform_base.py:722-738): 0 if |pred - org| <= 0.3, 1 if <= 0.6, 2 if <= 1, else 3
    (dropped).  The comparison is made on the float32 difference against the float32 thresholds, which is what numpy does
    for ``np.abs(float32 - float32) <= 0.3`` under NumPy >= 2 (a python float is a weak scalar; NumPy 1.x promoted the
    scalar comparison to float64, which differs only for a difference that rounds exactly onto a threshold)."""
    d = np.abs(np.asarray(pred, dtype=np.float32) - np.asarray(org, dtype=np.float32))
    return np.where(d <= np.float32(0.3), 0, np.where(d <= np.float32(0.6), 1, np.where(d <= np.float32(1), 2, 3))).astype(np.int64)


# ----------------------------------------------------------------------------------------------------------------
# TransformGraph
# ----------------------------------------------------------------------------------------------------------------
class _RegBceFn(torch.autograd.Function):
    """loss = get_bce_loss(fc(feat)[:, cols], target.double()).mean() (transform_base.py:416-424), forward and d loss / d feat, one launch (l2i_reg_bce_f32)."""

    @staticmethod
    def forward(ctx, feat, fc_w, fc_b, cols, target):
        from . import _lib
        feat = feat.detach().contiguous()
        assert feat.dtype == torch.float32 and feat.is_cuda and fc_w.shape[1] == feat.shape[1] and target.dtype in (torch.float32, torch.float64)
        B, F = feat.shape
        K = int(cols.numel())
        target = target.detach().contiguous()
        assert tuple(target.shape) == (B, K), (target.shape, B, K)
        loss = torch.empty(1, dtype=torch.float64, device=feat.device)
        preds = torch.empty(B, K, dtype=torch.float32, device=feat.device)
        g = torch.empty_like(feat)
        _lib.check(_lib.load().l2i_reg_bce_f32(_lib.ptr(loss), _lib.fptr(preds), _lib.fptr(g), _lib.fptr(feat), _lib.fptr(fc_w), _lib.fptr(fc_b), _lib.ptr(cols),
                                              _lib.ptr(target), int(target.dtype == torch.float64), B, F, K, 1e-12, _lib.stream_ptr()), 'l2i_reg_bce_f32')
        ctx.save_for_backward(g)
        ctx.preds = preds
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        return torch.mul(g, g_loss), None, None, None, None          # (fp32 tensor x 0-dim float64 tensor -> fp32: one launch)


class TransformGraph:
    def __init__(self, lr, walk_type, nsliders, loss_type, eps, N_f, trainEmbed, attrList, attrTable, layers, stylegan_opts,
                 nets=None):
        assert (loss_type in ['l2', 'lpips']), 'unimplemented loss'
        if not torch.cuda.is_available():
            raise RuntimeError('latent2im_amd runs on an MI355X (ROCm) device only: no GPU is visible and there is no CPU path')
        self.lr = lr
        self.useGPU = constants.useGPU
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.img_size = constants.resolution
        if nets is None:
            nets = load_networks(self.img_size, self.device)
        netG, netD, self.regressor, self.vgg19, self.weight_sources = nets
        self.module = StyleGAN(netG, netD)
        self.reg_optmizer = None

        self.attrTable = attrTable
        self.attrList = attrList
        self.attrIdx = self.get_attr_idx()

        self.dim_z = constants.DIM_Z
        self.Nsliders = nsliders
        self.num_channels = constants.NUM_CHANNELS
        self.BATCH_SIZE = constants.BATCH_SIZE
        self.BCE_loss_logits = nn.BCEWithLogitsLoss()
        self.ContentLoss = ContentLoss()
        self.trainEmbed = trainEmbed

        # the reference hard-codes step = 6 (256^2, transform_base.py:285); generalised: n_latent = 2*(step+1)
        self.step = int(math.log2(self.img_size)) - 2
        self.stylegan_opts = stylegan_opts
        self.layers = layers
        self.is_mlp = constants.WALK_IS_MLP

        if walk_type == 'linear':
            if self.trainEmbed:
                raise NotImplementedError('WalkEmbed is unused in the paper (transform_base.py:298-303) and out of scope')
            if stylegan_opts.latent == 'z':
                raise NotImplementedError('Not implemented setting of linear transformation for z')
            elif stylegan_opts.latent == 'w':
                cls = WalkMlpMultiW if constants.WALK_IS_MLP else WalkLinearMultiW    # reference: is_mlp hard-coded False (:290)
                self.walk = cls(self.dim_z, self.step, nsliders, self.attrList).to(self.device)
            else:
                raise NotImplementedError('Not implemented latent walk type:' '{}'.format(stylegan_opts.latent))
        elif 'NN' in walk_type:
            self.walk = WalkNonLinearW(self.dim_z, self.step, nsliders, self.attrList).to(self.device)
        else:
            raise NotImplementedError('unknown walk_type %r' % (walk_type,))

        # fp16 elements: the same Adam with GradScaler semantics on the device — a step whose gradient holds an inf / NaN is skipped and the dynamic
        # loss scale halves (optim.py); every other precision: torch's own, as in the reference (transform_base.py:329-331)
        self.loss_scaler = getattr(netG, 'scaler', None)
        if self.loss_scaler is not None:
            from . import optim
            self.optimizers = optim.GuardedAdam(self.walk.parameters(), lr=self.lr, betas=(0.5, 0.99), scaler=self.loss_scaler)
        else:
            self.optimizers = torch.optim.Adam(self.walk.parameters(), lr=self.lr, betas=(0.5, 0.99))
        self.walk_type = walk_type
        self.last_terms = None

    # -- reference helpers ---------------------------------------------------------------------------------------
    def get_attr_idx(self):
        return [self.attrTable[i] for i in self.attrList]

    def get_logits(self, inputs_dict, reshape=True):
        if self.stylegan_opts.latent == 'z':
            raise NameError("latent: the reference's Generator.forward fails for input_is_latent=False (networks.py:471-494)")
        w = inputs_dict['w']
        if isinstance(w, (list, tuple)):
            w = torch.stack(list(w)).transpose(0, 1)
        else:
            w = w.transpose(0, 1)
        outputs_orig, _ = self.module.netG(w.contiguous(), input_is_latent=True)
        return outputs_orig

    def get_w(self, z, is_single=False):
        w = self.module.netG.style(z)
        if is_single:
            return [w]
        return [w] * (self.step + 1) * 2

    def get_w_new_tensor(self, multi_ws, alpha, layers=None, name=None, trainEmbed=False, index_=None):
        if layers is not None:
            layers = [int(l) for l in layers]                 # the reference passes strings through
        return self.walk(multi_ws, alpha=alpha, layers=layers)

    def _attr_columns(self):
        """attrIdx as a device index tensor, built once: indexing with the python list would copy it host -> device on every call
        (a synchronous copy, and not permitted while the step is being captured into a hipGraph)."""
        t = getattr(self, '_attr_index', None)
        if t is None or t.numel() != len(self.attrIdx) or t.device != self.device:
            t = self._attr_index = torch.tensor(list(self.attrIdx), dtype=torch.long, device=self.device)
        return t

    def get_reg_preds(self, logit):
        preds = self.regressor(logit).index_select(1, self._attr_columns())        # integer column select: bit-exact
        if len(preds.size()) == 1:
            preds = preds.unsqueeze(1)
        return preds

    def get_alphas(self, alpha_org, alpha_target):
        """train.py flow (transform_base.py:405-408): epsilon = target - org."""
        return alpha_target - alpha_org

    def get_alphas_clamped(self, alpha_org, alpha_delta):
        """train_multi_attr.py:113 flow (graphs/pggan/transform_base.py:358-364): (clamped target, new delta)."""
        alpha_target = torch.clamp(alpha_org + alpha_delta, min=0, max=1)
        return alpha_target, alpha_target - alpha_org

    def get_bce_loss(self, pred, y, eps=1e-12):
        return -(y * pred.clamp(min=eps).log() + (1 - y) * (1 - pred).clamp(min=eps).log()).mean()

    def get_reg_loss(self, feed_dict):
        logit = feed_dict['logit']
        reg = self.regressor
        if constants.FUSED_REG_LOSS and hasattr(reg, 'features') and len(self.attrIdx) <= 64:
            # [r6] fc + column select + the float64 BCE and everything autograd would run backwards through them as ONE launch each way (csrc/l2i_loss.hip):
            # ~40 launches of a few microseconds between the regressor's last conv and its first gradient conv otherwise
            return _RegBceFn.apply(reg.features(logit), reg.fc_w, reg.fc_b, self._attr_columns(), feed_dict['alpha'])
        alpha_gt = feed_dict['alpha'].to(torch.double)
        preds = reg(logit).index_select(1, self._attr_columns())
        return self.get_bce_loss(preds, alpha_gt).mean()

    def prefetch_content_taps(self, org_img):
        """[r6] Start the VGG-19 prefix of the ORIGINAL image (transform_base.py:444-454: ``target = model(org).detach()``, no gradient) on the content
        branch's stream as soon as that image exists, i.e. while the regressor reads it and the second generator pass runs: 6 ms (fp32) of large
        matrix-bound launches that fill the chip where the regressor's tail and the generator's 4^2 .. 64^2 layers do not.  ``get_content_loss`` picks
        the taps up when it is handed the same tensor; the values are the same launches' outputs, issued earlier.  The drivers call this (capture.forward,
        trainer.train_step) when the content loss is on; without the call nothing changes."""
        # Measured (tools/ab/r06_prefetch_taps.sh, alternating runs on one box): the 16-bit path gains 0.8 % (c5 35.43 / 35.24 -> 35.08 / 34.99 ms per step: its
        # launches are short and the chip has holes to fill); the fp32 path LOSES 0.4 % (c3 99.27 / 99.54 -> 99.82 / 99.94: its launches saturate the chip
        # and the early taps only lengthen the critical chain) — so: on for the 16-bit path, off for fp32 (L2I_PREFETCH_TAPS=1 / 0 force it).
        from . import conv
        on = constants.PREFETCH_CONTENT_TAPS if constants.PREFETCH_CONTENT_TAPS is not None else conv.PRECISION in conv.H8_PRECISIONS
        if self.vgg19 is None or not constants.CONCURRENT_LOSS_BRANCHES or not on:
            return
        cur = torch.cuda.current_stream()
        side = self._side_streams()[1]
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            self._org_taps = (org_img, self.vgg19.org_taps(org_img.detach()))

    def get_content_loss(self, org_img, shifted_img):
        """List of four scalars (transform_base.py:426-454)."""
        pre = getattr(self, '_org_taps', None)
        self._org_taps = None
        taps = pre[1] if (pre is not None and pre[0] is org_img) else None
        losses = self.vgg19.content_losses(org_img, shifted_img, org_taps=taps)
        return [losses[i] for i in range(4)]

    # -- loss / optimiser ----------------------------------------------------------------------------------------
    def _side_streams(self):
        if getattr(self, '_streams', None) is None:
            self._streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return self._streams

    def get_w_loss(self, feed_dict, no_content_loss=False, no_gan_loss=False):
        """Total walk loss of optimizeParametersAll (transform_base.py:459-486) without the optimiser step."""
        logit = feed_dict['logit']
        gan_loss = content_losses = None
        # The three loss branches (discriminator, VGG content, regressor) are independent forward+backward chains over
        # the same image: each runs on its own HIP stream (autograd replays every node's backward on its forward
        # stream), so the tail of one network's launches overlaps the next one's instead of leaving CUs idle.
        cur = torch.cuda.current_stream()
        side = self._side_streams() if constants.CONCURRENT_LOSS_BRANCHES else (cur, cur)
        if not no_gan_loss:
            side[0].wait_stream(cur)
            with torch.cuda.stream(side[0]):
                D_fake_result = self.module.netD(logit)
                gan_loss = self.BCE_loss_logits(D_fake_result, torch.ones_like(D_fake_result))
        if not no_content_loss:
            side[1].wait_stream(cur)
            with torch.cuda.stream(side[1]):
                content_loss_list = self.get_content_loss(feed_dict['org'], feed_dict['logit'])
                content_losses = sum(content_loss_list) / len(content_loss_list)
        reg_loss = self.get_reg_loss(feed_dict)
        if side[0] is not cur:
            cur.wait_stream(side[0])
            cur.wait_stream(side[1])
        loss = reg_loss if (no_content_loss and no_gan_loss) else 10 * reg_loss
        if not no_content_loss:
            loss = loss + 0.05 * content_losses
        if not no_gan_loss:
            loss = loss + 0.05 * gan_loss
        self.last_terms = dict(reg=reg_loss.detach(), cont=None if content_losses is None else content_losses.detach(),
                               gan=None if gan_loss is None else gan_loss.detach())
        return loss

    def optimizeParametersAll(self, feed_dict, trainEmbed, updateGAN, no_content_loss=False, no_gan_loss=False):
        self.optimizers.zero_grad()
        loss = self.get_w_loss(feed_dict, no_content_loss, no_gan_loss)
        loss.backward()
        dist.average_gradients(self.walk.parameters())        # data parallel: one RCCL all-reduce of <= 184 KB
        self.optimizers.step()
        return loss

    optimize_parameters = optimizeParametersAll              # BASELINE.json spelling

    # -- checkpoints ---------------------------------------------------------------------------------------------
    def save_multi_models(self, save_path_w, save_path_gan, trainEmbed=False, updateGAN=False, single_transform_name=None):
        print('Save W and GAN in %s and %s' % (save_path_w, save_path_gan))
        if updateGAN:
            raise NotImplementedError('jointly training the GAN is not implemented in the reference either (train.py:40-41)')
        if dist.rank() == 0:
            torch.save(self.walk, save_path_w + '_walk_module.ckpt')

    def load_multi_models(self, save_path_w, save_path_gan, trainEmbed=False, updateGAN=False, single_transform_name=None):
        print('Load w in %s' % save_path_w)
        self.walk = torch.load(save_path_w, map_location=self.device, weights_only=False)

    def load_multi_models_from_single(self, save_path_ws, save_path_gan, trainEmbed=False, updateGAN=False,
                                      single_transform_name=None, index=None):
        for i in range(len(save_path_ws)):
            walk_ckpt = torch.load(save_path_ws[i], map_location=self.device, weights_only=False)
            with torch.no_grad():
                self.walk.w[index[i]] = walk_ckpt.w[0]

    # -- inference ("next" row f-1: vis_w.py) --------------------------------------------------------------------
    def clip_ims(self, ims):
        return np.uint8(np.clip(((ims + 1) / 2.0) * 255, 0, 255))

    def apply_alpha(self, graph_inputs, alpha_to_graph, layers=None, name=None, trainEmbed=False, index_=None, given_w=None):
        """transform_base.py:554-603 (w branch): returns (edited image, alpha_org, original image)."""
        with torch.no_grad():
            zs_batch = graph_inputs['z']
            if not torch.is_tensor(zs_batch):
                zs_batch = torch.Tensor(zs_batch).to(self.device)
            latent_w = given_w if given_w is not None else self.get_w(zs_batch)
            out_zs = self.get_logits({'w': latent_w})
            alpha_org = self.get_reg_preds(out_zs)
            alpha_delta = self.get_alphas(alpha_org, torch.Tensor(np.asarray(alpha_to_graph)).to(self.device))
            if index_ is not None:
                if len(self.attrIdx) == len(self.attrTable):
                    alpha_delta[:, index_] = torch.Tensor(np.asarray(alpha_to_graph)).to(self.device) - alpha_org[:, index_]
                else:
                    i = self.attrIdx.index(index_)
                    alpha_delta[:, i] = torch.Tensor(np.asarray(alpha_to_graph)[:, 0]).to(self.device) - alpha_org[:, i]
            latent_w_new = self.get_w_new_tensor(latent_w, alpha_delta, layers=layers, name=name, trainEmbed=trainEmbed,
                                                 index_=index_)
            best_im_out = self.get_logits({'w': latent_w_new})
        return best_im_out, alpha_org, out_zs

    def vis_multi_image_batch_alphas(self, graph_inputs, filename, alphas_to_graph, alphas_to_target, batch_start,
                                     layers=None, name=None, wgt=False, wmask=False, trainEmbed=False, computeL2=False,
                                     given_w=None, index_=None):
        """transform_base.py:606-659: one PNG strip per sample, one panel per requested alpha; file name
        ``<filename>_sample<i>[_wgt]_<alpha_org>.png``.  Returns the list of written paths."""
        from PIL import Image
        zs_batch = graph_inputs['z']
        ims_transformed = []
        for ag in alphas_to_graph:
            best_im_out, alpha_org, out_zs = self.apply_alpha({'z': torch.Tensor(zs_batch).to(self.device)}, ag, name=name,
                                                              layers=layers, trainEmbed=trainEmbed, given_w=given_w, index_=index_)
            ims_transformed.append(self.clip_ims(best_im_out.detach().cpu().numpy()))
        written = []
        for ii in range(zs_batch.shape[0]):
            a = alpha_org[ii, index_].item() if (index_ is not None and len(self.attrList) > 1) else alpha_org[ii].reshape(-1)[0].item()
            strip = np.concatenate([x[ii].transpose(1, 2, 0) for x in ims_transformed], axis=1)      # panels side by side
            path = filename + '_sample{}'.format(ii + batch_start) + ('_wgt' if wgt else '') + '_%.2f.png' % a
            print('Save in ', path)
            Image.fromarray(strip).save(path)
            written.append(path)
        return written

    def vis_multi_image_batch_alphas_compute_multi_attr(self, graph_inputs, filename, alphas_to_graph, alphas_to_target,
                                                        batch_start, layers=None, name=None, wgt=False, wmask=False,
                                                        trainEmbed=False, computeL2=False, given_w=None, index_=None):
        """Attribute-preservation half of eval.py (transform_base.py:675-767): for every requested alpha edit the batch,
        read all 40 regressor outputs of the edited and the original image, and sort every sample into one of three
        buckets by how far the TARGET attribute moved: |d| <= 0.3, <= 0.6, <= 1 (samples that moved further are dropped).
        Returns (multi_attr, attri_org, imgs, orgs): four lists of three lists (edited attrs [40], original attrs [40],
        uint8 edited image [3,R,R], uint8 original image)."""
        zs_batch = graph_inputs['z']
        multi_attr, attri_org, imgs, orgs = [[], [], []], [[], [], []], [[], [], []], [[], [], []]
        index_list = [index_] if type(index_) == int else index_
        with torch.no_grad():
            for ag1, at1 in zip(alphas_to_graph, alphas_to_target):
                best_im_out, alpha_org, out_zs = self.apply_alpha({'z': torch.Tensor(zs_batch).to(self.device)}, ag1, name=name,
                                                                  layers=layers, trainEmbed=trainEmbed, given_w=given_w,
                                                                  index_=index_)
                pred_attr = self.regressor(best_im_out).detach().cpu().numpy()       # [N, 40]
                org = self.regressor(out_zs).detach().cpu().numpy()
                best_im_out = self.clip_ims(best_im_out.detach().cpu().numpy())
                out_zs = self.clip_ims(out_zs.cpu().numpy())
                bucket = attribute_change_bucket(pred_attr[:, index_list[0]], org[:, index_list[0]])
                for i in range(pred_attr.shape[0]):
                    k = int(bucket[i])
                    if k < 3:
                        multi_attr[k].append(pred_attr[i])
                        attri_org[k].append(org[i])
                        imgs[k].append(best_im_out[i])
                        orgs[k].append(out_zs[i])
        return multi_attr, attri_org, imgs, orgs

    def vis_image_batch(self, graph_inputs, filename, batch_start, wgt=False, wmask=False, num_panels=7):
        raise NotImplementedError('Subclass should implement vis_image_batch')


class PixelTransform(TransformGraph):
    def __init__(self, *args, **kwargs):
        TransformGraph.__init__(self, *args, **kwargs)


# ----------------------------------------------------------------------------------------------------------------
# alpha samplers (utils/transforms.py:634-735 and the overrides in stylegan_v2_real/transform_op.py:65-77)
# ----------------------------------------------------------------------------------------------------------------
class FaceTransform:
    def __init__(self, atrr_name='Black_Hair'):
        self.atrr_name = atrr_name
        self.alpha_original = 1
        self.num_panel = 6
        self.embed_alpha_max = 1
        self.embedding_alpha = np.linspace(0.0, 1.0, self.num_panel)
        self.alpha_max = 1

    def get_train_alpha(self, zs_batch, N_attr=40, trainEmbed=False):
        """One target alpha ~ U(0,1)^N_attr per step, shared by the whole batch (global numpy RNG)."""
        batch_size = zs_batch.shape[0]
        if trainEmbed:
            index_ = np.random.choice(self.num_panel)
            alpha_val = self.embedding_alpha[index_]
            return np.ones((batch_size, self.Nsliders)) * (alpha_val / self.embed_alpha_max), alpha_val, index_
        alpha_val = np.random.uniform(0, 1, N_attr)
        return np.ones((batch_size, self.Nsliders)) * alpha_val, alpha_val, None

    def scale_test_alpha_for_graph(self, alpha, zs_batch, **kwargs):
        return alpha * np.ones((zs_batch.shape[0], self.Nsliders))

    def test_alphas(self):
        return np.linspace(0, 1, 9)

    def vis_alphas(self, num_panels):
        return np.linspace(0, 1, num_panels)


class SceneTransform:
    def __init__(self):
        self.alpha_max = 1
        self.num_panel = 6
        self.embed_alpha_max = 1
        self.embedding_alpha = np.linspace(0.0, 1.0, self.num_panel)

    def get_train_alpha(self, zs_batch, N_attr=40, trainEmbed=False):
        """One delta ~ U(-1,1)^N_attr per step, shared by the whole batch."""
        batch_size = zs_batch.shape[0]
        if trainEmbed:
            index_ = np.random.choice(self.num_panel)
            alpha_val = self.embedding_alpha[index_]
            return np.ones((batch_size, self.Nsliders)) * alpha_val, alpha_val, index_
        alpha_val = np.random.uniform(-1, 1, N_attr)
        return np.ones((batch_size, N_attr)) * alpha_val, alpha_val, None

    def scale_test_alpha_for_graph(self, alpha, zs_batch, **kwargs):
        return alpha * np.ones((zs_batch.shape[0], self.Nsliders))

    def test_alphas(self):
        return np.linspace(0, 1, 10)

    def vis_alphas(self, num_panels):
        return np.linspace(0, 1, num_panels)


# ----------------------------------------------------------------------------------------------------------------
# plugin lookup
# ----------------------------------------------------------------------------------------------------------------
def _make_graph(name, op_cls):
    def __init__(self, lr=0.001, walk_type='NNz', loss='l2', eps=1.41, N_f=4, **kwargs):
        nsliders = 1
        self.walk_type = walk_type
        self.num_channels = constants.NUM_CHANNELS
        self.Nsliders = nsliders
        self.img_size = constants.resolution
        PixelTransform.__init__(self, lr, walk_type, nsliders, loss, eps, N_f, **kwargs)
        op_cls.__init__(self)

    def vis_image_batch(self, graph_inputs, filename, batch_start, wgt=False, wmask=False, num_panels=7, max_alpha=None,
                        min_alpha=None, N_attr=40):
        zs_batch = graph_inputs['z']
        if max_alpha is not None and min_alpha is not None:
            alphas = np.linspace(min_alpha, max_alpha, num_panels)
        else:
            alphas = np.linspace(0, 1, num_panels)
        return [self.scale_test_alpha_for_graph(a, zs_batch) for a in alphas], list(alphas)

    return type(name, (PixelTransform, op_cls), {'__init__': __init__, 'vis_image_batch': vis_image_batch})


SceneGraph = _make_graph('SceneGraph', SceneTransform)
faceGraph = _make_graph('faceGraph', FaceTransform)


def get_transform_graphs(model):
    if model == 'pggan':                                   # BASELINE config 1: z-space walk on the in-repo PGGAN-256 generator
        from . import pggan
        return [pggan.SceneGraph, pggan.faceGraph]
    if model != 'stylegan_v2_real':
        raise ImportError("No module named 'graphs.%s' in this build (stylegan_v2_real and pggan are MI355X-native)" % model)
    return [SceneGraph, faceGraph]


def find_model_using_name(model, transform):
    """graphs/__init__.py:3-22: class whose lower-cased name equals transform.replace('_','') + 'graph'."""
    target = transform.replace('_', '') + 'graph'
    for g in get_transform_graphs(model):
        if g.__name__.lower() == target.lower():
            print('Find NAME: ', target.lower())
            return g
    print("In graphs.transform_graph_scene, there should be a Class with class name that matches %s in lowercase." % target)
    raise SystemExit(0)
