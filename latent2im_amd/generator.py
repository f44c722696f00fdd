"""StyleGAN2 generator (frozen weights) on the l2i HIP kernels: mapping network + synthesis stack with a
hand-scheduled backward that produces only d(image)/d(W+ latent) — the single gradient the walk needs.

Reference: graphs/stylegan_v2_real/networks.py:360-514 (Generator), :176-272 (ModulatedConv2d), :302-358
(StyledConv, ToRGB).  Arithmetic is restructured for the hardware, results are the same function:

* activation-modulated convolution: y = demod[b,o] * conv(x * s[b,i], W*scale) instead of materialising a per-sample
  weight and running a grouped conv (networks.py:235-270).  demod[b,o] = rsqrt(sum_i s[b,i]^2 * T[o,i] + 1e-8) with
  T[o,i] = sum_k (scale*W[o,i,k])^2 precomputed once.
* the stride-2 transposed conv of the up layers is issued as its four output phases (no zero-insertion MACs), then the
  4x4 blur with noise + bias + leaky-ReLU fused into the FIR kernel's epilogue.
* backward: one fused elementwise pass per layer (leaky-ReLU' + ToRGB branch + the two reductions that give d demod and
  d s_rgb), one input-gradient conv, one dot-reduction for d s.  No weight gradients (all weights are frozen,
  transform_base.py:329-331 optimises the walk only).
"""
import math

import numpy as np
import torch

from . import conv as C
from . import kernels as K
from . import specs

SQRT2 = math.sqrt(2.0)


def _t(a, device):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous().to(device)


class _Mod:
    """EqualLinear(style_dim, cin, bias_init=1) of a ModulatedConv2d (networks.py:221,148-156)."""

    def __init__(self, P, prefix, device):
        w = _t(P[prefix + '.modulation.weight'], device)
        self.A = (w * (1.0 / math.sqrt(w.shape[1]))).contiguous()          # [Cin, 512]
        self.b = _t(P[prefix + '.modulation.bias'], device)

    def __call__(self, wl):                                                 # [B,512] -> [B,Cin]
        return torch.addmm(self.b, wl, self.A.t())


class _StyledLayer:
    def __init__(self, P, prefix, cin, cout, upsample, device):
        w = torch.as_tensor(np.asarray(P[prefix + '.conv.weight']), dtype=torch.float32)[0]       # [Cout,Cin,3,3]
        scale = 1.0 / math.sqrt(cin * 9)
        ws = w * scale
        self.cin, self.cout, self.up = cin, cout, upsample
        self.conv = C.FrozenConv2d(ws, stride=2 if upsample else 1, padding=0 if upsample else 1,
                                   transposed=upsample, device=device)
        self.T = (ws * ws).sum((2, 3)).contiguous().to(device)              # [Cout, Cin]
        self.mod = _Mod(P, prefix + '.conv', device)
        self.noise_w = float(np.asarray(P[prefix + '.noise.weight']).reshape(-1)[0])
        self.bias = _t(P[prefix + '.activate.bias'], device)
        if upsample:
            self.blur_k = _t(P[prefix + '.conv.blur.kernel'], device)
            self.blur_k_flip = torch.flip(self.blur_k, [0, 1]).contiguous()


class _ToRGB:
    def __init__(self, P, prefix, cin, upsample, device):
        w = torch.as_tensor(np.asarray(P[prefix + '.conv.weight']), dtype=torch.float32)[0, :, :, 0, 0]   # [3,Cin]
        self.W = (w * (1.0 / math.sqrt(cin))).contiguous().to(device)
        self.mod = _Mod(P, prefix + '.conv', device)
        self.bias = _t(np.asarray(P[prefix + '.bias']).reshape(3), device)
        self.up = upsample
        if upsample:
            self.up_k = _t(P[prefix + '.upsample.kernel'], device)
            self.up_k_flip = torch.flip(self.up_k, [0, 1]).contiguous()


class Generator:
    """Frozen ``Generator(size, 512, 8)``.  ``style(z)`` = mapping network; ``synthesis(latent, noise)`` = forward
    with ``input_is_latent=True``.  Both are differentiable w.r.t. their first argument only."""

    def __init__(self, state, size, device='cuda', style_dim=512, n_mlp=8, lr_mlp=0.01):
        self.size, self.device, self.style_dim = size, device, style_dim
        self.log_size = int(math.log2(size))
        self.n_latent = self.log_size * 2 - 2
        self.num_layers = (self.log_size - 2) * 2 + 1
        P = state
        # mapping network: PixelNorm + n_mlp x EqualLinear(lr_mul, fused_lrelu)  (networks.py:374-382,148-151)
        self.mlp = []
        for i in range(1, n_mlp + 1):
            w = _t(P['style.%d.weight' % i], device)
            self.mlp.append(((w * ((1.0 / math.sqrt(w.shape[1])) * lr_mlp)).t().contiguous(),
                             _t(P['style.%d.bias' % i], device) * lr_mlp))
        self.const = _t(P['input.input'], device)                            # [1,512,4,4]
        geo, _ = specs.generator_geometry(size)
        self.layers = [_StyledLayer(P, name, cin, cout, up, device) for name, cin, cout, res, up in geo]
        self.rgbs = [_ToRGB(P, 'to_rgb1', geo[0][2], False, device)]
        for j in range(self.log_size - 2):
            self.rgbs.append(_ToRGB(P, 'to_rgbs.%d' % j, geo[2 + 2 * j][2], True, device))
        self.randomize_noise = True

    # -- mapping network -----------------------------------------------------------------------------------------
    def style(self, z):
        from .op import fused_leaky_relu
        x = z * torch.rsqrt(torch.mean(z * z, dim=1, keepdim=True) + 1e-8)
        for wt, b in self.mlp:
            x = fused_leaky_relu(torch.mm(x, wt), b)
        return x

    # -- synthesis -----------------------------------------------------------------------------------------------
    def synthesis(self, latent, noise=None):
        """latent [B, n_latent, 512] -> image [B,3,size,size].  ``noise``: list of [B,1,r,r] maps, or None: fresh
        N(0,1) maps when ``randomize_noise`` (only drawn for layers whose noise weight is non-zero)."""
        return _SynthesisFn.apply(latent, self, noise)

    def __call__(self, styles, input_is_latent=True, noise=None, randomize_noise=True, **_):
        """Reference call shape netG(w, input_is_latent=True) -> (image, None)  (networks.py:460-514)."""
        if not input_is_latent:
            raise NotImplementedError('input_is_latent=False is dead in the reference too (networks.py:471-474)')
        return self.synthesis(styles, noise), None

    def latent_index(self):
        """(style index per styled conv, style index per ToRGB)  (networks.py:495-506)."""
        conv_idx = [0] + [i for j in range(self.log_size - 2) for i in (1 + 2 * j, 2 + 2 * j)]
        rgb_idx = [1] + [3 + 2 * j for j in range(self.log_size - 2)]
        return conv_idx, rgb_idx


def _noise_for(gen, noise, li, b, res, device):
    L = gen.layers[li]
    if noise is not None:
        return noise[li].contiguous() if L.noise_w != 0.0 else None
    if gen.randomize_noise and L.noise_w != 0.0:
        return torch.randn(b, 1, res, res, device=device)
    return None


class _SynthesisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latent, gen, noise):
        B = latent.shape[0]
        dev = latent.device
        lat = latent.detach()
        conv_idx, rgb_idx = gen.latent_index()
        keep = latent.requires_grad
        saved = []
        x = gen.const.expand(B, -1, -1, -1).contiguous()
        skip = None
        for li, L in enumerate(gen.layers):
            s = L.mod(lat[:, conv_idx[li]].contiguous())
            demod = torch.rsqrt(torch.mm(s * s, L.T.t()) + 1e-8)
            h = x.shape[2]
            res = h * 2 if L.up else h
            nz = _noise_for(gen, noise, li, B, res, dev)
            if L.up:
                # the (2H+1)^2 map of the transposed conv is produced (2H+4)^2 (three zero rows / columns at the far edge, cropped by the
                # blur's negative far pad): whole 16-byte rows for the conv's stores and the FIR's row vectors
                if h >= 32 and L.conv.fwd_fused is not None:
                    t = L.conv.forward(x, out=torch.empty(B, L.cout, 2 * h + 4, 2 * h + 4, device=dev, dtype=torch.float32), in_scale=s, out_scale=demod)
                    fpad = (1, -2, 1, -2)
                else:
                    t = L.conv.forward(x, in_scale=s, out_scale=demod)
                    fpad = (1, 1, 1, 1)
                y = K.upfirdn2d(t, L.blur_k, pad=fpad, noise=nz, noise_w=L.noise_w, bias=L.bias,
                                act=K.ACT_LRELU, slope=0.2, gain=SQRT2)
                del t
            else:
                y = L.conv.forward(x, in_scale=s, out_scale=demod, noise=nz, noise_w=L.noise_w, bias=L.bias,
                                   act=C.ACT_LRELU, slope=0.2, gain=SQRT2)
            rec = dict(x=x if keep else None, y=y if keep else None, s=s, demod=demod, nz=nz)
            if li == 0 or (li % 2 == 0):                # conv1 and every second conv of a block feed a ToRGB
                R = gen.rgbs[li // 2]
                srgb = R.mod(lat[:, rgb_idx[li // 2]].contiguous())
                wmod = (R.W.unsqueeze(0) * srgb.unsqueeze(1)).contiguous()          # [B,3,C]
                rgb = K.torgb_fwd(y, wmod, R.bias)
                if R.up:
                    skip = K.upfirdn2d(skip, R.up_k, up=(2, 2), pad=(2, 1, 2, 1), addend=rgb)
                else:
                    skip = rgb
                rec['wmod'] = wmod
                rec['srgb'] = srgb
            saved.append(rec)
            x = y
        ctx.gen, ctx.saved, ctx.B = gen, saved if keep else None, B
        return skip

    @staticmethod
    def backward(ctx, g_img):
        gen, saved, B = ctx.gen, ctx.saved, ctx.B
        if saved is None:
            raise RuntimeError('synthesis was run without a differentiable latent')
        dev = g_img.device
        conv_idx, rgb_idx = gen.latent_index()
        g_lat = torch.zeros(B, gen.n_latent, gen.style_dim, device=dev, dtype=torch.float32)
        # gradient of every ToRGB output: skip_j = up(skip_{j-1}) + rgb_j  (networks.py:353-356)
        n_rgb = len(gen.rgbs)
        g_rgb = [None] * n_rgb
        g = g_img.contiguous()
        for j in range(n_rgb - 1, -1, -1):
            g_rgb[j] = g
            if j > 0:
                g = K.upfirdn2d(g, gen.rgbs[j].up_k_flip, up=(1, 1), down=(2, 2), pad=(1, 1, 1, 1))
        gin, gin_scale = None, None
        for li in range(len(gen.layers) - 1, -1, -1):
            L, rec = gen.layers[li], saved[li]
            has_rgb = 'wmod' in rec
            grgb = g_rgb[li // 2] if has_rgb else None
            dz, red_dz_z, red_x_grgb = K.sg2_act_bwd(rec['y'], gin, gin_scale, grgb, rec.get('wmod'), L.bias, rec['nz'],
                                                     L.noise_w, 0.2, SQRT2)
            if has_rgb:
                R = gen.rgbs[li // 2]
                d_srgb = (red_x_grgb * R.W.t().unsqueeze(0)).sum(2)                 # [B,C]
                g_lat[:, rgb_idx[li // 2]] += torch.mm(d_srgb, R.mod.A)
            demod, s = rec['demod'], rec['s']
            d_demod = red_dz_z / demod
            x = rec['x']
            hw = (x.shape[2], x.shape[3])
            if L.up:
                # gradient of the (2H+1)^2 map under the blur, produced (2H+4)^2: three more rows / columns at the far edge that the
                # stride-2 gradient conv never reads, so that its rows are whole 16-byte vectors (see discriminator.py)
                dt = K.upfirdn2d(dz, L.blur_k_flip, pad=(2, 5, 2, 5))
                del dz
                dxmod = L.conv.dgrad(dt, hw, in_scale=demod)
                del dt
            else:
                dxmod = L.conv.dgrad(dz, hw, in_scale=demod)
                del dz
            q = K.dot_reduce(dxmod, x)                                              # [B,Cin] = d s via x*s
            d_s = q - s * torch.mm(d_demod * demod * demod * demod, L.T)
            g_lat[:, conv_idx[li]] += torch.mm(d_s, L.mod.A)
            gin, gin_scale = dxmod, s
            rec['y'] = rec['x'] = None
        return g_lat, None, None
