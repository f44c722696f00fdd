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


class _ModPlan:
    """Segment tables of l2i_segmented_matvec_f32 (include/l2i.h) for one generator: all style-dependent vectors of a pass in four
    launches instead of ~10 tiny rocBLAS / elementwise launches per layer.

    forward   (2 launches): s = w_l A^T + bias for every styled conv and ToRGB (+ the ToRGB weights wmod = W * s), then
                            demod = rsqrt(s^2 T^T + 1e-8) for every styled conv                         (networks.py:148-156, 231-239, 346-351)
    backward  (2 launches): d s = q - s * ((red / demod * demod^3) T) per conv, then the whole latent gradient
                            g_lat[:, l] = sum over the layers fed by latent l of d s A (ToRGB: d s_rgb = sum_o red_rgb[..., o] W[o])
    Buffers are per-layer contiguous ([layer][B][C]) so that every layer's vector is a contiguous [B, C] view (the conv kernels take it by
    pointer); offsets that scale with the batch are stored per sample, the tables do not depend on B."""

    def __init__(self, gen, device):
        from . import _lib
        import ctypes
        layers, rgbs = gen.layers, gen.rgbs
        conv_idx, rgb_idx = gen.latent_index()
        sd, nl = gen.style_dim, gen.n_latent
        self.device, self.nl, self.sd = device, nl, sd
        self.cin = [L.cin for L in layers]
        self.cout = [L.cout for L in layers]
        self.crgb = [R.W.shape[1] for R in rgbs]
        for k in self.cin + self.cout + self.crgb + [sd]:
            assert k % 4 == 0 and k <= 512, 'l2i_segmented_matvec_f32: contraction lengths must be multiples of 4 up to 512 (got %d)' % k
        pre = lambda xs: [int(v) for v in np.concatenate([[0], np.cumsum(xs)[:-1]])] if xs else []
        self.s_off = pre(self.cin) + [sum(self.cin) + o for o in pre(self.crgb)]          # rows before each segment in s_all (convs, then ToRGBs)
        self.d_off = pre(self.cout)                                                       # ... in demod_all / red_dz_all
        self.r_off = pre(self.crgb)                                                       # channels before each ToRGB (x 3: wmod_all / red_rgb_all)
        self.n_s, self.n_d, self.n_r = sum(self.cin) + sum(self.crgb), sum(self.cout), sum(self.crgb)
        self.n_q = sum(self.cin)
        mods = [L.mod for L in layers] + [R.mod for R in rgbs]
        rows = self.cin + self.crgb
        lat_of = list(conv_idx) + list(rgb_idx)
        # weights, concatenated per use
        self.w_modT = torch.cat([m.A.t().contiguous().reshape(-1) for m in mods]).contiguous()          # [512, rows] per segment
        self.w_A = torch.cat([m.A.contiguous().reshape(-1) for m in mods]).contiguous()                 # [rows, 512] per segment
        self.bias = torch.cat([m.b.reshape(-1) for m in mods]).contiguous()
        self.w_rgb = torch.cat([R.W.contiguous().reshape(-1) for R in rgbs]).contiguous()               # [3, C] per ToRGB
        self.w_Tt = torch.cat([L.T.t().contiguous().reshape(-1) for L in layers]).contiguous()          # [cin, cout] per conv
        self.w_T = torch.cat([L.T.contiguous().reshape(-1) for L in layers]).contiguous()               # [cout, cin] per conv
        a_off = pre([r * sd for r in rows])
        t_off = pre([ci * co for ci, co in zip(self.cin, self.cout)])
        nconv = len(layers)

        def part(K, in_off_c=0, in_off_b=0, in_bstride=0, w_pitch=0, pre_=0, aux_off=0, w_off=0):
            return _lib.SegmvPart(K, in_off_c, in_off_b, in_bstride, w_pitch, pre_, aux_off, 0, w_off)

        def seg(rows_, parts, out_off_c=0, out_off_b=0, out_bstride=0, epi=0, bias_off=0, e_off_b=0, e_bstride=0, rgb_off_b=-1, rgb_w_off=0):
            sg = _lib.SegmvSeg()
            sg.rows, sg.nparts, sg.out_off_c, sg.out_off_b, sg.out_bstride, sg.epi, sg.bias_off = rows_, len(parts), out_off_c, out_off_b, out_bstride, epi, bias_off
            sg.e_off_c, sg.e_off_b, sg.e_bstride, sg.rgb_off_b, sg.rgb_w_off = 0, e_off_b, e_bstride, rgb_off_b, rgb_w_off
            for i, pt in enumerate(parts):
                sg.part[i] = pt
            return sg

        mod, dem, ds = [], [], []
        for i, r in enumerate(rows):
            j = i - nconv
            mod.append(seg(r, [part(sd, in_off_c=lat_of[i] * sd, in_bstride=nl * sd, w_pitch=r, w_off=a_off[i])], out_off_b=self.s_off[i], out_bstride=r,
                           epi=0, bias_off=self.s_off[i], rgb_off_b=(3 * self.r_off[j] if j >= 0 else -1), rgb_w_off=(3 * self.r_off[j] if j >= 0 else 0)))
        for i in range(nconv):
            ci, co = self.cin[i], self.cout[i]
            dem.append(seg(co, [part(ci, in_off_b=self.s_off[i], in_bstride=ci, w_pitch=co, pre_=1, w_off=t_off[i])], out_off_b=self.d_off[i], out_bstride=co, epi=1))
            ds.append(seg(ci, [part(co, in_off_b=self.d_off[i], in_bstride=co, w_pitch=ci, pre_=2, w_off=t_off[i])], out_off_b=self.s_off[i], out_bstride=ci,
                          epi=2, e_off_b=self.s_off[i], e_bstride=ci))
        glat = []
        for l in range(nl):
            parts = []
            for i in range(nconv):
                if conv_idx[i] == l:
                    parts.append(part(self.cin[i], in_off_b=self.s_off[i], in_bstride=self.cin[i], w_pitch=sd, pre_=0, w_off=a_off[i]))
            for j in range(len(rgbs)):
                if rgb_idx[j] == l:       # pre 3: the [B, C, 3] reduction of the ToRGB, read from the in2 slot
                    parts.append(part(self.crgb[j], in_off_b=3 * self.r_off[j], in_bstride=0, w_pitch=sd, pre_=3, aux_off=3 * self.r_off[j], w_off=a_off[nconv + j]))
            assert 1 <= len(parts) <= 2
            glat.append(seg(sd, parts, out_off_c=l * sd, out_bstride=nl * sd, epi=3))

        def table(segs):
            arr = (_lib.SegmvSeg * len(segs))(*segs)
            raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8).clone()
            blocks = [(i, rb) for i, sg in enumerate(segs) for rb in range((sg.rows + 63) // 64)]
            return raw.to(device), torch.tensor(blocks, dtype=torch.int32).reshape(-1).to(device), len(blocks)
        self.t_mod, self.t_dem, self.t_ds, self.t_glat = table(mod), table(dem), table(ds), table(glat)

    # -- views -------------------------------------------------------------------------------------------------------------------------
    def s(self, s_all, B, li):
        return s_all[B * self.s_off[li]:B * (self.s_off[li] + self.cin[li])].view(B, self.cin[li])

    def demod(self, d_all, B, li):
        return d_all[B * self.d_off[li]:B * (self.d_off[li] + self.cout[li])].view(B, self.cout[li])

    def wmod(self, w_all, B, j):
        c = self.crgb[j]
        return w_all[3 * B * self.r_off[j]:3 * B * (self.r_off[j] + c)].view(B, 3, c)

    def red_rgb(self, r_all, B, j):
        c = self.crgb[j]
        return r_all[3 * B * self.r_off[j]:3 * B * (self.r_off[j] + c)].view(B, c, 3)

    # -- launches ----------------------------------------------------------------------------------------------------------------------
    def forward(self, lat):
        """lat [B, n_latent, 512] contiguous -> (s_all, demod_all, wmod_all), flat, per-layer contiguous."""
        B, dev = lat.shape[0], lat.device
        s_all = torch.empty(B * self.n_s, device=dev, dtype=torch.float32)
        d_all = torch.empty(B * self.n_d, device=dev, dtype=torch.float32)
        w_all = torch.empty(3 * B * self.n_r, device=dev, dtype=torch.float32)
        K.segmented_matvec(s_all, lat, self.w_modT, self.t_mod[0], self.t_mod[1], self.t_mod[2], B, bias=self.bias, wmod=w_all, wrgb=self.w_rgb)
        K.segmented_matvec(d_all, s_all, self.w_Tt, self.t_dem[0], self.t_dem[1], self.t_dem[2], B)
        return s_all, d_all, w_all

    def reductions(self, B, dev):
        """One zeroed buffer for every per-layer reduction of a backward pass: (red_dz_z like demod_all, q like the conv part of s_all,
        red_x_grgb [layer][B][C][3])."""
        buf = torch.zeros(B * (self.n_d + self.n_q + 3 * self.n_r), device=dev, dtype=torch.float32)
        return buf[:B * self.n_d], buf[B * self.n_d:B * (self.n_d + self.n_q)], buf[B * (self.n_d + self.n_q):]

    def backward(self, B, s_all, d_all, red_dz, q, red_rgb):
        """-> g_lat [B, n_latent, 512]."""
        dev = s_all.device
        ds_all = torch.empty(B * self.n_q, device=dev, dtype=torch.float32)
        K.segmented_matvec(ds_all, red_dz, self.w_T, self.t_ds[0], self.t_ds[1], self.t_ds[2], B, in2=d_all, e1=q, e2=s_all)
        g_lat = torch.empty(B, self.nl, self.sd, device=dev, dtype=torch.float32)
        # parts of kind 0 read d s (`in`), parts of kind 3 the ToRGB reductions (`in2`)
        K.segmented_matvec(g_lat, ds_all, self.w_A, self.t_glat[0], self.t_glat[1], self.t_glat[2], B, in2=red_rgb, wrgb=self.w_rgb)
        return g_lat


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
        self.modplan = _ModPlan(self, device)

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


def _draw_noise(gen, noise, b, device):
    """Per-layer noise maps of one pass (networks.py:281-286: a fresh N(0,1) map per NoiseInjection call): the caller's explicit list, or — with
    ``randomize_noise`` — ONE normal draw for the whole pass cut into the per-layer maps (17 launches less per pass than a draw per layer; every
    element is still an independent N(0,1) sample), or nothing."""
    if noise is not None or not gen.randomize_noise:
        return None
    sizes = []
    for li, L in enumerate(gen.layers):
        res = 4 << ((li + 1) // 2)                          # conv1 at 4, then (up, conv) pairs at 8, 16, ...
        sizes.append(b * res * res if L.noise_w != 0.0 else 0)
    if sum(sizes) == 0:
        return None
    flat = torch.randn(sum(sizes), device=device)
    out, off = [], 0
    for li, n in enumerate(sizes):
        res = 4 << ((li + 1) // 2)
        out.append(flat[off:off + n].view(b, 1, res, res) if n else None)
        off += n
    return out


def _noise_for(gen, noise, li, b, res, device, drawn=None):
    L = gen.layers[li]
    if noise is not None:
        return noise[li].contiguous() if L.noise_w != 0.0 else None
    if drawn is not None:
        assert drawn[li] is None or drawn[li].shape[2] == res
        return drawn[li]
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
        plan = gen.modplan
        s_all, d_all, w_all = plan.forward(lat.contiguous())       # every modulation / demodulation / ToRGB weight of the pass: two launches
        drawn = _draw_noise(gen, noise, B, dev)
        for li, L in enumerate(gen.layers):
            s, demod = plan.s(s_all, B, li), plan.demod(d_all, B, li)
            h = x.shape[2]
            res = h * 2 if L.up else h
            nz = _noise_for(gen, noise, li, B, res, dev, drawn)
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
                wmod = plan.wmod(w_all, B, li // 2)                                  # [B,3,C] = scale * W * s_rgb
                rgb = K.torgb_fwd(y, wmod, R.bias)
                if R.up:
                    skip = K.upfirdn2d(skip, R.up_k, up=(2, 2), pad=(2, 1, 2, 1), addend=rgb)
                else:
                    skip = rgb
                rec['wmod'] = wmod
            saved.append(rec)
            x = y
        ctx.gen, ctx.saved, ctx.B = gen, saved if keep else None, B
        ctx.mod = (s_all, d_all) if keep else None
        return skip

    @staticmethod
    def backward(ctx, g_img):
        gen, saved, B = ctx.gen, ctx.saved, ctx.B
        if saved is None:
            raise RuntimeError('synthesis was run without a differentiable latent')
        dev = g_img.device
        conv_idx, rgb_idx = gen.latent_index()
        plan = gen.modplan
        s_all, d_all = ctx.mod
        red_dz, q_all, red_rgb = plan.reductions(B, dev)               # every per-layer reduction of this pass lands in one zeroed buffer
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
            # [r5] gin (the gradient w.r.t. layer li + 1's modulated input) and y (that layer's input) both pass through this kernel: layer li + 1's
            # style gradient sum_p gin * y is formed here instead of by a dot_reduce pass that reads both maps again
            dz, _, _ = K.sg2_act_bwd(rec['y'], gin, gin_scale, grgb, rec.get('wmod'), L.bias, rec['nz'], L.noise_w, 0.2, SQRT2,
                                     red=plan.demod(red_dz, B, li), red_rgb=plan.red_rgb(red_rgb, B, li // 2) if has_rgb else None,
                                     red_q=plan.s(q_all, B, li + 1).view(-1) if gin is not None else None)
            demod, s = rec['demod'], rec['s']
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
            if li == 0:                                                               # (the constant input has no producing layer)
                K.dot_reduce(dxmod, x, out=plan.s(q_all, B, li).view(-1))            # [B,Cin] = d s via x*s
            gin, gin_scale = dxmod, s
            rec['y'] = rec['x'] = None
        # d s = q - s * ((d demod * demod^3) T) for every conv, then the whole latent gradient (d s A, and d s_rgb A of the ToRGBs): two launches
        g_lat = plan.backward(B, s_all, d_all, red_dz, q_all, red_rgb)
        return g_lat, None, None
