"""Oracle: StyleGAN2 generator / discriminator and their two native ops, plain torch on CPU.

Functional restatement over a reference-layout ``state_dict`` (name -> tensor);
dtype-generic (run it in float64 for a tight ground truth).  TEST
INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# ----------------------------------------------------------------------------------------------
# native op 1: fused_bias_act   (reference graphs/stylegan_v2_real/op/fused_bias_act_kernel.cu:18-49)
# ----------------------------------------------------------------------------------------------
def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    """Semantics table of the kernel's ``switch (act*10+grad)`` (.cu:36-47).

    bias indexes dim 1 (``(xi / step_b) % size_b`` with step_b = prod(dims>=2), .cu:67-71);
    empty ``b`` / ``ref`` mean "not used" (.cu:63-64).
    """
    if b is not None and b.numel():
        x = x + b.reshape([1, -1] + [1] * (x.dim() - 2))
    code = act * 10 + grad
    if code in (10, 11):
        y = x
    elif code in (12, 32):
        y = torch.zeros_like(x)
    elif code == 30:
        y = torch.where(x > 0, x, x * alpha)
    elif code == 31:
        y = torch.where(ref > 0, x, x * alpha)
    else:                                    # kernel's `default:` falls into case 10
        y = x
    return y * scale


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """FusedLeakyReLUFunction.forward (op/fused_act.py:51-60): act=3, grad=0."""
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


def fused_leaky_relu_backward(grad_out, out, negative_slope=0.2, scale=SQRT2):
    """FusedLeakyReLUFunctionBackward.forward (op/fused_act.py:19-37): act=3, grad=1, ref=out; grad_bias = sum."""
    gi = fused_bias_act(grad_out, None, out, 3, 1, negative_slope, scale)
    dims = [0] + list(range(2, gi.dim()))
    return gi, gi.sum(dims)


# ----------------------------------------------------------------------------------------------
# native op 2: upfirdn2d   (reference op/upfirdn2d.py:87-149, op/upfirdn2d_kernel.cu:52-137)
# ----------------------------------------------------------------------------------------------
def upfirdn2d_out_size(n, up, down, pad0, pad1, k):
    """op/upfirdn2d.py:102-103."""
    return (n * up + pad0 + pad1 - k) // down + 1


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """[N,C,H,W] -> zero-insert x``up`` -> pad/crop (pad0 before, pad1 after, both axes) -> correlate with the
    FLIPPED FIR (i.e. true convolution, .cu:79) -> keep every ``down``-th sample."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    pad0, pad1 = pad
    z = x.new_zeros(n, c, h * up, w * up)
    z[:, :, ::up, ::up] = x                               # sample at (i*up), zeros after it
    z = F.pad(z, [max(pad0, 0), max(pad1, 0), max(pad0, 0), max(pad1, 0)])
    hh, ww = z.shape[2], z.shape[3]
    z = z[:, :, max(-pad0, 0):hh - max(-pad1, 0), max(-pad0, 0):ww - max(-pad1, 0)]
    wk = torch.flip(kernel.to(x.dtype), [0, 1]).reshape(1, 1, kh, kw)
    y = F.conv2d(z.reshape(n * c, 1, z.shape[2], z.shape[3]), wk)
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def upfirdn2d_backward(grad_out, kernel, up, down, pad, in_hw):
    """UpFirDn2d.backward (op/upfirdn2d.py:105-115,18-45): same op, flipped kernel, up<->down, g_pad."""
    kh, kw = kernel.shape
    in_h, in_w = in_hw
    out_h, out_w = grad_out.shape[2], grad_out.shape[3]
    pad0, _ = pad
    g0 = kw - pad0 - 1
    g1x = in_w * up - out_w * down + pad0 - up + 1
    g1y = in_h * up - out_h * down + pad0 - up + 1
    assert g1x == g1y, 'square maps only on this path'
    return upfirdn2d(grad_out, torch.flip(kernel, [0, 1]), up=down, down=up, pad=(g0, g1x))


# ----------------------------------------------------------------------------------------------
# generator   (reference networks.py)
# ----------------------------------------------------------------------------------------------
def pixel_norm(x):
    """PixelNorm (networks.py:11-16)."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward (networks.py:148-156)."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    out = F.linear(x, weight * scale)
    if activation:
        return fused_leaky_relu(out, bias * lr_mul)
    return out + bias * lr_mul


def style_mlp(P, z, n_mlp=8, lr_mlp=0.01):
    """Generator.style (networks.py:374-382): PixelNorm + n_mlp x EqualLinear(lr_mul=0.01, fused_lrelu)."""
    x = pixel_norm(z)
    for i in range(1, n_mlp + 1):
        x = equal_linear(x, P['style.%d.weight' % i], P['style.%d.bias' % i], lr_mul=lr_mlp, activation=True)
    return x


def modulated_conv2d(P, prefix, x, w, demodulate=True, upsample=False):
    """ModulatedConv2d.forward (networks.py:231-272), written per sample instead of groups=batch."""
    weight = P[prefix + '.weight'][0]                       # [Cout, Cin, k, k]
    cout, cin, k, _ = weight.shape
    s = equal_linear(w, P[prefix + '.modulation.weight'], P[prefix + '.modulation.bias'])   # [B, Cin]
    scale = 1.0 / math.sqrt(cin * k * k)
    outs = []
    for b in range(x.shape[0]):
        wb = scale * weight * s[b].reshape(1, cin, 1, 1)
        if demodulate:
            d = torch.rsqrt((wb * wb).sum([1, 2, 3]) + 1e-8)
            wb = wb * d.reshape(cout, 1, 1, 1)
        if upsample:
            # conv_transpose2d(stride 2, pad 0) with weight [Cin, Cout, k, k] (networks.py:246-255)
            o = F.conv_transpose2d(x[b:b + 1], wb.transpose(0, 1), stride=2, padding=0)
        else:
            o = F.conv2d(x[b:b + 1], wb, padding=k // 2)
        outs.append(o)
    out = torch.cat(outs, 0)
    if upsample:
        # Blur(pad=(1,1), kernel*4) (networks.py:197-203,256)
        out = upfirdn2d(out, P[prefix + '.blur.kernel'], pad=(1, 1))
    return out


def styled_conv(P, prefix, x, w, noise=None, upsample=False):
    """StyledConv.forward (networks.py:330-336): modconv -> NoiseInjection (:281-286) -> FusedLeakyReLU."""
    out = modulated_conv2d(P, prefix + '.conv', x, w, True, upsample)
    if noise is not None:
        out = out + P[prefix + '.noise.weight'] * noise
    return fused_leaky_relu(out, P[prefix + '.activate.bias'])


def to_rgb(P, prefix, x, w, skip=None):
    """ToRGB.forward (networks.py:349-358); Upsample pads (2,1) (networks.py:38-43)."""
    out = modulated_conv2d(P, prefix + '.conv', x, w, demodulate=False)
    out = out + P[prefix + '.bias']
    if skip is not None:
        out = out + upfirdn2d(skip, P[prefix + '.upsample.kernel'], up=2, pad=(2, 1))
    return out


def generator_synthesis(P, latent, noise=None):
    """Generator.forward with input_is_latent=True (networks.py:494-514).

    latent [B, n_latent, 512]; ``noise``: list of explicit [B,1,r,r] maps (one per styled conv) or None
    (= the noise term is absent, equal to the reference when every ``noise.weight`` is 0)."""
    b = latent.shape[0]
    n_latent = latent.shape[1]
    log_size = (n_latent + 2) // 2
    nz = (lambda i: None) if noise is None else (lambda i: noise[i])
    out = P['input.input'].repeat(b, 1, 1, 1)
    out = styled_conv(P, 'conv1', out, latent[:, 0], nz(0))
    skip = to_rgb(P, 'to_rgb1', out, latent[:, 1])
    i = 1
    for j in range(log_size - 2):
        out = styled_conv(P, 'convs.%d' % (2 * j), out, latent[:, i], nz(2 * j + 1), upsample=True)
        out = styled_conv(P, 'convs.%d' % (2 * j + 1), out, latent[:, i + 1], nz(2 * j + 2))
        skip = to_rgb(P, 'to_rgbs.%d' % j, out, latent[:, i + 2], skip)
        i += 2
    return skip


# ----------------------------------------------------------------------------------------------
# discriminator   (reference networks.py:517-645)
# ----------------------------------------------------------------------------------------------
def _equal_conv(x, weight, bias=None, stride=1, padding=0):
    """EqualConv2d.forward (networks.py:111-120)."""
    scale = 1.0 / math.sqrt(weight.shape[1] * weight.shape[2] * weight.shape[3])
    return F.conv2d(x, weight * scale, bias=bias, stride=stride, padding=padding)


def _conv_layer(P, prefix, x, k, downsample=False, bias=True, activate=True):
    """ConvLayer (networks.py:517-563): [Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU]."""
    idx = 0
    if downsample:
        p = (4 - 2) + (k - 1)
        x = upfirdn2d(x, P['%s.%d.kernel' % (prefix, idx)], pad=((p + 1) // 2, p // 2))
        idx += 1
        stride, padding = 2, 0
    else:
        stride, padding = 1, k // 2
    cb = P.get('%s.%d.bias' % (prefix, idx)) if (bias and not activate) else None
    x = _equal_conv(x, P['%s.%d.weight' % (prefix, idx)], cb, stride, padding)
    idx += 1
    if activate:
        if bias:
            x = fused_leaky_relu(x, P['%s.%d.bias' % (prefix, idx)])
        else:
            x = F.leaky_relu(x, 0.2) * SQRT2
    return x


def discriminator_forward(P, img):
    """Discriminator.forward (networks.py:627-645) incl. minibatch stddev (group 4)."""
    n_res = len([k for k in P if k.endswith('.skip.1.weight')])
    out = _conv_layer(P, 'convs.0', img, 1)
    for n in range(1, n_res + 1):
        p = 'convs.%d' % n
        y = _conv_layer(P, p + '.conv1', out, 3)
        y = _conv_layer(P, p + '.conv2', y, 3, downsample=True)
        s = _conv_layer(P, p + '.skip', out, 1, downsample=True, bias=False, activate=False)
        out = (y + s) / SQRT2
    b, c, h, w = out.shape
    group = min(b, 4)
    sd = out.reshape(group, -1, 1, c, h, w)
    sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8)
    sd = sd.mean([2, 3, 4], keepdim=True).squeeze(2)
    sd = sd.repeat(group, 1, h, w)
    out = torch.cat([out, sd], 1)
    out = _conv_layer(P, 'final_conv', out, 3)
    out = out.reshape(b, -1)
    out = equal_linear(out, P['final_linear.0.weight'], P['final_linear.0.bias'], activation=True)
    out = equal_linear(out, P['final_linear.1.weight'], P['final_linear.1.bias'])
    return out
