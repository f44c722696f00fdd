"""Deterministic synthetic weights / inputs for the walk-training path.

There is no network access for checkpoints, and the reference publishes none
(README.md:17-19 are Google-Drive links), so every benchmark and parity run
uses weights regenerated from numpy ``RandomState`` seeds *keyed by parameter
name* (order independent): the build container, the oracle and the GPU box all
produce bit-identical tensors from ``(name, shape, seed)`` alone.

Initial distributions follow the reference's own constructors
(``networks.py:97-99,135-138,217-221,279,293``) so the synthetic network has
the same signal statistics as a freshly constructed reference model; the
frozen ResNet-50 / VGG-19 use He-normal convs (torchvision's default) so that
activations stay O(1) through 50 layers.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np

from . import specs


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fir_kernel(k=(1, 3, 3, 1), gain=1.0):
    """make_kernel (networks.py:19-27): outer product, normalised, optional up^2 gain."""
    k = np.asarray(k, dtype=np.float32)
    k2 = k[None, :] * k[:, None]
    k2 = k2 / k2.sum()
    return (k2 * np.float32(gain)).astype(np.float32)


def generator_state(size, seed=100, style_dim=512, n_mlp=8, noise_strength=0.0, lr_mlp=0.01):
    """name -> float32 ndarray for reference ``Generator(size, 512, 8)``."""
    out = OrderedDict()
    for name, shape in specs.generator_spec(size, style_dim, n_mlp).items():
        r = _rs('G.' + name, seed)
        if name.startswith('style.') and name.endswith('.weight'):
            v = r.randn(*shape) / lr_mlp                      # EqualLinear: randn.div_(lr_mul)
        elif name.startswith('style.') and name.endswith('.bias'):
            v = r.randn(*shape) * 10.0                         # effective bias = bias*lr_mul ~ 0.1
        elif name.endswith('modulation.bias'):
            v = 1.0 + 0.1 * r.randn(*shape)                    # bias_init=1
        elif name.endswith('blur.kernel') or name.endswith('upsample.kernel'):
            v = fir_kernel(gain=4.0)                           # Blur(upsample_factor=2) / Upsample(factor=2)
        elif name.endswith('noise.weight'):
            v = noise_strength * r.randn(*shape) if noise_strength else np.zeros(shape)
        elif name.endswith('activate.bias') or name.endswith('.bias'):
            v = 0.1 * r.randn(*shape)
        else:                                                  # conv / modulation weights, const input, noises
            v = r.randn(*shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32).reshape(shape)
    return out


def discriminator_state(size, seed=200):
    """name -> float32 ndarray for reference ``Discriminator(size)`` (random-init, as in the reference:
    netD is never loaded, transform_base.py:540-548)."""
    out = OrderedDict()
    for name, shape in specs.discriminator_spec(size).items():
        r = _rs('D.' + name, seed)
        if name.endswith('.kernel'):
            v = fir_kernel(gain=1.0)
        elif name.endswith('.bias'):
            v = 0.1 * r.randn(*shape)
        else:
            v = r.randn(*shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32).reshape(shape)
    return out


def resnet50_state(num_classes=40, seed=300):
    """name -> ndarray for the attribute regressor; BN statistics are perturbed so that folding is exercised;
    fc.bias = 0.5 puts raw predictions inside (0,1) where the BCE of transform_base.py:412-414 has gradient."""
    out = OrderedDict()
    for name, shape in specs.resnet50_spec(num_classes).items():
        r = _rs('R.' + name, seed)
        if name.endswith('num_batches_tracked'):
            out[name] = np.zeros((), dtype=np.int64)
            continue
        if name.endswith('running_var'):
            v = r.uniform(0.5, 1.5, shape)
        elif name.endswith('running_mean'):
            v = 0.1 * r.randn(*shape)
        elif '.bn' in name or name.startswith('bn1') or 'downsample.1' in name:
            if name.endswith('weight'):
                # last BN of each bottleneck kept small so that residual sums stay O(1)
                v = r.uniform(0.2, 0.4, shape) if name.endswith('bn3.weight') else r.uniform(0.8, 1.2, shape)
            else:
                v = 0.05 * r.randn(*shape)
        elif name == 'fc.weight':
            v = r.randn(*shape) * (0.1 / math.sqrt(shape[1]))
        elif name == 'fc.bias':
            v = 0.5 + 0.05 * r.randn(*shape)
        else:                                                  # conv weights: He normal, fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            v = r.randn(*shape) * math.sqrt(2.0 / fan_out)
        out[name] = np.ascontiguousarray(v, dtype=np.float32).reshape(shape)
    return out


def vgg19_prefix_state(seed=400):
    out = OrderedDict()
    for name, shape in specs.vgg19_prefix_spec().items():
        r = _rs('V.' + name, seed)
        if name.endswith('bias'):
            v = 0.05 * r.randn(*shape)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            v = r.randn(*shape) * math.sqrt(2.0 / fan_in)
        out[name] = np.ascontiguousarray(v, dtype=np.float32).reshape(shape)
    return out


def walk_init(n_attr, n_latent, dim=512, seed=7):
    """WalkLinearMultiW init N(0, 0.02) (transform_base.py:147) from a *seeded* generator
    (the reference uses the unseeded global np.random)."""
    return np.random.RandomState(seed).normal(0.0, 0.02, [n_attr, n_latent, dim]).astype(np.float32)


def z_sample(n, seed=0, dim_z=512):
    """graph_util.z_sample (graph_util.py:5-8): RandomState(seed).randn(n, dim_z), float64."""
    return np.random.RandomState(seed).randn(n, dim_z)


def noise_maps(size, batch, seed=500):
    """Explicit per-layer noise [B,1,r,r] for parity runs (Generator.forward noise=[...], networks.py:468,476-483)."""
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    maps = []
    for l in range(num_layers):
        res = 2 ** ((l + 5) // 2)
        maps.append(_rs('noise.%d' % l, seed).randn(batch, 1, res, res).astype(np.float32))
    return maps


def pggan_generator_state(seed=100):
    """name -> float32 ndarray for ``model_256.Generator(511, 1)``: N(0,1) ``weight_orig`` like EqualConv2d's constructor
    (model_256.py:99-100; the sqrt(2/fan_in) scale is applied at run time), small biases, N(0,1) label embedding."""
    out = OrderedDict()
    for name, shape in specs.pggan_generator_spec().items():
        r = _rs('PG.' + name, seed)
        if name.endswith('weight_orig') or name == 'label_embed.weight':
            v = r.randn(*shape)
        elif name.endswith('.weight'):
            v = r.randn(*shape) / math.sqrt(shape[1])               # to_rgb: plain nn.Conv2d
        else:
            v = 0.1 * r.randn(*shape)
        out[name] = v.astype(np.float32)
    return out
