"""Parameter layouts (name -> shape) of the frozen networks on the walk-training path.

Names and shapes follow the *reference's* ``state_dict()`` keys so that real
checkpoints (``ckpt['g_ema']`` for the generator, ``ckpt['model']`` for the
regressor, torchvision ``vgg19().features`` for the perceptual prefix) load
unchanged:

* StyleGAN2 generator / discriminator: reference
  ``graphs/stylegan_v2_real/networks.py:360-438`` (Generator.__init__) and
  ``:587-625`` (Discriminator.__init__).
* ResNet-50 (torchvision tag v0.5.0, fc -> 40): reference call site
  ``graphs/stylegan_v2_real/transform_base.py:522-528``; architecture per
  SURVEY.md Appendix C (source not vendored in the reference).
* VGG-19 ``features`` prefix (layers 0..7): reference call site
  ``graphs/stylegan_v2_real/transform_base.py:536-538`` / ``:426-454``.

Pure python (no torch) so that the spec can be unit-tested on any box.
"""
from collections import OrderedDict
import math

# channels per resolution, reference networks.py:384-394 / :591-601
def sg2_channels(channel_multiplier=2):
    return {
        4: 512, 8: 512, 16: 512, 32: 512,
        64: 256 * channel_multiplier,
        128: 128 * channel_multiplier,
        256: 64 * channel_multiplier,
        512: 32 * channel_multiplier,
        1024: 16 * channel_multiplier,
    }


def _styled_conv(spec, prefix, cin, cout, k, style_dim, upsample):
    # ModulatedConv2d: networks.py:217-221 ; Blur buffer :81 ; NoiseInjection :279 ; FusedLeakyReLU bias
    spec[prefix + '.conv.weight'] = (1, cout, cin, k, k)
    if upsample:
        spec[prefix + '.conv.blur.kernel'] = (4, 4)
    spec[prefix + '.conv.modulation.weight'] = (cin, style_dim)
    spec[prefix + '.conv.modulation.bias'] = (cin,)
    spec[prefix + '.noise.weight'] = (1,)
    spec[prefix + '.activate.bias'] = (cout,)


def _to_rgb(spec, prefix, cin, style_dim, upsample):
    # ToRGB: networks.py:339-347 (bias, upsample.kernel buffer, 1x1 modconv without demod)
    spec[prefix + '.bias'] = (1, 3, 1, 1)
    if upsample:
        spec[prefix + '.upsample.kernel'] = (4, 4)
    spec[prefix + '.conv.weight'] = (1, 3, cin, 1, 1)
    spec[prefix + '.conv.modulation.weight'] = (cin, style_dim)
    spec[prefix + '.conv.modulation.bias'] = (cin,)


def generator_spec(size, style_dim=512, n_mlp=8, channel_multiplier=2):
    """state_dict layout of reference ``Generator(size, style_dim, n_mlp)`` in registration order."""
    ch = sg2_channels(channel_multiplier)
    log_size = int(math.log2(size))
    assert 2 ** log_size == size and 3 <= log_size <= 10
    spec = OrderedDict()
    for i in range(n_mlp):                      # style = Sequential(PixelNorm, EqualLinear x n_mlp)
        spec['style.%d.weight' % (i + 1)] = (style_dim, style_dim)
        spec['style.%d.bias' % (i + 1)] = (style_dim,)
    spec['input.input'] = (1, ch[4], 4, 4)
    _styled_conv(spec, 'conv1', ch[4], ch[4], 3, style_dim, False)
    _to_rgb(spec, 'to_rgb1', ch[4], style_dim, False)
    cin = ch[4]
    convs, rgbs = OrderedDict(), OrderedDict()
    for j, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        _styled_conv(convs, 'convs.%d' % (2 * j), cin, cout, 3, style_dim, True)
        _styled_conv(convs, 'convs.%d' % (2 * j + 1), cout, cout, 3, style_dim, False)
        _to_rgb(rgbs, 'to_rgbs.%d' % j, cout, style_dim, True)
        cin = cout
    spec.update(convs)
    spec.update(rgbs)
    num_layers = (log_size - 2) * 2 + 1
    for l in range(num_layers):                 # networks.py:412-415
        res = (l + 5) // 2
        spec['noises.noise_%d' % l] = (1, 1, 2 ** res, 2 ** res)
    return spec


def generator_geometry(size, channel_multiplier=2):
    """[(name, cin, cout, out_res, upsample)] for the synthesis convs, plus n_latent."""
    ch = sg2_channels(channel_multiplier)
    log_size = int(math.log2(size))
    layers = [('conv1', ch[4], ch[4], 4, False)]
    cin = ch[4]
    for j, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        layers.append(('convs.%d' % (2 * j), cin, cout, 2 ** i, True))
        layers.append(('convs.%d' % (2 * j + 1), cout, cout, 2 ** i, False))
        cin = cout
    return layers, log_size * 2 - 2


def _conv_layer(spec, prefix, cin, cout, k, downsample, bias=True, activate=True):
    # ConvLayer(nn.Sequential): networks.py:517-563. index: [Blur] EqualConv2d [FusedLeakyReLU]
    idx = 0
    if downsample:
        spec['%s.%d.kernel' % (prefix, idx)] = (4, 4)
        idx += 1
    spec['%s.%d.weight' % (prefix, idx)] = (cout, cin, k, k)
    if bias and not activate:
        spec['%s.%d.bias' % (prefix, idx)] = (cout,)
    idx += 1
    if activate and bias:
        spec['%s.%d.bias' % (prefix, idx)] = (cout,)


def discriminator_spec(size, channel_multiplier=2):
    """state_dict layout of reference ``Discriminator(size)`` (networks.py:587-625)."""
    ch = sg2_channels(channel_multiplier)
    log_size = int(math.log2(size))
    spec = OrderedDict()
    _conv_layer(spec, 'convs.0', 3, ch[size], 1, False)
    cin = ch[size]
    for n, i in enumerate(range(log_size, 2, -1)):
        cout = ch[2 ** (i - 1)]
        p = 'convs.%d' % (n + 1)
        _conv_layer(spec, p + '.conv1', cin, cin, 3, False)
        _conv_layer(spec, p + '.conv2', cin, cout, 3, True)
        _conv_layer(spec, p + '.skip', cin, cout, 1, True, bias=False, activate=False)
        cin = cout
    _conv_layer(spec, 'final_conv', cin + 1, ch[4], 3, False)
    spec['final_linear.0.weight'] = (ch[4], ch[4] * 16)
    spec['final_linear.0.bias'] = (ch[4],)
    spec['final_linear.1.weight'] = (1, ch[4])
    spec['final_linear.1.bias'] = (1,)
    return spec


RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))   # (planes, blocks, stride)


def _bn(spec, prefix, c):
    spec[prefix + '.weight'] = (c,)
    spec[prefix + '.bias'] = (c,)
    spec[prefix + '.running_mean'] = (c,)
    spec[prefix + '.running_var'] = (c,)
    spec[prefix + '.num_batches_tracked'] = ()


def resnet50_spec(num_classes=40):
    """torchvision v0.5.0 ``resnet50`` with ``fc = Linear(2048, num_classes)``."""
    spec = OrderedDict()
    spec['conv1.weight'] = (64, 3, 7, 7)
    _bn(spec, 'bn1', 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            p = 'layer%d.%d' % (li + 1, b)
            spec[p + '.conv1.weight'] = (planes, inplanes, 1, 1)
            _bn(spec, p + '.bn1', planes)
            spec[p + '.conv2.weight'] = (planes, planes, 3, 3)
            _bn(spec, p + '.bn2', planes)
            spec[p + '.conv3.weight'] = (planes * 4, planes, 1, 1)
            _bn(spec, p + '.bn3', planes * 4)
            if b == 0:
                spec[p + '.downsample.0.weight'] = (planes * 4, inplanes, 1, 1)
                _bn(spec, p + '.downsample.1', planes * 4)
            inplanes = planes * 4
    spec['fc.weight'] = (num_classes, 2048)
    spec['fc.bias'] = (num_classes,)
    return spec


VGG19_PREFIX = ((0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128))   # (features idx, cin, cout)


def vgg19_prefix_spec():
    """``vgg19().features`` layers 0..7 — the four convs whose outputs are conv_1..conv_4."""
    spec = OrderedDict()
    for idx, cin, cout in VGG19_PREFIX:
        spec['%d.weight' % idx] = (cout, cin, 3, 3)
        spec['%d.bias' % idx] = (cout,)
    return spec


def pggan_generator_spec():
    """name -> shape of ``model_256.Generator(511, 1).state_dict()`` (graphs/pggan/model_256.py:188-227; EqualLR keeps ``weight_orig``)."""
    from collections import OrderedDict
    chans = ((512, 512), (512, 512), (512, 512), (512, 512), (512, 256), (256, 128), (128, 64), (64, 32), (32, 16))
    spec = OrderedDict()
    spec['label_embed.weight'] = (1, 1)
    for i, (cin, cout) in enumerate(chans):
        spec['progression.%d.conv.0.conv.bias' % i] = (cout,)
        spec['progression.%d.conv.0.conv.weight_orig' % i] = (cout, cin, 4, 4) if i == 0 else (cout, cin, 3, 3)
        spec['progression.%d.conv.3.conv.bias' % i] = (cout,)
        spec['progression.%d.conv.3.conv.weight_orig' % i] = (cout, cout, 3, 3)
    for i, (cin, cout) in enumerate(chans):
        spec['to_rgb.%d.weight' % i] = (3, cout, 1, 1)
        spec['to_rgb.%d.bias' % i] = (3,)
    return spec
