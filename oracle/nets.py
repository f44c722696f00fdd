"""Oracle: the two third-party frozen networks of the path — torchvision ResNet-50 (v0.5.0, fc->40) and the
VGG-19 ``features`` prefix — restated from the published architectures because their
source is not under /root/reference and torchvision is not installed.  PARITY UNPINNED (see oracle/__init__.py).
TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F

RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def _bn(P, prefix, x):
    """BatchNorm2d in eval mode: (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias."""
    return F.batch_norm(x, P[prefix + '.running_mean'].to(x.dtype), P[prefix + '.running_var'].to(x.dtype),
                        P[prefix + '.weight'], P[prefix + '.bias'], training=False, eps=1e-5)


def resnet50_forward(P, x):
    """Reference call sites transform_base.py:396-403,416-424 (regressor(img)); eval mode (:267).
    The image is consumed as-is: no resize, no renormalisation of the [-1,1] generator output."""
    x = F.conv2d(x, P['conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(P, 'bn1', x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            p = 'layer%d.%d' % (li + 1, b)
            s = stride if b == 0 else 1
            idt = x
            o = F.relu(_bn(P, p + '.bn1', F.conv2d(x, P[p + '.conv1.weight'])))
            o = F.relu(_bn(P, p + '.bn2', F.conv2d(o, P[p + '.conv2.weight'], stride=s, padding=1)))   # v1.5: stride on 3x3
            o = _bn(P, p + '.bn3', F.conv2d(o, P[p + '.conv3.weight']))
            if b == 0:
                idt = _bn(P, p + '.downsample.1', F.conv2d(x, P[p + '.downsample.0.weight'], stride=s))
            x = F.relu(o + idt)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, P['fc.weight'], P['fc.bias'])


VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def vgg19_taps(P, img):
    """Normalization (transform_base.py:44-54) applied to the raw [-1,1] image, then ``features`` layers 0..7;
    returns the four PRE-ReLU conv outputs conv_1..conv_4 (transform_base.py:426-454)."""
    mean = torch.tensor(VGG_MEAN, dtype=img.dtype).reshape(1, 3, 1, 1)
    std = torch.tensor(VGG_STD, dtype=img.dtype).reshape(1, 3, 1, 1)
    x = (img - mean) / std
    c1 = F.conv2d(x, P['0.weight'], P['0.bias'], padding=1)
    c2 = F.conv2d(F.relu(c1), P['2.weight'], P['2.bias'], padding=1)
    c3 = F.conv2d(F.max_pool2d(F.relu(c2), 2, 2), P['5.weight'], P['5.bias'], padding=1)
    c4 = F.conv2d(F.relu(c3), P['7.weight'], P['7.bias'], padding=1)
    return [c1, c2, c3, c4]


def resnet50_train_forward(P, x, momentum=0.1):
    """The same network with BatchNorm in TRAINING mode (scene_regressor_256.py:111-153 never calls ``model.eval()``): batch statistics
    normalise, ``running_mean`` / ``running_var`` of ``P`` are updated in place (momentum 0.1, unbiased variance), differentiable w.r.t.
    every floating-point entry of ``P``."""
    def bn(prefix, t):
        return F.batch_norm(t, P[prefix + '.running_mean'], P[prefix + '.running_var'], P[prefix + '.weight'], P[prefix + '.bias'],
                            training=True, momentum=momentum, eps=1e-5)
    x = F.conv2d(x, P['conv1.weight'], stride=2, padding=3)
    x = F.relu(bn('bn1', x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            p = 'layer%d.%d' % (li + 1, b)
            s = stride if b == 0 else 1
            idt = x
            o = F.relu(bn(p + '.bn1', F.conv2d(x, P[p + '.conv1.weight'])))
            o = F.relu(bn(p + '.bn2', F.conv2d(o, P[p + '.conv2.weight'], stride=s, padding=1)))
            o = bn(p + '.bn3', F.conv2d(o, P[p + '.conv3.weight']))
            if b == 0:
                idt = bn(p + '.downsample.1', F.conv2d(x, P[p + '.downsample.0.weight'], stride=s))
            x = F.relu(o + idt)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, P['fc.weight'], P['fc.bias'])


def resnet50_train_step(P, data, label, lr=1e-4, steps=1):
    """scene_regressor_256.py:147-153 restated: preds = model(data); loss = MSELoss()(preds, label).mean(); backward; Adam(lr).step() over every
    parameter.  ``P`` (tensors) is updated in place; returns (losses, gradients of the LAST step by name)."""
    names = [k for k, v in P.items() if v.dtype.is_floating_point and not k.endswith('running_mean') and not k.endswith('running_var')]
    for k in names:
        P[k].requires_grad_(True)
    opt = torch.optim.Adam([P[k] for k in names], lr=lr)
    losses, grads = [], {}
    for _ in range(steps):
        preds = resnet50_train_forward(P, data)
        opt.zero_grad()
        loss = F.mse_loss(preds, label).mean()
        loss.backward()
        grads = {k: P[k].grad.detach().clone() for k in names}
        opt.step()
        losses.append(loss.detach())
    for k in names:
        P[k].requires_grad_(False)
    return losses, grads
