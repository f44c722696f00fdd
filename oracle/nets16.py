"""Oracle for the 16-bit path's ResNet-50: the float64 network of oracle/nets.py with the STORAGE ROUNDING of latent2im_amd/nets16.py restated
— eval-mode BatchNorm folded into the conv weights, the folded weights of every conv rounded to bfloat16, the image rounded where the stem kernel reads it, and every feature map
rounded to bfloat16 where the GPU path stores it (after the stem's ReLU, after each conv's bias / residual / ReLU epilogue) — while every sum
stays float64.  The rounding is a straight-through estimator for autograd, so the gradient is the exact gradient of the piecewise-linear network
whose ReLU / max-pool masks come from the ROUNDED activations: what the GPU path computes, up to the order of its fp32 sums and the bf16 rounding
of its gradient maps (unbiased, 2^-9 relative per map).

Why it exists (round 4): against the exact float64 oracle the 16-bit ResNet-50's input gradient has a cosine of only 0.95 - 0.97 — fifty ReLU layers,
and a feature map that is 2^-9 off flips the masks of the units near zero.  That number measures bf16 STORAGE, not the kernels, and a bound loose
enough to hold it (0.95) would also hold a wrong halo column.  Against THIS oracle the same gradient must agree to a cosine > 0.999, so the test
has teeth; the distance between this oracle and the exact one is reported beside it as the price of the format.

TEST INFRASTRUCTURE ONLY (tests/, tools/bf16_study.py).  Reference call sites as oracle/nets.py (transform_base.py:396-403, 416-424).
"""
import torch
import torch.nn.functional as F

from .nets import RESNET50_LAYERS


def q(t, perturb=0.0):
    """Round to bfloat16 (through float32, as the GPU epilogue rounds its fp32 accumulator) with a straight-through gradient.  ``perturb``: relative
    Gaussian noise on the value BEFORE it is rounded (1e-7 = an fp32 summation-order difference): the sensitivity study of tools/bf16_study.py."""
    v = t.detach()
    if perturb:
        v = v * (1.0 + perturb * torch.randn_like(v))
    return t + (v.float().to(torch.bfloat16).to(t.dtype) - t.detach())


def _fold(P, conv_name, bn_name, round_w=True, eps=1e-5):
    w = P[conv_name + '.weight'].double()
    k = P[bn_name + '.weight'].double() / torch.sqrt(P[bn_name + '.running_var'].double() + eps)
    wf = (w * k.reshape(-1, 1, 1, 1)).float()                      # latent2im_amd/regressor.py:_fold_bn hands fp32 to the packers
    b = (P[bn_name + '.bias'].double() - P[bn_name + '.running_mean'].double() * k).float()
    if round_w:
        wf = wf.to(torch.bfloat16).float()
    return wf.double(), b.double()


def resnet50_forward_bf16(P, x, perturb=0.0):
    """[B,3,H,W] float64 -> [B,40]: storage rounding of nets16._ResNet16Fn.forward, float64 sums."""
    _q = q
    if perturb:
        _q = lambda t: q(t, perturb)
    return _resnet50_q(P, x, _q)


def _resnet50_q(P, x, q):
    w, b = _fold(P, 'conv1', 'bn1')                                # [r5] the 7x7 stem runs on l2i_conv_img_h8: 16-bit weights, the image rounded to 16 bits
    x = q(F.relu(F.conv2d(q(x), w, b, stride=2, padding=3)))       # inside the kernel, the h8 store of its epilogue (rounds 1-4: fp32 kernels + cast_to_h8)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for bi in range(blocks):
            p = 'layer%d.%d' % (li + 1, bi)
            s = stride if bi == 0 else 1
            w1, b1 = _fold(P, p + '.conv1', p + '.bn1')
            w2, b2 = _fold(P, p + '.conv2', p + '.bn2')
            w3, b3 = _fold(P, p + '.conv3', p + '.bn3')
            y1 = q(F.relu(F.conv2d(x, w1, b1)))
            y2 = q(F.relu(F.conv2d(y1, w2, b2, stride=s, padding=1)))
            idt = x
            if bi == 0:
                wd, bd = _fold(P, p + '.downsample.0', p + '.downsample.1')
                idt = q(F.conv2d(x, wd, bd, stride=s))
            x = q(F.relu(F.conv2d(y2, w3, b3) + idt))
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)                      # fp32 sums on the GPU (kernels16.dot_reduce)
    return F.linear(x, P['fc.weight'].double(), P['fc.bias'].double())
