"""Training the attribute regressor (reference scene_regressor_256.py:86-171): torchvision ResNet-50 v1.5 with
``fc = Linear(2048, 40)``, BatchNorm in TRAINING mode (the reference never calls ``model.eval()``: batch statistics, running
statistics updated with momentum 0.1 — in its test loop too), ``MSELoss(preds, label).mean()``, ``Adam(model.parameters(), lr=1e-4)``,
checkpoints ``{'model': state_dict, 'optm': optimizer.state_dict()}`` that ``constants.reg_path`` then loads for the walk training.

Everything heavy runs on libl2i_hip.so: the forward / input-gradient convolutions are the walk path's kernels (weights repacked on the
GPU after every optimiser step), the weight gradients ``l2i_conv2d_wgrad_f32`` and the training-mode BatchNorm passes ``l2i_bn_*``
(csrc/l2i_train.hip).  torch supplies tensors, the [C]-sized statistics algebra, the 2048x40 fc GEMMs and ``torch.optim.Adam`` (the
reference uses it too).  No autograd graph is built: the backward is scheduled by hand like the frozen networks' input-gradients.
"""
import csv
import os

import numpy as np
import torch

from . import _lib
from . import conv as C
from . import kernels as K
from .specs import RESNET50_LAYERS

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def conv_wgrad(x, gy, k, stride, pad):
    """dW [Cout,Cin,k,k] of a correlation y = conv(x, W, stride, pad) given gy = dL/dy."""
    lib = _lib.load()
    b, cin, h, w = x.shape
    _, cout, oh, ow = gy.shape
    dw = torch.zeros(cout, cin, k, k, device=x.device, dtype=torch.float32)
    _lib.check(lib.l2i_conv2d_wgrad_f32(_lib.fptr(dw), _lib.fptr(x), _lib.fptr(gy), b, cin, h, w, cout, oh, ow, k, k, stride, pad, pad,
                                        _lib.stream_ptr()), 'l2i_conv2d_wgrad_f32')
    return dw


class _BN:
    """BatchNorm2d parameters + the training-mode forward / backward on the l2i_bn_* kernels."""

    def __init__(self, P, name, device):
        t = lambda k: torch.as_tensor(np.asarray(P[name + '.' + k]), dtype=torch.float32).clone().to(device)
        self.weight, self.bias = t('weight'), t('bias')
        self.running_mean, self.running_var = t('running_mean'), t('running_var')
        self.num_batches_tracked = torch.as_tensor(np.asarray(P.get(name + '.num_batches_tracked', 0))).to(torch.long).to(device)
        self.name = name

    def forward(self, x, residual=None, relu=True):
        lib = _lib.load()
        b, c, h, w = x.shape
        n = b * h * w
        s = torch.zeros(2, c, device=x.device, dtype=torch.float64)
        _lib.check(lib.l2i_bn_stats_f32(_lib.ptr(s[0]), _lib.ptr(s[1]), _lib.fptr(x), b, c, h * w, _lib.stream_ptr()), 'l2i_bn_stats_f32')
        mean64 = s[0] / n
        var64 = (s[1] / n - mean64 * mean64).clamp_(min=0.0)                    # biased variance normalises (BatchNorm2d.forward)
        mean, invstd = mean64.float(), torch.rsqrt(var64 + BN_EPS).float()
        with torch.no_grad():                                                  # running statistics: momentum 0.1, UNBIASED variance
            self.running_mean.mul_(1 - BN_MOMENTUM).add_(mean, alpha=BN_MOMENTUM)
            self.running_var.mul_(1 - BN_MOMENTUM).add_((var64 * (n / max(n - 1, 1))).float(), alpha=BN_MOMENTUM)
            self.num_batches_tracked += 1
        scale = (self.weight * invstd).contiguous()
        shift = (self.bias - mean * scale).contiguous()
        y = torch.empty_like(x)
        _lib.check(lib.l2i_bn_apply_f32(_lib.fptr(y), _lib.fptr(x), _lib.fptr(scale), _lib.fptr(shift), _lib.fptr(residual), int(relu), b, c, h * w,
                                        _lib.stream_ptr()), 'l2i_bn_apply_f32')
        return y, (x, mean.contiguous(), invstd.contiguous())

    def backward(self, gy, saved, out_mask=None, want_masked=False):
        """gy = dL/d(output of relu(bn(x) [+ residual])); out_mask = that output when a ReLU followed.  Returns (dx, d weight, d bias,
        masked gy or None)."""
        lib = _lib.load()
        x, mean, invstd = saved
        b, c, h, w = x.shape
        n = b * h * w
        s = torch.zeros(2, c, device=x.device, dtype=torch.float64)
        _lib.check(lib.l2i_bn_bwd_reduce_f32(_lib.ptr(s[0]), _lib.ptr(s[1]), _lib.fptr(gy), _lib.fptr(out_mask), _lib.fptr(x), _lib.fptr(mean),
                                             _lib.fptr(invstd), b, c, h * w, _lib.stream_ptr()), 'l2i_bn_bwd_reduce_f32')
        d_bias, d_weight = s[0].float(), s[1].float()
        m_dy, m_dyxh = (s[0] / n).float().contiguous(), (s[1] / n).float().contiguous()
        dx = torch.empty_like(x)
        gm = torch.empty_like(x) if want_masked else None
        _lib.check(lib.l2i_bn_bwd_apply_f32(_lib.fptr(dx), _lib.fptr(gm), _lib.fptr(gy), _lib.fptr(out_mask), _lib.fptr(x), _lib.fptr(mean),
                                            _lib.fptr(invstd), _lib.fptr(self.weight), _lib.fptr(m_dy), _lib.fptr(m_dyxh), b, c, h * w,
                                            _lib.stream_ptr()), 'l2i_bn_bwd_apply_f32')
        return dx, d_weight, d_bias, gm

    def tensors(self):
        return [('weight', self.weight), ('bias', self.bias)]


class _Conv:
    """A trainable bias-free convolution: the weight lives on the GPU; ``plan()`` (re)builds the packed forward / input-gradient
    launches of the frozen-conv machinery from the current weight (called once per step, after the optimiser)."""

    def __init__(self, P, name, stride, padding, device):
        self.weight = torch.as_tensor(np.asarray(P[name + '.weight']), dtype=torch.float32).clone().to(device)
        self.stride, self.padding, self.k = stride, padding, self.weight.shape[2]
        self.name, self.fc = name, None

    def plan(self):
        self.fc = C.FrozenConv2d(self.weight.detach(), stride=self.stride, padding=self.padding, device=self.weight.device)

    def forward(self, x):
        return self.fc.forward(x)

    def dgrad(self, gy, in_hw):
        return self.fc.dgrad(gy, in_hw)

    def wgrad(self, x, gy):
        return conv_wgrad(x, gy, self.k, self.stride, self.padding)


class TrainableResNet50:
    """torchvision ResNet-50 v1.5 (stride on the 3x3, downsample on the first block of a layer, adaptive average pool), every parameter
    trainable, BatchNorm in training mode.  ``state_dict()`` / ``load_state_dict()`` use torchvision's key names, so the checkpoints are
    the ones ``graph.load_networks`` (and the reference) read."""

    def __init__(self, state, device='cuda', num_classes=40):
        P = state
        self.device = device
        self.stem, self.stem_bn = _Conv(P, 'conv1', 2, 3, device), _BN(P, 'bn1', device)
        self.blocks = []
        for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
            for b in range(blocks):
                p = 'layer%d.%d' % (li + 1, b)
                s = stride if b == 0 else 1
                self.blocks.append(dict(
                    c1=_Conv(P, p + '.conv1', 1, 0, device), b1=_BN(P, p + '.bn1', device),
                    c2=_Conv(P, p + '.conv2', s, 1, device), b2=_BN(P, p + '.bn2', device),
                    c3=_Conv(P, p + '.conv3', 1, 0, device), b3=_BN(P, p + '.bn3', device),
                    cd=_Conv(P, p + '.downsample.0', s, 0, device) if b == 0 else None,
                    bd=_BN(P, p + '.downsample.1', device) if b == 0 else None))
        self.fc_w = torch.as_tensor(np.asarray(P['fc.weight']), dtype=torch.float32).clone().to(device)
        self.fc_b = torch.as_tensor(np.asarray(P['fc.bias']), dtype=torch.float32).clone().to(device)
        assert self.fc_w.shape[0] == num_classes
        self.replan()

    # -- parameters (torchvision order: conv1, bn1, layer*.*, fc) ------------------------------------------------------------
    def named_parameters(self):
        out = [('conv1.weight', self.stem.weight)] + [('bn1.' + k, t) for k, t in self.stem_bn.tensors()]
        for blk in self.blocks:
            for c, b in (('c1', 'b1'), ('c2', 'b2'), ('c3', 'b3'), ('cd', 'bd')):
                if blk[c] is not None:
                    out.append((blk[c].name + '.weight', blk[c].weight))
                    out += [(blk[b].name + '.' + k, t) for k, t in blk[b].tensors()]
        return out + [('fc.weight', self.fc_w), ('fc.bias', self.fc_b)]

    def parameters(self):
        return [t for _, t in self.named_parameters()]

    def _bns(self):
        return [self.stem_bn] + [blk[b] for blk in self.blocks for b in ('b1', 'b2', 'b3', 'bd') if blk[b] is not None]

    def state_dict(self):
        sd = {k: t.detach().clone() for k, t in self.named_parameters()}
        for bn in self._bns():
            sd[bn.name + '.running_mean'], sd[bn.name + '.running_var'] = bn.running_mean.clone(), bn.running_var.clone()
            sd[bn.name + '.num_batches_tracked'] = bn.num_batches_tracked.clone()
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, t in self.named_parameters():
                t.copy_(torch.as_tensor(sd[k]).to(t.device))
            for bn in self._bns():
                bn.running_mean.copy_(torch.as_tensor(sd[bn.name + '.running_mean']).to(self.device))
                bn.running_var.copy_(torch.as_tensor(sd[bn.name + '.running_var']).to(self.device))
                if bn.name + '.num_batches_tracked' in sd:
                    bn.num_batches_tracked.copy_(torch.as_tensor(sd[bn.name + '.num_batches_tracked']).to(bn.num_batches_tracked.device))
        self.replan()

    def replan(self):
        """Repack every convolution from its current weight (after an optimiser step)."""
        self.stem.plan()
        for blk in self.blocks:
            for c in ('c1', 'c2', 'c3', 'cd'):
                if blk[c] is not None:
                    blk[c].plan()

    # -- forward / backward ----------------------------------------------------------------------------------------------------
    def forward(self, img, keep=True):
        """[B,3,H,W] -> [B,40] in training mode; ``keep``: save what ``backward`` needs."""
        x = img.detach().contiguous()
        z0 = self.stem.forward(x)
        a0, s0 = self.stem_bn.forward(z0)
        p0, idx0 = K.maxpool2d_fwd(a0, 3, 2, 1)
        saved = dict(img=x, s0=s0, a0=a0, idx0=idx0, blocks=[])
        cur = p0
        for blk in self.blocks:
            z1 = blk['c1'].forward(cur)
            y1, s1 = blk['b1'].forward(z1)
            z2 = blk['c2'].forward(y1)
            y2, s2 = blk['b2'].forward(z2)
            z3 = blk['c3'].forward(y2)
            if blk['cd'] is not None:
                zd = blk['cd'].forward(cur)
                idt, sd = blk['bd'].forward(zd, relu=False)
            else:
                idt, sd = cur, None
            out, s3 = blk['b3'].forward(z3, residual=idt, relu=True)
            saved['blocks'].append((cur, y1, s1, y2, s2, s3, sd, out))
            cur = out
        b, c, h, w = cur.shape
        feat = K.dot_reduce(cur) * (1.0 / (h * w))
        self._saved = dict(saved, feat=feat, last=(b, c, h, w)) if keep else None
        return torch.addmm(self.fc_b, feat, self.fc_w.t())

    __call__ = forward

    def backward(self, g_preds):
        """dL/d preds [B,40] -> {parameter name: gradient} (and nothing else: the image gets no gradient here)."""
        sv = self._saved
        grads = {'fc.weight': torch.mm(g_preds.t(), sv['feat']), 'fc.bias': g_preds.sum(0)}
        b, c, h, w = sv['last']
        g = (torch.mm(g_preds, self.fc_w) * (1.0 / (h * w))).reshape(b, c, 1, 1).expand(b, c, h, w).contiguous()
        for blk, saved in zip(reversed(self.blocks), reversed(sv['blocks'])):
            g = self.block_backward(blk, saved, g, grads)
        self.stem_backward(sv, g, grads)
        self._saved = None
        return grads

    @staticmethod
    def block_backward(blk, saved, g, grads):
        """One bottleneck block: g = dL/d(block output) -> dL/d(block input); parameter gradients are added to ``grads`` by name."""
        cur, y1, s1, y2, s2, s3, sd, out = saved
        in_hw = (cur.shape[2], cur.shape[3])
        # out = relu(bn3(z3) + idt): the masked gradient feeds bn3 and the identity branch
        g_z3, dw, db, gm = blk['b3'].backward(g, s3, out_mask=out, want_masked=True)
        grads[blk['b3'].name + '.weight'], grads[blk['b3'].name + '.bias'] = dw, db
        grads[blk['c3'].name + '.weight'] = blk['c3'].wgrad(y2, g_z3)
        g_y2 = blk['c3'].dgrad(g_z3, (y2.shape[2], y2.shape[3]))
        g_z2, dw, db, _ = blk['b2'].backward(g_y2, s2, out_mask=y2)
        grads[blk['b2'].name + '.weight'], grads[blk['b2'].name + '.bias'] = dw, db
        grads[blk['c2'].name + '.weight'] = blk['c2'].wgrad(y1, g_z2)
        g_y1 = blk['c2'].dgrad(g_z2, (y1.shape[2], y1.shape[3]))
        g_z1, dw, db, _ = blk['b1'].backward(g_y1, s1, out_mask=y1)
        grads[blk['b1'].name + '.weight'], grads[blk['b1'].name + '.bias'] = dw, db
        grads[blk['c1'].name + '.weight'] = blk['c1'].wgrad(cur, g_z1)
        g_in = blk['c1'].dgrad(g_z1, in_hw)
        if blk['cd'] is not None:
            g_zd, dw, db, _ = blk['bd'].backward(gm, sd)
            grads[blk['bd'].name + '.weight'], grads[blk['bd'].name + '.bias'] = dw, db
            grads[blk['cd'].name + '.weight'] = blk['cd'].wgrad(cur, g_zd)
            return K.axpby(g_in, blk['cd'].dgrad(g_zd, in_hw))
        return K.axpby(g_in, gm)

    def stem_backward(self, sv, g, grads):
        """maxpool <- relu <- bn1 <- conv1: g = dL/d(pooled map); the image itself gets no gradient."""
        a0 = sv['a0']
        g_a0 = K.maxpool2d_bwd(g, sv['idx0'], (a0.shape[2], a0.shape[3]), 3, 2, 1)
        g_z0, dw, db, _ = self.stem_bn.backward(g_a0, sv['s0'], out_mask=a0)
        grads['bn1.weight'], grads['bn1.bias'] = dw, db
        grads['conv1.weight'] = self.stem.wgrad(sv['img'], g_z0)


def mse_loss_and_grad(preds, label):
    """nn.MSELoss()(preds, label).mean() (scene_regressor_256.py:136,150) and its gradient w.r.t. preds."""
    d = preds - label
    return (d * d).mean(), d * (2.0 / d.numel())


def train_step(model, optimizer, data, label):
    """scene_regressor_256.py:147-153: forward, zero_grad, loss, backward, optimiser step.  Returns the loss (device scalar)."""
    preds = model(data)
    optimizer.zero_grad()
    loss, g = mse_loss_and_grad(preds, label.to(preds.device).float())
    grads = model.backward(g)
    for k, t in model.named_parameters():
        t.grad = grads[k]
    optimizer.step()
    model.replan()
    return loss


def make_optimizer(model, lr=1e-4):
    return torch.optim.Adam(model.parameters(), lr=lr)                          # scene_regressor_256.py:116


def save_ckpt(path, model, optimizer):
    """scene_regressor_256.py:166-170: {'model', 'optm'} — what constants.reg_path points at for the walk training."""
    torch.save({'model': {k: v.cpu() for k, v in model.state_dict().items()}, 'optm': optimizer.state_dict()}, path)


def load_ckpt(path, model, optimizer):
    ckpt = torch.load(path, map_location='cpu')
    model.load_state_dict(ckpt['model'])
    optimizer.load_state_dict(ckpt['optm'])
    return model, optimizer


def load_labelfile(path):
    """annotations.tsv: name <tab> 40 x 'value,confidence' (scene_regressor_256.py:67-74)."""
    labels = {}
    with open(path, 'r') as f:
        for line in csv.reader(f, delimiter='\t'):
            labels[line[0]] = np.array([float(i.split(',')[0]) for i in line[1:]])
    return labels


class CustomDataset:
    """The reference dataset (scene_regressor_256.py:27-64): images under folder/*/* listed in a split file; Resize(256) + CenterCrop(256)
    + ToTensor + Normalize(0.5, 0.5) with PIL and numpy (torchvision is not a dependency of this build)."""

    def __init__(self, folder_path, label_dict, split_file, image_size=256):
        import glob
        self.label_dict, self.image_size = label_dict, image_size
        with open(split_file, 'r') as f:
            split = set(i.strip() for i in f.readlines())
        self.image_list = [i for i in sorted(glob.glob(folder_path + '/*/*')) if '/'.join(i.split('/')[-2:]) in split]

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, index):
        from PIL import Image
        path = self.image_list[index]
        im = Image.open(path).convert('RGB')
        w, h = im.size
        # torchvision Resize(int): the short side becomes image_size, the long side int(image_size * long / short) (truncated, not rounded)
        size = self.image_size
        new = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
        im = im.resize(new, Image.BILINEAR)
        w, h = im.size
        l, t = (w - self.image_size) // 2, (h - self.image_size) // 2
        a = np.asarray(im.crop((l, t, l + self.image_size, t + self.image_size)), dtype=np.float32) / 255.0
        x = torch.from_numpy((a.transpose(2, 0, 1) - 0.5) / 0.5)
        return x, torch.Tensor(self.label_dict['/'.join(path.split('/')[-2:])])


def main(data_path='/transient_scene/imageAlignedLD/', label_path='/transient_scene/annotations/annotations.tsv',
         split_path='/transient_scene/training_test_splits/random_split/', n_epoch=500, batch_size=32, out_dir='./checkpoint_256', state=None):
    """The reference script's __main__ (scene_regressor_256.py:86-171) without tensorboard / tqdm: train, evaluate (in training mode, like the
    reference), save a checkpoint per epoch."""
    from . import synth
    os.makedirs(out_dir, exist_ok=True)
    labels = load_labelfile(label_path)
    train = CustomDataset(data_path, labels, split_path + 'training.txt')
    test = CustomDataset(data_path, labels, split_path + 'test.txt')
    # the reference starts from torchvision's ImageNet weights (a download); offline the caller passes a state dict, or seeded random init
    model = TrainableResNet50(state if state is not None else synth.resnet50_state(seed=300), device='cuda')
    optimizer = make_optimizer(model)
    for epoch in range(n_epoch):
        order = np.random.permutation(len(train))
        for i in range(0, len(order), batch_size):
            batch = [train[j] for j in order[i:i + batch_size]]
            data = torch.stack([b[0] for b in batch]).cuda()
            label = torch.stack([b[1] for b in batch]).cuda()
            loss = train_step(model, optimizer, data, label)
        if epoch != 0:
            tl = []
            for i in range(0, len(test), batch_size):
                batch = [test[j] for j in range(i, min(len(test), i + batch_size))]
                preds = model(torch.stack([b[0] for b in batch]).cuda(), keep=False)
                tl.append(float(mse_loss_and_grad(preds, torch.stack([b[1] for b in batch]).cuda())[0]))
            print('Test epoch %d; Loss: %.5f' % (epoch, float(np.mean(tl))))
        save_ckpt(os.path.join(out_dir, '%s_dict.model' % str(epoch + 1).zfill(3)), model, optimizer)
    return model
