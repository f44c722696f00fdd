"""Tolerance study of the 16-bit path (conv.PRECISION = 'bf16': bf16 h8 feature maps, one bf16 MFMA per MAC; [r5] 'f16': IEEE fp16 elements with
static power-of-two gradient scales) against the float64 CPU oracle on the same z / weights: per network (forward and input gradient) and for
whole training steps at 64^2 (batch 4), 256^2 (batch 2), 1024^2 (batch 1).  Prints one JSON line per case; profiles/r0N_{bf16,fp16}_tolerance.json
are this script's outputs on the MI355X.
usage: python tools/bf16_study.py [sizes] [--precision bf16|f16] [--noise_strength S]
       python tools/bf16_study.py [sizes] --probe [--precision ...] [--batch B]: no oracle; one training step with latent2im_amd.nets16.PROBE set:
       max / median magnitude (and zero fraction) of the forward and gradient maps of every branch — what the fp16 gradient scales are read from."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from latent2im_amd import conv, selfcheck, synth
from oracle import nets as onets, nets16 as onets16, sg2, step as ostep

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
DEV = 'cuda'
import oracle
torch.set_num_threads(min(32, oracle.host_cpus()))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


def l2rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def networks(size, batch):
    from latent2im_amd import nets16
    conv.PRECISION = PRECISION
    # (fp16: the networks are driven by unit-scale random upstream gradients here, not by the loss — built stand-alone they carry no scaler: no scaling)
    out = dict(case='networks', precision=PRECISION, size=size, batch=batch)
    dt = torch.float64
    rs = np.random.RandomState(size)
    stG = synth.generator_state(size, seed=100, noise_strength=0.05)
    G = nets16.Generator(stG, size, device=DEV)
    PG = ostep.to_torch(stG, dt)
    lat = rs.randn(batch, G.n_latent, 512)
    noise = synth.noise_maps(size, batch)
    gy = rs.randn(batch, 3, size, size)
    lg = T(lat).float().to(DEV).requires_grad_(True)
    img = G.synthesis(lg, [T(n).to(DEV) for n in noise])
    img.backward(T(gy).float().to(DEV))
    lo = T(lat).requires_grad_(True)
    img_o = sg2.generator_synthesis(PG, lo, [T(n).double() for n in noise])
    go, = torch.autograd.grad(img_o, lo, T(gy))
    out.update(G_img_relmax=rel(img, img_o), G_img_l2=l2rel(img, img_o), G_grad_cos=cosine(lg.grad, go), G_grad_l2=l2rel(lg.grad, go))
    x = img_o.detach()
    for name, net, fwd in (('R', nets16.ResNet50(synth.resnet50_state(seed=300), device=DEV), lambda P, t: onets.resnet50_forward(P, t)),
                           ('D', nets16.Discriminator(synth.discriminator_state(size, seed=200), size, device=DEV), lambda P, t: sg2.discriminator_forward(P, t))):
        P = ostep.to_torch(synth.resnet50_state(seed=300) if name == 'R' else synth.discriminator_state(size, seed=200), dt)
        xg = x.float().to(DEV).requires_grad_(True)
        y = net(xg)
        gyy = T(rs.randn(*y.shape))
        y.backward(gyy.float().to(DEV))
        xo = x.clone().requires_grad_(True)
        yo = fwd(P, xo)
        gxo, = torch.autograd.grad(yo, xo, gyy)
        out.update({name + '_out_relmax': rel(y, yo), name + '_grad_cos': cosine(xg.grad, gxo), name + '_grad_l2': l2rel(xg.grad, gxo)})
        if name == 'R':
            # ... and against the float64 oracle WITH the path's bf16 storage rounding restated (oracle/nets16.py): what is left is kernel arithmetic
            # (order of fp32 sums, bf16 gradient maps); R16fmt_* = that oracle against the exact one = the price of the storage format alone
            xq = x.clone().requires_grad_(True)
            yq = onets16.resnet50_forward_bf16(P, xq)
            gxq, = torch.autograd.grad(yq, xq, gyy)
            out.update(R16_out_relmax=rel(y, yq), R16_grad_cos=cosine(xg.grad, gxq), R16_grad_l2=l2rel(xg.grad, gxq), R16_grad_relmax=rel(xg.grad, gxq),
                       R16fmt_out_relmax=rel(yq, yo), R16fmt_grad_cos=cosine(gxq, gxo), R16fmt_grad_l2=l2rel(gxq, gxo))
            # R16floor_*: the storage-rounding oracle against ITSELF with every value perturbed by 1e-7 relative before it is rounded — the size of an fp32
            # summation-order difference.  A feature map that lands on the other side of a bf16 rounding boundary moves by 2^-9, fifty layers amplify
            # it, masks flip: no two fp32-accumulating implementations of this network agree better than this, so the kernels are held to the floor
            # (x 1.35), not to zero
            torch.manual_seed(1)
            xp = x.clone().requires_grad_(True)
            yp = onets16.resnet50_forward_bf16(P, xp, perturb=1e-7)
            gxp, = torch.autograd.grad(yp, xp, gyy)
            out.update(R16floor_out_relmax=rel(yp, yq), R16floor_grad_cos=cosine(gxp, gxq), R16floor_grad_l2=l2rel(gxp, gxq))
    stV = synth.vgg19_prefix_state(seed=400)
    V = nets16.VGG19Prefix(stV, device=DEV)
    other = torch.roll(x, 5, 3)
    xg = x.float().to(DEV).requires_grad_(True)
    losses = V.content_losses(other.float().to(DEV), xg)
    losses.sum().backward()
    xo = x.clone().requires_grad_(True)
    _, lo_ = ostep.content_loss(ostep.to_torch(stV, dt), other, xo)
    gxo, = torch.autograd.grad(sum(lo_), xo)
    out.update(V_loss_rel=[float(abs(float(a) - float(b)) / float(b)) for a, b in zip(losses, lo_)], V_grad_cos=cosine(xg.grad, gxo), V_grad_l2=l2rel(xg.grad, gxo))
    return out


PRECISION = 'bf16'


def probe(size, batch, attrs, clamp, transform='face'):
    from latent2im_amd import constants, nets16
    conv.PRECISION = PRECISION
    gr = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform=transform)
    zs = synth.z_sample(batch, seed=21)
    rs = np.random.RandomState(22)
    alpha = np.ones((batch, len(attrs))) * (rs.uniform(-1, 1, len(attrs)) if clamp else rs.uniform(0, 1, len(attrs)))
    nets16.PROBE = []
    try:
        r = selfcheck.run_step(gr, zs, alpha, clamp=clamp, optimize=False)
        torch.cuda.synchronize()
    finally:
        rows, nets16.PROBE = nets16.PROBE, None
    print('== %s %d^2 batch %d, %d attrs: loss %.6f, walk-gradient max %.3e, finite %s; scales (log2) %s' % (
        PRECISION, size, batch, len(attrs), float(r['loss']), float(r['grad'].abs().max()), bool(torch.isfinite(r['grad']).all() and torch.isfinite(r['x1']).all()),
        gr.loss_scaler.log2 if gr.loss_scaler is not None else '-'))
    for tag, shape, mx, med, zf in rows:
        print('  %-16s %-24s max %.3e (2^%6.1f)  median %.3e (2^%6.1f)  zeros %.3f' % (tag, 'x'.join(str(v) for v in shape), mx, np.log2(mx) if mx > 0 else -999,
                                                                                    med, np.log2(med) if med > 0 else -999, zf))
    constants.resolution, constants.BATCH_SIZE = 256, 4


def step(size, batch, attrs, clamp, transform='face', cached=None, hipgraph=False):
    """One training step on the 16-bit path against the float64 oracle.  ``cached``: a tests.oracle_cache.Cached case holding that oracle evaluation
    (the GPU tests pass the committed one instead of spending 8 - 36 s of CPU per case; image errors are then taken at its 4096 probe pixels).
    ``hipgraph``: forward + backward REPLAYED from one hipGraph (capture.CapturedStep), the way bench.py --config c5 runs the step."""
    from latent2im_amd import capture, constants
    conv.PRECISION = PRECISION
    gr = selfcheck.build_graph(size, attrs, batch, lr=1e-3, transform=transform)
    zs = synth.z_sample(batch, seed=21)
    rs = np.random.RandomState(22)
    alpha = np.ones((batch, len(attrs))) * (rs.uniform(-1, 1, len(attrs)) if clamp else rs.uniform(0, 1, len(attrs)))
    if hipgraph:
        cs = capture.CapturedStep(gr, batch, len(attrs), clamp=clamp)
        r = cs(zs, alpha, optimize=False)
        r = cs(zs, alpha, optimize=False)                   # the second replay: the graph holds no state of its own
    else:
        r = selfcheck.run_step(gr, zs, alpha, clamp=clamp, optimize=False)
    torch.cuda.synchronize()
    dt = torch.float64
    idx = gr.attrIdx
    pg = gr.regressor(r['x1'])[:, idx].double().cpu()
    if cached is not None:
        o = cached
        po = o['po'].double()
        img = dict(x0_relmax=o.image_rel_to_max(r['x0'], 'x0'), x1_relmax=o.image_rel_to_max(r['x1'], 'x1'))
    else:
        nets = dict(G=ostep.to_torch(synth.generator_state(size, seed=100), dt), D=ostep.to_torch(synth.discriminator_state(size, seed=200), dt),
                    R=ostep.to_torch(synth.resnet50_state(seed=300), dt), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400), dt))
        o = ostep.train_step_bounded(nets, T(synth.walk_init(len(attrs), gr.module.netG.n_latent, seed=7)).to(dt), T(zs).to(dt), T(alpha).to(dt), gr.attrIdx, clamp_variant=clamp)
        po = onets.resnet50_forward(nets['R'], o['x1'])[:, idx].double()
        img = dict(x0_relmax=rel(r['x0'], o['x0']), x1_relmax=rel(r['x1'], o['x1']), x1_l2=l2rel(r['x1'], o['x1']))
    tgt = o['target'].double()
    per_attr = lambda p: -(tgt * p.clamp(min=1e-12).log() + (1 - tgt) * (1 - p).clamp(min=1e-12).log()).mean(0)
    out = dict(case='step', precision=PRECISION, size=size, batch=batch, attrs=len(attrs), clamp=clamp, hipgraph=bool(hipgraph), finite=bool(torch.isfinite(r['grad']).all() and torch.isfinite(r['x1']).all()),
               **img,
               a0_absmax=float((r['a0'].double().cpu() - o['alpha_org']).abs().max()), eps_absmax=float((r['eps'].double().cpu() - o['eps']).abs().max()),
               reg_rel=abs(float(r['terms']['reg']) - float(o['reg'])) / abs(float(o['reg'])), cont_rel=abs(float(r['terms']['cont']) - float(o['cont'])) / abs(float(o['cont'])),
               gan_rel=abs(float(r['terms']['gan']) - float(o['gan'])) / abs(float(o['gan'])), loss_rel=abs(float(r['loss']) - float(o['loss'])) / abs(float(o['loss'])),
               per_attr_reg_loss_delta=[float(v) for v in (per_attr(pg) - per_attr(po)).abs()], per_attr_reg_loss=[float(v) for v in per_attr(po)],
               grad_cos=cosine(r['grad'], o['grad']), grad_l2=l2rel(r['grad'], o['grad']), grad_relmax=rel(r['grad'], o['grad']))
    constants.resolution, constants.BATCH_SIZE = 256, 4
    return out


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('sizes', nargs='?', default='64,256,1024')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--probe', action='store_true')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--noise_strength', type=float, default=None, help='generator NoiseInjection weights of the synthetic state (default: constants.SYNTH_NOISE_STRENGTH)')
    a = ap.parse_args()
    PRECISION = a.precision
    if a.noise_strength is not None:
        from latent2im_amd import constants
        constants.SYNTH_NOISE_STRENGTH = a.noise_strength
    sizes = [int(s) for s in a.sizes.split(',')]
    if a.probe:
        for size in sizes:
            batch = a.batch or {64: 4, 256: 2, 1024: 1}.get(size, 1)
            probe(size, batch, ['Smiling'], False)
            probe(size, batch, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], True, 'scene')
        sys.exit(0)
    for size in sizes:
        batch = {64: 4, 256: 2, 1024: 1}.get(size, 1)
        if size <= 256:
            print(json.dumps(networks(size, min(batch, 2))), flush=True)
        print(json.dumps(step(size, batch, ['Smiling'], False)), flush=True)
        print(json.dumps(step(size, batch, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], True, 'scene')), flush=True)
