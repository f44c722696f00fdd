"""The training drivers' loop (reference train.py:25-134 and train_multi_attr.py:43-231), single process or one
process per GPU.  Call order per iteration is the reference's: get_w -> get_logits -> get_reg_preds -> get_train_alpha ->
get_alphas -> get_w_new_tensor -> get_logits -> optimizeParametersAll.  Under data parallelism every rank draws the same
z (seed = epoch) and the same alpha (same numpy seed) and takes every world-th sample of the batch (dist.shard: strided shards keep the discriminator's stddev groups whole)."""
import logging
import math
import os
import time

import numpy as np
import torch

from . import constants, dist, hostutil


def _configure_logging(path, append=False):
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    for h in list(root.handlers):
        root.removeHandler(h)
    fmt = logging.Formatter('%(asctime)s %(levelname)s %(message)s')
    fh = logging.FileHandler(path, mode='a' if append else 'w')
    fh.setFormatter(fmt)
    root.addHandler(fh)
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    root.addHandler(sh)


def make_samples(img_tensor, output_dir, epoch, optim_iter, batch_size, pre_path='results', name='test'):
    """PNG grid of a batch (train.py:137-144); off the timed path (every save_freq iterations)."""
    from PIL import Image
    img = img_tensor.detach().cpu().numpy()
    img = np.uint8(np.clip(((img + 1) / 2.0) * 255, 0, 255)).transpose(0, 2, 3, 1)
    cols = max(int(math.sqrt(batch_size)), 1)
    rows = (img.shape[0] + cols - 1) // cols
    h, w = img.shape[1], img.shape[2]
    grid = np.zeros((rows * h, cols * w, 3), np.uint8)
    for i in range(img.shape[0]):
        r, c = divmod(i, cols)
        grid[r * h:(r + 1) * h, c * w:(c + 1) * w] = img[i]
    Image.fromarray(grid).save('{}/{}/{}_{}_{}.png'.format(output_dir, pre_path, epoch, optim_iter, name))


def train_step(graphs, zs_batch, attrList, layers=None, trainEmbed=False, updateGAN=False, opt=None, multi_attr=False):
    """One iteration of the hot loop.  ``zs_batch``: this rank's [B_local, 512] numpy slice.  Returns
    (loss tensor, alpha_for_target, out_zs, transformed_output)."""
    if opt is not None and getattr(opt, 'hip_graph', False) and not trainEmbed and zs_batch.shape[0] == _captured_batch(graphs, zs_batch):
        # --hip_graph: the same calls, recorded once (capture.CapturedStep) and replayed; alpha is still drawn on the host
        alpha_for_graph, alpha_for_target, _ = graphs.get_train_alpha(zs_batch, N_attr=len(attrList), trainEmbed=trainEmbed)
        r = graphs._captured_step(zs_batch, alpha_for_graph)
        return r['loss'], alpha_for_target, r['x0'], r['x1']
    z_global = torch.Tensor(zs_batch).to(graphs.device)                       # train.py:56
    w_global = graphs.get_w(z_global)                                          # :62
    out_zs = graphs.get_logits({'z': z_global, 'w': w_global})                 # :66
    if not (opt and opt.no_content_loss) and hasattr(graphs, 'prefetch_content_taps'):
        graphs.prefetch_content_taps(out_zs)                                   # the VGG taps of the original start beside the regressor / second generator pass
    alpha_org = graphs.get_reg_preds(out_zs)                                   # :69
    alpha_for_graph, alpha_for_target, _ = graphs.get_train_alpha(zs_batch, N_attr=len(attrList), trainEmbed=trainEmbed)
    ag = torch.tensor(alpha_for_graph).float().to(graphs.device)               # :84-85
    if multi_attr:                                                             # train_multi_attr.py:113,133
        alpha_target, epsilon = graphs.get_alphas_clamped(alpha_org, ag)
    else:                                                                      # train.py:86
        alpha_target, epsilon = ag, graphs.get_alphas(alpha_org, ag)
    w_new = graphs.get_w_new_tensor(w_global, epsilon, layers=layers)          # :89
    transformed_output = graphs.get_logits({'z': z_global, 'w': w_new})        # :94
    feed_dict = {'w': w_new, 'org': out_zs, 'logit': transformed_output, 'alpha': alpha_target}
    loss = graphs.optimizeParametersAll(feed_dict, trainEmbed=trainEmbed, updateGAN=updateGAN,
                                        no_content_loss=bool(opt and opt.no_content_loss),
                                        no_gan_loss=bool(opt and opt.no_gan_loss))
    return loss, alpha_for_target, out_zs, transformed_output


def _captured_batch(graphs, zs_batch):
    """Build the hipGraph of the step on first use (the local batch size is static from then on; a ragged last batch runs eagerly)."""
    st = getattr(graphs, '_captured_step', None)
    if st is None:
        from . import capture
        o = graphs._captured_opts
        st = graphs._captured_step = capture.CapturedStep(graphs, zs_batch.shape[0], len(o['attrList']), clamp=o['multi_attr'], layers=o['layers'],
                                                          no_content_loss=o['no_content_loss'], no_gan_loss=o['no_gan_loss'])
    return st.z.shape[0]


def train(graphs, graph_inputs, output_dir, attrList, layers=None, save_freq=100, trainEmbed=False, updateGAN=False,
          opt=None, multi_attr=False):
    """train.py:25-134.  Side effect on the host process, undone on return: after the first step every Python object alive is moved to the
    garbage collector's permanent generation (``gc.freeze``: a full collection over the networks' ~270 000 host objects takes 73 ms on the thread
    that launches the step's kernels); ``gc.unfreeze()`` runs in a ``finally`` when training ends, so a caller that drops
    this graph and builds another (notebooks, eval, tests) keeps a working cycle collector.  ``opt.no_gc_freeze`` switches the freeze off."""
    try:
        return _train(graphs, graph_inputs, output_dir, attrList, layers, save_freq, trainEmbed, updateGAN, opt, multi_attr)
    finally:
        import gc
        gc.unfreeze()


def _train(graphs, graph_inputs, output_dir, attrList, layers, save_freq, trainEmbed, updateGAN, opt, multi_attr):
    is_main = dist.rank() == 0
    if is_main:
        os.makedirs(os.path.join(output_dir, 'results'), exist_ok=True)
        _configure_logging(os.path.join(output_dir, 'log.txt'), append=False)
        logging.info('weight sources: {}'.format(getattr(graphs, 'weight_sources', None)))
    graphs._captured_opts = dict(attrList=attrList, multi_attr=multi_attr, layers=None if layers is None else [int(l) for l in layers],
                                 no_content_loss=bool(opt and opt.no_content_loss), no_gan_loss=bool(opt and opt.no_gan_loss))
    n_epoch = (opt.n_epoch if opt is not None and getattr(opt, 'n_epoch', None) else (3 if multi_attr else 10))
    batch_size = constants.BATCH_SIZE
    num_samples = graph_inputs['z'].shape[0]
    sl = dist.shard(batch_size)
    sync_log = not (opt is not None and getattr(opt, 'no_log_sync', False))
    loss_values = []
    for epoch in range(n_epoch):
        if updateGAN:
            raise NotImplementedError('ERROR: jointly training is not implemented yet')      # train.py:40-41
        iters = num_samples // batch_size
        if opt is not None and getattr(opt, 'max_iters', None):
            iters = min(iters, opt.max_iters)
        graph_inputs = hostutil.graph_input(graphs, num_samples, seed=epoch)                 # train.py:45
        print('Number of the training epochs and iterations: ', n_epoch, iters)
        for i in range(iters):
            batch_start = i * batch_size
            start_time = time.time()
            s = slice(batch_start, min(num_samples, batch_start + batch_size))
            zs_batch = hostutil.batch_input(graph_inputs, s)['z'][sl]
            loss, at, out_zs, transformed = train_step(graphs, zs_batch, attrList, layers, trainEmbed, updateGAN, opt, multi_attr)
            if epoch == 0 and i == 0 and not (opt is not None and getattr(opt, 'no_gc_freeze', False)):
                from . import capture
                capture.freeze_host_objects()                 # everything built lazily by the first step is long-lived: out of the collector's way
            if sync_log:
                curr = loss.detach().cpu().item()                                             # train.py:110
                loss_values.append(curr)
                if is_main:
                    logging.info('T, epc, bst, lss, alpha: {}, {}, {}, {}, {}'.format(
                        time.time() - start_time, epoch, batch_start, loss, round(float(at[0]), 2)))
            if is_main and (i % save_freq == 0):
                make_samples(out_zs, output_dir, epoch, i * batch_size, batch_size, name='org_%.2f' % (round(float(at[0]), 2)))
                make_samples(transformed, output_dir, epoch, i * batch_size, batch_size, name='logit_%.2f' % (round(float(at[0]), 2)))
        graphs.save_multi_models('{}/model_w_{}'.format(output_dir, epoch), '{}/model_gan_{}.ckpt'.format(output_dir, epoch),
                                 trainEmbed=trainEmbed, updateGAN=updateGAN)
    graphs.save_multi_models('{}/model_w_{}_final'.format(output_dir, n_epoch), '{}/model_gan_{}_final.ckpt'.format(output_dir, n_epoch),
                             trainEmbed=trainEmbed, updateGAN=updateGAN)
    if multi_attr and is_main and loss_values:
        np.save(os.path.join(output_dir, 'loss_values.npy'), np.asarray(loss_values))        # train_multi_attr.py:226-231
    return loss_values


def main(multi_attr=False, argv=None):
    from . import graph as graph_mod
    from .options import TrainOptions
    opt = TrainOptions().parse(print_opt=(int(os.environ.get('RANK', '0')) == 0), argv=argv)
    dist.select_gpu(opt.gpu)                         # train.py:150 — before the first torch.cuda call (init_from_env makes it)
    rk, world, local = dist.init_from_env()
    dist.assert_distinct_devices()                   # N ranks over RCCL must sit on N distinct physical GPUs
    if opt.synthetic_weights:
        constants.ALLOW_SYNTHETIC_WEIGHTS = True
    if opt.precision:
        from . import conv
        conv.PRECISION = opt.precision
    if opt.resolution:
        constants.resolution = opt.resolution
    if opt.batch_size:
        constants.BATCH_SIZE = opt.batch_size
    if opt.seed is not None:
        np.random.seed(opt.seed)
    elif world > 1:
        np.random.seed(1234)                         # identical walk init and alpha stream on every rank
    graph_kwargs = hostutil.set_graph_kwargs(opt)
    model = graph_mod.find_model_using_name(opt.model, opt.transform)
    g = model(**graph_kwargs)
    if world > 1:                                    # one source of truth for the trainable state
        dist.broadcast_parameters(g.walk.parameters())
    if rk == 0:                                      # which frozen weights this run trained against: stdout, opt.yml (and log.txt in train())
        print('weight sources: ', g.weight_sources)
        yml = os.path.join(opt.output_dir, 'opt.yml')
        if os.path.isfile(yml):
            import yaml
            with open(yml) as f:
                dump = yaml.load(f, Loader=yaml.FullLoader)
            dump['weight_sources'] = dict(g.weight_sources)
            with open(yml, 'wt') as f:
                yaml.dump(dump, f, default_flow_style=False)
    graph_inputs = hostutil.graph_input(g, opt.num_samples, seed=0)
    attrList = graph_kwargs['attrList']
    print('attrlist: ', attrList)
    train(g, graph_inputs, opt.output_dir, attrList, layers=graph_kwargs['layers'], save_freq=opt.model_save_freq,
          trainEmbed=opt.trainEmbed, updateGAN=opt.updateGAN, opt=opt, multi_attr=multi_attr)
