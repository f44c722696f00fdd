"""Evaluation driver: attribute preservation of a trained walk (reference eval.py:21-239).

For every batch and every target attribute the batch is edited at ``num_panels`` alphas, every sample is bucketed by how
far the target attribute moved (|d| <= 0.3 / 0.6 / 1, graph.attribute_change_bucket) and the metric is the mean absolute
change of the 39 OTHER regressor outputs per bucket.  The identity-preservation half of the reference (facenet
InceptionResnetV1 cosine similarity, eval.py:27-31,170-189) needs a third-party network that is not part of this path and
is not built.

Reference quirk kept: eval.py:203-209 sit OUTSIDE the batch loop, so only the buckets of the LAST batch / LAST target
attribute enter the printed metric; ``all_batches=True`` accumulates every batch instead."""
import os
from collections import OrderedDict

import numpy as np

from . import constants, dist, hostutil
from .vis import VisOptions


def attribute_preservation(multi_attrs, original_attrs, index_):
    """eval.py:211-239: per non-empty bucket, (sum, mean) of |edited - original| over all attributes but ``index_``."""
    results, results_avg = [], []
    for k in range(3):
        org_k, new_k = np.array(original_attrs[k]), np.array(multi_attrs[k])
        if org_k.shape[0] == 0:
            continue
        org = np.hstack([org_k[:, :int(index_)], org_k[:, int(index_ + 1):]])
        changed = np.hstack([new_k[:, :int(index_)], new_k[:, int(index_ + 1):]])
        results.append(np.sum(np.abs(changed - org)))
        results_avg.append(np.mean(np.abs(changed - org)))
    return results, results_avg


def main(argv=None, all_batches=False):
    from . import graph as graph_mod
    v = VisOptions()
    v.initialize()
    v.parser.add_argument('--num_samples', type=int, default=10)
    v.parser.add_argument('--num_panels', type=int, default=7)
    v.parser.add_argument('--max_alpha', type=float, default=1)
    v.parser.add_argument('--min_alpha', type=float, default=0)
    v.parser.add_argument('--layers', type=str, default=None)
    v.parser.add_argument('--target_attrList', type=str, default=None)
    v.parser.add_argument('--trainEmbed', action='store_true')
    v.parser.add_argument('--updateGAN', action='store_true')
    opt, conf = v.parse(argv)
    dist.select_gpu(opt.gpu)                         # before the first torch.cuda call (vis_w.py / eval.py set CUDA_VISIBLE_DEVICES)
    dist.init_from_env()
    if getattr(conf, 'synthetic_weights', False):    # the training run's explicit choice travels in its opt.yml
        constants.ALLOW_SYNTHETIC_WEIGHTS = True
    if getattr(conf, 'resolution', None):
        constants.resolution = conf.resolution
    if getattr(conf, 'batch_size', None):
        constants.BATCH_SIZE = conf.batch_size
    output_dir = opt.output_dir if opt.output_dir else os.path.join(conf.output_dir, 'images')
    os.makedirs(output_dir, exist_ok=True)
    g = graph_mod.find_model_using_name(conf.model, conf.transform)(**hostutil.set_graph_kwargs(conf))
    g.load_multi_models(opt.save_path_w, None, trainEmbed=opt.trainEmbed, updateGAN=opt.updateGAN)
    graph_inputs = hostutil.graph_input(g, opt.num_samples, seed=opt.noise_seed)
    epochs = opt.save_path_w.split('/')[-1].split('_')[2]
    filename = os.path.join(output_dir, 'w_{}_seed{}'.format(epochs, opt.noise_seed))
    name = conf.attrList.strip().split(',')[0]
    layers = None if opt.layers in (None, 'None') else [int(i) for i in opt.layers.split(',')]

    attr_path = opt.attrPath or conf.attrPath
    attrList, attrTable = [], OrderedDict()
    with open(attr_path, 'r') as f:
        for i, line in enumerate(f.readlines()):
            if line.strip():
                attrList.append(line.strip())
                attrTable.update({line.strip(): i})
    assert len(attrList) == 40, ' len(attrList) should be 40'
    own = attrList if not conf.attrList else conf.attrList.split(',')
    target_attrList = own if not opt.target_attrList else opt.target_attrList.strip().split(',')
    print('target_attrList: ', target_attrList)

    multi_attrs, original_attrs = [[], [], []], [[], [], []]
    multi_attr = org_attr = None
    index_ = None
    bs = constants.BATCH_SIZE
    for batch_start in range(0, opt.num_samples, bs):
        s = slice(batch_start, min(opt.num_samples, batch_start + bs))
        batch = hostutil.batch_input(graph_inputs, s)
        new_filename = filename + '_{}_max{}_min{}'.format(name, opt.max_alpha, opt.min_alpha)
        ag, at = g.vis_image_batch(batch, new_filename, s.start, num_panels=opt.num_panels, max_alpha=opt.max_alpha,
                                   min_alpha=opt.min_alpha, wgt=True)
        for t in target_attrList:
            index_ = attrTable[t]
            multi_attr, org_attr, _, _ = g.vis_multi_image_batch_alphas_compute_multi_attr(
                batch, new_filename + '_attr_%s' % t, alphas_to_graph=ag, alphas_to_target=at, layers=layers, batch_start=s.start,
                wgt=False, wmask=False, trainEmbed=opt.trainEmbed, computeL2=False, index_=index_)
            if all_batches:
                for k in range(3):
                    multi_attrs[k] += multi_attr[k]
                    original_attrs[k] += org_attr[k]
    if not all_batches and multi_attr is not None:
        for k in range(3):
            multi_attrs[k] += multi_attr[k]
            original_attrs[k] += org_attr[k]
    results, results_avg = attribute_preservation(multi_attrs, original_attrs, index_)
    print('[ATTRIBUTE PRESERVATION] Results on 3 epsilon segments', ['%.4f' % i for i in results_avg])
    return dict(results=results, results_avg=results_avg, bucket_sizes=[len(m) for m in multi_attrs], index_=index_)
